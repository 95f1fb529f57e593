// Unlocalised collapse of the serial EnSRF loop into one pass over the state.
//
// With loc in (None, False) every row of the reference's update
//   kcov_i = Xbp_i . ye_k/(Nens-1); xam_i += kcov_i/kdenom*innov;
//   Xap_i  = Xbp_i - beta*kcov_i/kdenom * ye_k          (ensrf.py:95-141)
// is the same linear map of that row for all i, so after P observations
//   Xap = Xbp * T        (T: M x M)      xam = xbm + Xbp * w    (w: M)
// where (T, w) are what the loop does to the rows of the identity matrix
// (carried through Phase A as M extra rows of the obs block).  This kernel
// applies [T | w] to every state row: one HBM read and one write of the state
// for the whole assimilation cycle.
//
// It is a (rows x M) . (M x (M+1)) float64 contraction, done on the matrix
// cores with v_mfma_f64_16x16x4_f64:
//   - a wave owns a tile of 16 consecutive rows (contiguous in memory);
//   - lane l = (g = l>>4, n = l&15) loads members {8u+2g, 8u+2g+1} of row n with
//     16-byte global loads (a quad of lane-groups covers 64 contiguous bytes);
//     the K order of the contraction is permuted to match, so the A operand
//     goes HBM -> VGPR -> MFMA with no LDS staging;
//   - [T | w] lives in LDS (once per workgroup), pre-arranged so each MFMA's B
//     operand is one conflict-free ds_read_b64 per lane;
//   - the next tile's rows are prefetched while the current tile is multiplied.
#include "efa_device.h"
#include "efa_internal.h"

namespace efa {
namespace {

typedef double v4f64 __attribute__((ext_vector_type(4)));

#ifndef EFA_T_WAVES
#define EFA_T_WAVES 8
#endif
#ifndef EFA_T_PREFETCH
#define EFA_T_PREFETCH 1
#endif
constexpr int kThreadsT = 64 * EFA_T_WAVES;  // waves per workgroup, sharing its LDS copy of [T|w]
constexpr bool kPrefetchT = EFA_T_PREFETCH != 0;

// Column tiling of the product.  Perturbation form (!FUSED): M member columns plus the column of w (the mean
// increment): NU/2 + 1 tiles of 16.  Member form (FUSED: prior members in, posterior members out) needs no
// separate w column -- posterior_j = mean + sum_k Xp_k (T_kj + w_k), so w is folded into the LDS image of T --
// and when the last 16-column tile would hold at most four members (M % 16 in 1..4, i.e. HALF with NU odd) those
// columns are done by ONE v_mfma_f64_4x4x4_4b_f64 per K step (17 cycles instead of 64) over the same A operand:
// its four 4 x 4 x 4 blocks are the tile's four row groups, A[blk][i][k] sits in lane 16 k + 4 blk + i = 16 k + row
// exactly as in the 16 x 16 x 4 instruction, B[blk][k][j] in lane 16 k + 4 blk + j, D[blk][i][j] in lane
// 16 i + 4 blk + j (tools/mfma44_layout_probe.hip).  M = 100: 6 x 64 + 17 cycles per K step instead of 7 x 64.
template <int NU, bool FUSED, bool HALF>
struct TShape {
  static constexpr bool NARROW = FUSED && HALF && (NU & 1);
  static constexpr int NT = !FUSED ? NU / 2 + 1 : (NARROW ? NU / 2 : (NU + 1) / 2);
  static constexpr int NTA = NT > 0 ? NT : 1;    // array extent (NT = 0: M <= 4 in member form)
  static constexpr int kSteps = 2 * NU;          // K steps of 4 members
  static constexpr size_t wide_doubles = (size_t)kSteps * NT * 64;
  static constexpr size_t lds_doubles = wide_doubles + (NARROW ? (size_t)kSteps * 64 : 0);
};

// Loads of one tile are branch-free (clamped addresses; the partial last chunk is zeroed
// by a select at the point of use) and the prefetch is unconditional, so the compiler can
// keep the next tile's 16-byte loads in flight behind counted vmcnt waits while the current
// tile is multiplied.  With conditional loads it falls back to vmcnt(0) and the overlap is lost.
// HALF: the last chunk holds at most four members (M % 8 in 1..4).  Lane-group g then loads member
// 8(NU-1)+g alone, so the chunk is ONE K step of the matrix core instead of two half-empty ones
// (M = 100: 25 steps instead of 26).
// AL: rows are 16-byte aligned (M even and an aligned base): 16-byte loads.  Otherwise (odd M) the two
// members of a pair are loaded separately -- half the load width, the price of an odd ensemble size only.
template <int NU, bool HALF, bool AL>
__device__ __forceinline__ void load_tile(const double* __restrict__ X, long row_clamped, int M, int g,
                                          double (&a)[2 * NU]) {
  const double* p = X + (size_t)row_clamped * M;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (HALF && u == NU - 1) {
      const int m = 8 * u + g;
      a[2 * u] = p[(m < M) ? m : M - 1];
      a[2 * u + 1] = 0.0;
    } else if (AL) {
      int m0 = 8 * u + 2 * g;
      if (u == NU - 1) m0 = (m0 < M) ? m0 : M - 2;  // last chunk may be partial: clamp the address
      const double2 v = *reinterpret_cast<const double2*>(p + m0);
      a[2 * u] = v.x;
      a[2 * u + 1] = v.y;
    } else {
      const int m0 = 8 * u + 2 * g;
      a[2 * u] = p[(m0 < M) ? m0 : M - 1];          // clamped addresses; invalid slots are zeroed at the point of use
      a[2 * u + 1] = p[(m0 + 1 < M) ? m0 + 1 : M - 1];
    }
  }
}

template <int NU, bool FUSED, bool HALF, bool AL>
__global__ __launch_bounds__(kThreadsT) void k_transform(const TransformArgs p) {
  using Sh = TShape<NU, FUSED, HALF>;
  constexpr int NT = Sh::NT;
  constexpr bool NARROW = Sh::NARROW;
  extern __shared__ __align__(16) double Bs[];  // [(u*2+h)*NT + t][64], then the narrow tile's [(u*2+h)][64]
  const int M = p.M;
  const int tid = threadIdx.x;

  // Stage [T | w] (member form: T + w 1^T) in MFMA-B order: Bs[step s=(u,h)][t][lane (g,n)] = Text[8u+2g+h][16t+n]
  for (int i = tid; i < (int)Sh::lds_doubles; i += kThreadsT) {
    const int l = i & 63;
    const int g = l >> 4, n = l & 15;
    int s, j;
    if (i < (int)Sh::wide_doubles) {
      const int st = i >> 6;
      s = st / Sh::NTA;
      j = 16 * (st % Sh::NTA) + n;
    } else {  // narrow tile: B[blk][k = g][j = l & 3] for every block
      s = (i - (int)Sh::wide_doubles) >> 6;
      j = 16 * NT + (l & 3);
    }
    const int u = s >> 1, h = s & 1;
    const int m = (HALF && u == NU - 1) ? (h == 0 ? 8 * u + g : M) : 8 * u + 2 * g + h;  // (row M: never used, stays 0)
    double v = 0.0;
    if (m < M) {
      if (j < M) v = FUSED ? p.T[(size_t)m * M + j] + p.w[m] : p.T[(size_t)m * M + j];
      else if (j == M && !FUSED) v = p.w[m];
    }
    Bs[i] = v;
  }
  __syncthreads();

  const int lane = tid & 63;
  const int g = lane >> 4, n = lane & 15;
#ifdef EFA_T_CLOCKSTAMP  /* tools/transform_clock.py: shader clock over the kernel (non-fused launches only) */
  const unsigned long long tm0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
#endif
  const long ntiles = (p.nrows + 15) / 16;
  const long wave = (long)blockIdx.x * (kThreadsT / 64) + (tid >> 6);
  const long nwaves = (long)gridDim.x * (kThreadsT / 64);
  const int tM = M >> 4, nM = M & 15;  // tile / lane column holding the mean increment
  const long last_row = p.nrows - 1;
  // this lane's two slots of the last chunk are real members?  (the second only matters for an odd M)
  const bool last_ok = (HALF ? (8 * (NU - 1) + g) : (8 * (NU - 1) + 2 * g)) < M;
  const bool last_ok1 = !HALF && (8 * (NU - 1) + 2 * g + 1) < M;

  double a[2 * NU], an[2 * NU];
  long tile = wave;
  if (tile < ntiles) {
    const long r = tile * 16 + n;
    load_tile<NU, HALF, AL>(p.Xin, r < last_row ? r : last_row, M, g, a);
  }
  if (kPrefetchT) {  // the first tile has arrived before the loop is entered: its header then needs no vmcnt wait, which on
#pragma unroll      // the back edge would also wait for the previous tile's stores
    for (int c = 0; c < 2 * NU; ++c) asm volatile("" : "+v"(a[c]));
  }

  while (tile < ntiles) {
    const long next = tile + nwaves;
    const long r0 = tile * 16;
    double xm4[4] = {0.0, 0.0, 0.0, 0.0};  // prior means of rows 4v+g (perturbation form)
    if (!FUSED) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const long rr = r0 + 4 * v + g;
        xm4[v] = p.xin[rr < last_row ? rr : last_row];
      }
    }
    if (kPrefetchT) {  // prefetch (the last iteration harmlessly re-reads its own tile)
      const long r = (next < ntiles ? next : tile) * 16 + n;
      load_tile<NU, HALF, AL>(p.Xin, r < last_row ? r : last_row, M, g, an);
    }
    if (!last_ok) a[2 * NU - 2] = 0.0;
    if (!last_ok1) a[2 * NU - 1] = 0.0;

    double rmean = 0.0;  // prior mean of row n (FUSED only)
    if (FUSED) {
      double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int c = 0; c < 2 * NU; ++c) s4[c & 3] += a[c];
      double s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      rmean = s / (double)M;
#pragma unroll
      for (int c = 0; c < 2 * NU - 2; ++c) a[c] -= rmean;
      a[2 * NU - 2] = last_ok ? a[2 * NU - 2] - rmean : 0.0;
      a[2 * NU - 1] = last_ok1 ? a[2 * NU - 1] - rmean : 0.0;
    }

    v4f64 acc[Sh::NTA];
    double accn = 0.0;  // narrow tile: D[row 4 ((lane >> 2) & 3) + (lane >> 4)][col 16 NT + (lane & 3)]
#pragma unroll
    for (int t = 0; t < Sh::NTA; ++t) acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < (HALF ? 2 * NU - 1 : 2 * NU); ++s) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const double b = Bs[((size_t)s * NT + t) * 64 + lane];
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b, acc[t], 0, 0, 0);
      }
      if (NARROW) accn = __builtin_amdgcn_mfma_f64_4x4x4f64(a[s], Bs[Sh::wide_doubles + (size_t)s * 64 + lane], accn, 0, 0, 0);
      // keep the scheduler from hoisting all 2*NU*NT LDS reads (register blow-up)
      __builtin_amdgcn_sched_barrier(0);
    }

    // The next tile's rows are taken over HERE, before this tile's stores are issued: vmcnt counts loads and stores
    // in one queue on this architecture, so a wait for the prefetched loads placed AFTER the stores (the top of the
    // next iteration) is also a wait for the stores.  (Measured: the launch time is the same either way -- the stores
    // drain well within a tile's MFMA phase -- but this is the order that does not depend on it.)
    if (kPrefetchT) {
#pragma unroll
      for (int c = 0; c < 2 * NU; ++c) a[c] = an[c];
#pragma unroll
      for (int c = 0; c < 2 * NU; ++c) asm volatile("" : "+v"(a[c]));  // the wait for the loads belongs here
    }
    // acc[t][v] = D[row 4v+g][col 16t+n]  (layout probed on gfx950: tools/mfma_probe.hip)
    double base[4];  // posterior mean of rows 4v+g: prior mean + column M of the product
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (FUSED) {
        base[v] = __shfl(rmean, 4 * v + g, 64);  // w is folded into the image of T
      } else {
        double dm = 0.0;
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (t == tM) dm = __shfl(acc[t][v], (lane & 48) | nM, 64);
        base[v] = xm4[v] + dm;
      }
    }
    const int nrow = 4 * ((lane >> 2) & 3) + g;  // the narrow tile's row and column of this lane
    const int ncol = 16 * NT + (lane & 3);
    const double nval = NARROW ? __shfl(rmean, nrow, 64) + accn : 0.0;
    if (r0 + 16 <= p.nrows) {  // full tile: no row checks (wave-uniform)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = 16 * t + n;
        if (col < M) {
#pragma unroll
          for (int v = 0; v < 4; ++v)
            p.Xout[(size_t)(r0 + 4 * v + g) * M + col] = FUSED ? (base[v] + acc[t][v]) : acc[t][v];
        }
      }
      if (NARROW && ncol < M) p.Xout[(size_t)(r0 + nrow) * M + ncol] = nval;
      if (!FUSED && n == 0) {
#pragma unroll
        for (int v = 0; v < 4; ++v) p.xout[r0 + 4 * v + g] = base[v];
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = 16 * t + n;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const long row = r0 + 4 * v + g;
          if (col < M && row < p.nrows)
            p.Xout[(size_t)row * M + col] = FUSED ? (base[v] + acc[t][v]) : acc[t][v];
        }
      }
      if (NARROW && ncol < M && r0 + nrow < p.nrows) p.Xout[(size_t)(r0 + nrow) * M + ncol] = nval;
      if (!FUSED && n == 0) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const long row = r0 + 4 * v + g;
          if (row < p.nrows) p.xout[row] = base[v];
        }
      }
    }

    if (!kPrefetchT && next < ntiles) {
      const long r = next * 16 + n;
      load_tile<NU, HALF, AL>(p.Xin, r < last_row ? r : last_row, M, g, a);
    }
    tile = next;
  }
#ifdef EFA_T_CLOCKSTAMP
  if (!FUSED && blockIdx.x == 0 && tid == 0) {
    p.xout[0] = (double)(__builtin_amdgcn_s_memtime() - tm0);
    p.xout[1] = (double)(__builtin_amdgcn_s_memrealtime() - tr0);
  }
#endif
}

template <int NU, bool FUSED, bool HALF, bool AL>
hipError_t transform_launch(const TransformArgs& a, hipStream_t s) {
  using Sh = TShape<NU, FUSED, HALF>;
  const size_t lds = (Sh::lds_doubles ? Sh::lds_doubles : 64) * sizeof(double);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_transform<NU, FUSED, HALF, AL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  const long ntiles = (a.nrows + 15) / 16;
  const int per_cu = (lds <= 80 * 1024) ? 2 : 1;
  long grid = (ntiles + EFA_T_WAVES - 1) / EFA_T_WAVES;
  if (grid > 256L * per_cu) grid = 256L * per_cu;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_transform<NU, FUSED, HALF, AL>), dim3((unsigned)grid), dim3(kThreadsT), lds, s, a);
  return hipGetLastError();
}

template <int NU>
hipError_t transform_nu(const TransformArgs& a, hipStream_t s) {
  const int rem = a.M % 8;
  const bool half = rem != 0 && rem <= 4;
  const bool al = (a.M % 2 == 0) && (reinterpret_cast<uintptr_t>(a.Xin) & 15u) == 0;
  if (al) {
    if (half) return a.fused_members ? transform_launch<NU, true, true, true>(a, s) : transform_launch<NU, false, true, true>(a, s);
    return a.fused_members ? transform_launch<NU, true, false, true>(a, s) : transform_launch<NU, false, false, true>(a, s);
  }
  if (half) return a.fused_members ? transform_launch<NU, true, true, false>(a, s) : transform_launch<NU, false, true, false>(a, s);
  return a.fused_members ? transform_launch<NU, true, false, false>(a, s) : transform_launch<NU, false, false, false>(a, s);
}


// ---- ensembles of 137 .. 256 members ----------------------------------------------------------------------
// The LDS image of [T | w] no longer fits one CU (M^2 x 8 B > 160 KB), so the product is cut into column groups of
// kWideTiles 16-wide tiles: blockIdx.y takes one group, stages only that part of the image and re-reads the
// rows (2 .. 5 reads of the state instead of 1; still one write, and still ONE pass where the sweep path would
// make one read+write pass per 64 observations).  Same operand layouts as k_transform; no prefetch (the A
// operand of 256 members is 128 registers), no narrow last tile.
constexpr int kWideTiles = 4;
template <int NU, bool FUSED, bool AL>
__global__ __launch_bounds__(kThreadsT) void k_transform_wide(const TransformArgs p) {
  constexpr int NTG = kWideTiles;
  constexpr int kSteps = 2 * NU;
  extern __shared__ __align__(16) double Bs[];  // [(u*2+h)*NTG + t][64] for this group's tiles
  const int M = p.M;
  const int tid = threadIdx.x;
  const int t0 = (int)blockIdx.y * NTG;         // first 16-wide tile of this group
  for (int i = tid; i < kSteps * NTG * 64; i += kThreadsT) {
    const int l = i & 63, st = i >> 6;
    const int t = st % NTG, s = st / NTG;
    const int u = s >> 1, h = s & 1;
    const int g = l >> 4, n = l & 15;
    const int m = 8 * u + 2 * g + h;
    const int j = 16 * (t0 + t) + n;
    double v = 0.0;
    if (m < M) {
      if (j < M) v = FUSED ? p.T[(size_t)m * M + j] + p.w[m] : p.T[(size_t)m * M + j];
      else if (j == M && !FUSED) v = p.w[m];
    }
    Bs[i] = v;
  }
  __syncthreads();

  const int lane = tid & 63;
  const int g = lane >> 4, n = lane & 15;
  const long ntiles = (p.nrows + 15) / 16;
  const long nwaves = (long)gridDim.x * (kThreadsT / 64);
  const int tM = M >> 4, nM = M & 15;  // tile / lane column holding the mean increment (perturbation form)
  const long last_row = p.nrows - 1;
  for (long tile = (long)blockIdx.x * (kThreadsT / 64) + (tid >> 6); tile < ntiles; tile += nwaves) {
    const long r0 = tile * 16;
    const long r = r0 + n;
    double a[2 * NU];
    load_tile<NU, false, AL>(p.Xin, r < last_row ? r : last_row, M, g, a);
#pragma unroll
    for (int c = 0; c < 2 * NU; ++c) {  // slots beyond M (only in the last chunk) are zero
      const int m = 8 * (c >> 1) + 2 * g + (c & 1);
      if (m >= M) a[c] = 0.0;
    }
    double rmean = 0.0;
    if (FUSED) {
      double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int c = 0; c < 2 * NU; ++c) s4[c & 3] += a[c];
      double sm = (s4[0] + s4[1]) + (s4[2] + s4[3]);
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      rmean = sm / (double)M;
#pragma unroll
      for (int c = 0; c < 2 * NU; ++c) {
        const int m = 8 * (c >> 1) + 2 * g + (c & 1);
        a[c] = (m < M) ? a[c] - rmean : 0.0;
      }
    }
    v4f64 acc[NTG];
#pragma unroll
    for (int t = 0; t < NTG; ++t) acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
#pragma unroll
      for (int t = 0; t < NTG; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], Bs[((size_t)s * NTG + t) * 64 + lane], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    double base[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const long rr = r0 + 4 * v + g;
      if (FUSED) {
        base[v] = __shfl(rmean, 4 * v + g, 64);
      } else {
        double dm = 0.0;
#pragma unroll
        for (int t = 0; t < NTG; ++t)
          if (t0 + t == tM) dm = __shfl(acc[t][v], (lane & 48) | nM, 64);
        base[v] = p.xin[rr < last_row ? rr : last_row] + dm;
      }
    }
#pragma unroll
    for (int t = 0; t < NTG; ++t) {
      const int col = 16 * (t0 + t) + n;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const long row = r0 + 4 * v + g;
        if (col < M && row < p.nrows) p.Xout[(size_t)row * M + col] = FUSED ? (base[v] + acc[t][v]) : acc[t][v];
      }
    }
    if (!FUSED && n == 0 && tM >= t0 && tM < t0 + NTG) {  // the group that holds column M writes the means
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const long row = r0 + 4 * v + g;
        if (row < p.nrows) p.xout[row] = base[v];
      }
    }
  }
}

template <int NU>
hipError_t transform_wide_nu(const TransformArgs& a, hipStream_t s) {
  const size_t lds = (size_t)2 * NU * kWideTiles * 64 * sizeof(double);
  const bool al = (a.M % 2 == 0) && (reinterpret_cast<uintptr_t>(a.Xin) & 15u) == 0;
  const int ncols = a.fused_members ? a.M : a.M + 1;
  const int groups = ((ncols + 15) / 16 + kWideTiles - 1) / kWideTiles;
  const long ntiles = (a.nrows + 15) / 16;
  long gx = (ntiles + EFA_T_WAVES - 1) / EFA_T_WAVES;
  if (gx > 256) gx = 256;
  if (gx < 1) gx = 1;
  const dim3 grid((unsigned)gx, (unsigned)groups);
#define EFA_WIDE_LAUNCH(F, A)                                                                                       \
  do {                                                                                                              \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_transform_wide<NU, F, A>),                  \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
    if (e != hipSuccess) return e;                                                                                  \
    hipLaunchKernelGGL((k_transform_wide<NU, F, A>), grid, dim3(kThreadsT), lds, s, a);                              \
  } while (0)
  if (a.fused_members) {
    if (al) EFA_WIDE_LAUNCH(true, true);
    else EFA_WIDE_LAUNCH(true, false);
  } else {
    if (al) EFA_WIDE_LAUNCH(false, true);
    else EFA_WIDE_LAUNCH(false, false);
  }
#undef EFA_WIDE_LAUNCH
  return hipGetLastError();
}

}  // namespace

// one launch with the whole [T | w] image in LDS up to M = 136; column groups above (k_transform_wide) up to 256
bool transform_supported(int M) { return M >= 2 && M <= 256; }

hipError_t launch_transform(const TransformArgs& a, hipStream_t s) {
  if (!transform_supported(a.M)) return hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(a.Xin) & 7u) != 0) return hipErrorInvalidValue;
  if (a.nrows <= 0) return hipSuccess;
  const int nu = (a.M + 7) / 8;
  switch (nu) {
    case 1: return transform_nu<1>(a, s);
    case 2: return transform_nu<2>(a, s);
    case 3: return transform_nu<3>(a, s);
    case 4: return transform_nu<4>(a, s);
    case 5: return transform_nu<5>(a, s);
    case 6: return transform_nu<6>(a, s);
    case 7: return transform_nu<7>(a, s);
    case 8: return transform_nu<8>(a, s);
    case 9: return transform_nu<9>(a, s);
    case 10: return transform_nu<10>(a, s);
    case 11: return transform_nu<11>(a, s);
    case 12: return transform_nu<12>(a, s);
    case 13: return transform_nu<13>(a, s);
    case 14: return transform_nu<14>(a, s);
    case 15: return transform_nu<15>(a, s);
    case 16: return transform_nu<16>(a, s);
    case 17: return transform_nu<17>(a, s);
    case 18: return transform_wide_nu<18>(a, s);
    case 19: return transform_wide_nu<19>(a, s);
    case 20: return transform_wide_nu<20>(a, s);
    case 21: return transform_wide_nu<21>(a, s);
    case 22: return transform_wide_nu<22>(a, s);
    case 23: return transform_wide_nu<23>(a, s);
    case 24: return transform_wide_nu<24>(a, s);
    case 25: return transform_wide_nu<25>(a, s);
    case 26: return transform_wide_nu<26>(a, s);
    case 27: return transform_wide_nu<27>(a, s);
    case 28: return transform_wide_nu<28>(a, s);
    case 29: return transform_wide_nu<29>(a, s);
    case 30: return transform_wide_nu<30>(a, s);
    case 31: return transform_wide_nu<31>(a, s);
    case 32: return transform_wide_nu<32>(a, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace efa
