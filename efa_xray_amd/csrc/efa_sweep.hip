// Fused covariance / gain / square-root update sweep (Phase B and the obs-row
// part of Phase A) and the serial obs-space diag kernel (Phase A), float64.
//
// Replaces, per observation k of a batch (ensrf.py line numbers):
//   kcov = Xbp . ye / (Nens-1)          :95
//   kcov *= taper                       :115
//   kmat = kcov / kdenom                :119
//   xam  = xbm + kmat*innov             :130
//   Xap  = Xbp - (beta*kmat)^T . ye     :136-141
// for every row of the (rows x M) block, all obs of the batch applied while
// the row sits in registers: one HBM read + one write of the block per batch.
//
// Thread layout ("L lanes per row", L = 4 or 16): L consecutive lanes own one
// row; lane j of the group holds members {2L*c+2j, 2L*c+2j+1 : c = 0..NC-1}, so
// every global access is a 16-byte load and a group covers 32*L contiguous
// bytes per chunk.  A wave64 streams 64/L consecutive rows (contiguous in
// memory).  The M-long dot product is 2*NC FMAs per lane + a DPP butterfly over
// the group; no LDS traffic for the state, LDS only broadcasts the batch's ye.
//   L = 4 : fewest instructions per row -- the throughput layout (state rows).
//   L = 16: shortest per-observation dependency chain -- the latency layout,
//           used when there are too few rows to fill the chip (the obs block).
#include "efa_device.h"
#include "efa_internal.h"
#include "efa_rows.h"

namespace efa {

namespace {

constexpr int kThreads = 256;

// ---------------------------------------------------------------------------
// Sweep: rows x batch.
// LDS: ye_s[nb][S] (zero padded, S = 2*L*NC) then coef_s[nb][4].
// ---------------------------------------------------------------------------
template <int L, int NC, bool VEC>
__global__ __launch_bounds__(kThreads) void k_sweep(const SweepArgs a) {
  extern __shared__ __align__(16) double smem[];
  constexpr int S = 2 * L * NC;
  constexpr int kRowsPerBlock = kThreads / L;
  double* ye_s = smem;
  double* coef_s = smem + (size_t)a.nb * S;
  double* wtab = coef_s + (size_t)a.nb * kCoefStride;  // [rows per block][kMaxBatch], table mode only
  const int tid = threadIdx.x;
  const int M = a.M;

  if (VEC) {
    // 16-byte copies, four in flight per thread (the batch image is small and latency-bound)
    constexpr int S2 = S / 2;
    double2* dst = reinterpret_cast<double2*>(ye_s);
    const int M2 = M / 2;
    const int total = a.nb * S2;
#pragma unroll 4
    for (int i = tid; i < total; i += kThreads) {
      const int k = i / S2, m2 = i - k * S2;
      dst[i] = (m2 < M2) ? reinterpret_cast<const double2*>(a.Ye + (size_t)k * a.ye_stride)[m2]
                         : make_double2(0.0, 0.0);
    }
  } else {
#pragma unroll 4
    for (int i = tid; i < a.nb * S; i += kThreads) {
      const int k = i / S, m = i - k * S;
      ye_s[i] = (m < M) ? a.Ye[(size_t)k * a.ye_stride + m] : 0.0;
    }
  }
  for (int i = tid; i < a.nb * kCoefStride; i += kThreads) coef_s[i] = a.coef[i];
  __syncthreads();

  const int j = tid & (L - 1);
  const int r = tid / L;
  const double rM1 = 1.0 / (double)(M - 1);
  const long nblocks = (a.nrows + kRowsPerBlock - 1) / kRowsPerBlock;
  unsigned long long assim_mask = 0ull;  // obs of the batch that are assimilated (wave-uniform)
  for (int k = 0; k < a.nb; ++k)
    if (__builtin_amdgcn_readfirstlane((int)(coef_s[k * kCoefStride + 3] != 0.0))) assim_mask |= 1ull << k;

  for (long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const long row = blk * kRowsPerBlock + r;
    const bool live = (row < a.nrows) && !(row >= a.skip_lo && row < a.skip_hi);
    double x[2 * NC];
    double xm = 0.0;
    if (live) {
      load_row<L, NC, VEC>(a.Xin + (size_t)row * M, M, j, x);
      xm = a.xin[row];
    } else {
#pragma unroll
      for (int c = 0; c < 2 * NC; ++c) x[c] = 0.0;
    }
    // one observation applied to the row held in registers
    auto apply = [&](int k, double w) {
      const double* ck = coef_s + k * kCoefStride;
      double y[2 * NC];
      lds_read_row<L, NC>(ye_s + k * S, j, y);
      const double dot = group_dot<L, NC>(x, y);
      double kc = dot * rM1;         // kcov = dot/(Nens-1)
      kc = w * kc;                   // localisation
      const double km = kc * ck[1];  // kmat = kcov/kdenom
      xm = xm + km * ck[0];          // xam = xbm + kmat*innov
      const double kb = ck[2] * km;  // beta*kmat
#pragma unroll
      for (int c = 0; c < 2 * NC; ++c) x[c] = __builtin_fma(-kb, y[c], x[c]);
    };

    if (a.taper_mode == kTaperTable) {
      // All nb taper values of this row are fetched up front (16 independent loads per lane,
      // lane j takes obs k = j mod 4) and parked in LDS, and the wave's set of obs with any
      // non-zero weight is reduced to one 64-bit mask: the k loop then visits only those obs
      // and never waits on a dependent global load.  Rows beyond 2*halfwidth of every ob of the
      // batch cost one load burst and no arithmetic (F3: zero weight == bit-unchanged row).
      const long col = live ? (row % a.ncol) : 0;
      double* wrow = wtab + (size_t)r * kMaxBatch;
      unsigned long long act = 0ull;
#pragma unroll
      for (int i = 0; i < kMaxBatch / L; ++i) {
        const int kk = L * i + j;
        const double wv = (live && kk < a.nb) ? a.W[(size_t)kk * a.ncol + col] : 0.0;
        wrow[kk] = wv;
        const unsigned long long bal = __ballot(wv != 0.0);
#pragma unroll
        for (int qq = 0; qq < L; ++qq)
          if (bal & (0x1111111111111111ull << qq)) act |= 1ull << (L * i + qq);
      }
      act &= assim_mask;
      while (act) {
        const int k = __builtin_ctzll(act);
        act &= act - 1;
        apply(k, wrow[k]);
      }
    } else {
      double rlat = 0.0, rlon = 0.0;
      const bool obs_taper = (a.taper_mode == kTaperObs) && live && (row < a.taper_rows);
      if (obs_taper) {
        rlat = a.row_lat[row];
        rlon = a.row_lon[row];
      }
      for (int k = 0; k < a.nb; ++k) {
        if (!((assim_mask >> k) & 1ull)) continue;  // not assimilated (uniform)
        double w = 1.0;
        if (obs_taper) w = gaspari_cohn(haversine_km(a.ob_lat[k], a.ob_lon[k], rlat, rlon), a.ob_hw[k]);
        apply(k, w);
      }
    }

    if (live) {
      store_row<L, NC, VEC>(a.Xout + (size_t)row * M, M, j, x);
      if (j == 0) a.xout[row] = xm;
    }
  }
}

// ---------------------------------------------------------------------------
// Diag: the batch's own nb obs rows, serial in k.  One workgroup, quad per row.
//
// The serial chain is kept short:
//  - the quad that owns row k+1 computes that row's variance and the scalar gain
//    factors right after it has applied ob k and publishes (ye, mye, mean(ye),
//    innov, rden, beta) through LDS; every other quad only does dot + update;
//  - one barrier per step and NO vector-memory operation inside the loop (the
//    workgroup barrier implies vmcnt(0): a global store per step would put its
//    completion latency on every step).  Published ye rows stay in an LDS ring
//    [nb][S] and are copied to Ye_rec after the loop; per-ob scalars stay in the
//    owner's registers;
//  - the row mean that np.var needs is carried incrementally
//    (mean(y - kb*ye) = mean(y) - kb*mean(ye)), so the variance is one reduction;
//  - the ob's own posterior variance is (1-kb)^2 * var (its row is scaled by
//    1-kb, ensrf.py:141), not a second reduction.
// LDS: ye_ring[nb][S], sc[nb][8], tw[nb][nb] (GC only).
// ---------------------------------------------------------------------------
template <int NC, bool VEC>
__global__ __launch_bounds__(kThreads) void k_diag(const DiagArgs a) {
  extern __shared__ __align__(16) double smem[];
  constexpr int L = 4;
  constexpr int S = 2 * L * NC;
  const int nb = a.nb;
  double* ye_ring = smem;                  // [nb][S]
  double* sc = smem + (size_t)nb * S;      // [nb][8]: mye, ymean, innov, rden, beta, active
  double* tw = sc + (size_t)nb * 8;        // [nb][nb]
  const int tid = threadIdx.x;
  const int q = tid & 3;
  const int r = tid >> 2;
  const int M = a.M;
  const bool live = r < nb;
  const long row = a.b0 + r;
  const double rM1 = 1.0 / (double)(M - 1);
  const double dM = (double)M;
  const double invM = 1.0 / dM;

  if (a.loc_mode != 0) {
    // taper of ob k against ob j of the batch: observation.py:68-83
    for (int i = tid; i < nb * nb; i += kThreads) {
      const int k = i / nb, jj = i - k * nb;
      const double d = haversine_km(a.ob_lat[a.b0 + k], a.ob_lon[a.b0 + k], a.ob_lat[a.b0 + jj],
                                    a.ob_lon[a.b0 + jj]);
      tw[i] = gaspari_cohn(d, a.ob_hw[a.b0 + k]);
    }
  }

  double x[2 * NC];
  double xm = 0.0, my_val = 0.0, my_err = 1.0;
  bool my_asm = false;
  if (live) {
    load_row<L, NC, VEC>(a.Yp + (size_t)row * M, M, q, x);
    xm = a.ym[row];
    my_val = a.ob_value[row];
    my_err = a.ob_error[row];
    my_asm = a.ob_assim[row] != 0;
  } else {
#pragma unroll
    for (int c = 0; c < 2 * NC; ++c) x[c] = 0.0;
  }
  double rmean = group_rowsum<L, NC>(x) / dM;  // padding slots hold 0

  double o_prior_mean = 0.0, o_prior_var = 0.0, o_innov = 0.0, o_rden = 0.0, o_beta = 0.0;
  double o_post_mean = 0.0, o_post_var = 0.0;
  bool o_done = false;

  // publish(k): called by the quad that owns row k, with its row current through ob k-1
  auto publish = [&](int k) {
    double* yb = ye_ring + (size_t)k * S;
#pragma unroll
    for (int c = 0; c < NC; ++c)
      *reinterpret_cast<double2*>(yb + 2 * L * c + 2 * q) = make_double2(x[2 * c], x[2 * c + 1]);
    // np.var, ddof=0 (:69).  Padding slots hold exactly 0, so they add mean^2 each: remove it.
    const double ss = group_sumsq_about<L, NC>(x, rmean) - (double)(S - M) * (rmean * rmean);
    const double varye = ss * invM;
    const double innov = my_val - xm;                                       // :85
    const double kdenom = varye + my_err;                                   // :91
    const double rden = 1.0 / kdenom;
    const double beta = 1.0 / (1.0 + sqrt(my_err * rden));                  // :135
    if (q == 0) {
      double* s = sc + (size_t)k * 8;
      s[0] = xm;
      s[1] = rmean;
      s[2] = innov;
      s[3] = rden;
      s[4] = beta;
      s[5] = my_asm ? 1.0 : 0.0;
    }
    o_prior_mean = xm;     // :66
    o_prior_var = varye;   // :70
    o_innov = innov;
    o_rden = rden;
    o_beta = beta;
  };

  if (r == 0) publish(0);

  for (int k = 0; k < nb; ++k) {
    __syncthreads();
    const double* s = sc + (size_t)k * 8;
    const bool active = s[5] != 0.0;  // uniform
    if (active) {
      double y[2 * NC];
      lds_read_row<L, NC>(ye_ring + (size_t)k * S, q, y);
      const double dot = group_dot<L, NC>(x, y);
      double kc = dot * rM1;                                                 // :95
      if (a.loc_mode != 0) kc = (live ? tw[k * nb + r] : 0.0) * kc;          // :115
      const double km = kc * s[3];                                           // :119
      xm = xm + km * s[2];                                                   // :130
      const double kb = s[4] * km;                                           // :136
      rmean = __builtin_fma(-kb, s[1], rmean);
#pragma unroll
      for (int c = 0; c < 2 * NC; ++c) x[c] = __builtin_fma(-kb, y[c], x[c]);  // :141
      if (r == k) {
        // posterior diagnostics of ob k (:144-149): its own row was scaled by (1 - kb)
        const double f = 1.0 - kb;
        o_post_var = (f * f) * o_prior_var;
        o_post_mean = xm;
        o_done = true;
      }
    }
    if (r == k + 1 && k + 1 < nb) publish(k + 1);
  }
  __syncthreads();

  // write-back: recorded trajectory, rows, per-ob scalars
  {
    const int S2 = S / 2;
    for (int i = tid; i < nb * S2; i += kThreads) {
      const int k = i / S2, m = 2 * (i - k * S2);
      double* dst = a.Ye_rec + (size_t)(a.b0 + k) * M;
      const double2 v = *reinterpret_cast<const double2*>(ye_ring + (size_t)k * S + m);
      if (VEC) {
        if (m < M) *reinterpret_cast<double2*>(dst + m) = v;
      } else {
        if (m < M) dst[m] = v.x;
        if (m + 1 < M) dst[m + 1] = v.y;
      }
    }
  }
  if (live) {
    store_row<L, NC, VEC>(a.Yp + (size_t)row * M, M, q, x);
    if (q == 0) {
      a.ym[row] = xm;
      a.prior_mean[row] = o_prior_mean;
      a.prior_var[row] = o_prior_var;
      double* ck = a.coef + (size_t)row * kCoefStride;
      ck[0] = my_asm ? o_innov : 0.0;
      ck[1] = my_asm ? o_rden : 0.0;
      ck[2] = my_asm ? o_beta : 0.0;
      ck[3] = my_asm ? 1.0 : 0.0;
      a.assimilated[row] = o_done ? 1 : 0;   // :74-76, :149
      if (o_done) {
        a.post_mean[row] = o_post_mean;
        a.post_var[row] = o_post_var;
      }
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <int L, int NC>
hipError_t sweep_launch(const SweepArgs& a, bool vec, hipStream_t s) {
  constexpr int kRowsPerBlock = kThreads / L;
  const long nblocks = (a.nrows + kRowsPerBlock - 1) / kRowsPerBlock;
  const size_t lds = ((size_t)a.nb * 2 * L * NC + (size_t)a.nb * kCoefStride +
                      (a.taper_mode == kTaperTable ? (size_t)kRowsPerBlock * kMaxBatch : 0)) * sizeof(double);
  long grid = nblocks < 256L * 8 ? nblocks : 256L * 8;
  if (grid < 1) grid = 1;
  if (lds > 64 * 1024) {
    hipError_t e = vec ? hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sweep<L, NC, true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                       : hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sweep<L, NC, false>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  if (vec)
    hipLaunchKernelGGL((k_sweep<L, NC, true>), dim3((unsigned)grid), dim3(kThreads), lds, s, a);
  else
    hipLaunchKernelGGL((k_sweep<L, NC, false>), dim3((unsigned)grid), dim3(kThreads), lds, s, a);
  return hipGetLastError();
}

template <int NC>
hipError_t sweep4(const SweepArgs& a, bool vec, hipStream_t s) { return sweep_launch<4, NC>(a, vec, s); }

template <int NC>
hipError_t diag_nc(const DiagArgs& a, bool vec, hipStream_t s) {
  const size_t lds = diag_lds_bytes(8 * NC, a.nb, a.loc_mode);
  if (lds > 64 * 1024) {
    hipError_t e;
    if (vec)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_diag<NC, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_diag<NC, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  if (vec)
    hipLaunchKernelGGL((k_diag<NC, true>), dim3(1), dim3(kThreads), lds, s, a);
  else
    hipLaunchKernelGGL((k_diag<NC, false>), dim3(1), dim3(kThreads), lds, s, a);
  return hipGetLastError();
}

}  // namespace

#define EFA_NCH_SWITCH(FN, ...)                                  \
  switch (nch) {                                                 \
    case 1: return FN<1>(__VA_ARGS__);                           \
    case 2: return FN<2>(__VA_ARGS__);                           \
    case 3: return FN<3>(__VA_ARGS__);                           \
    case 4: return FN<4>(__VA_ARGS__);                           \
    case 5: return FN<5>(__VA_ARGS__);                           \
    case 6: return FN<6>(__VA_ARGS__);                           \
    case 7: return FN<7>(__VA_ARGS__);                           \
    case 8: return FN<8>(__VA_ARGS__);                           \
    case 9: return FN<9>(__VA_ARGS__);                           \
    case 10: return FN<10>(__VA_ARGS__);                         \
    case 11: return FN<11>(__VA_ARGS__);                         \
    case 12: return FN<12>(__VA_ARGS__);                         \
    case 13: return FN<13>(__VA_ARGS__);                         \
    case 14: return FN<14>(__VA_ARGS__);                         \
    case 15: return FN<15>(__VA_ARGS__);                         \
    case 16: return FN<16>(__VA_ARGS__);                         \
    case 20: return FN<20>(__VA_ARGS__);                         \
    case 24: return FN<24>(__VA_ARGS__);                         \
    case 32: return FN<32>(__VA_ARGS__);                         \
    default: return hipErrorInvalidValue;                        \
  }

int sweep_slots(int M) {  // padded row length (doubles) of the quad layout's LDS image
  int nch = (M + 7) / 8;
  if (nch > 16) nch = (nch <= 20) ? 20 : (nch <= 24) ? 24 : 32;
  return 8 * nch;
}

size_t diag_lds_bytes(int slots, int nb, int loc_mode) {
  return ((size_t)nb * slots + (size_t)nb * 8 + (loc_mode ? (size_t)nb * nb : 0)) * sizeof(double);
}

hipError_t launch_sweep(const SweepArgs& a, hipStream_t s) {
  if (a.M < 2 || a.M > kMaxMembers || a.nb < 1 || a.nb > kMaxBatch) return hipErrorInvalidValue;
  if (a.nrows <= 0) return hipSuccess;
  const bool vec = (a.M % 2 == 0) && (a.ye_stride % 2 == 0) && aligned16(a.Xin) && aligned16(a.Xout) && aligned16(a.Ye);
  const int nch = sweep_slots(a.M) / 8;
  EFA_NCH_SWITCH(sweep4, a, vec, s)
}

hipError_t launch_diag(const DiagArgs& a, hipStream_t s) {
  if (a.M < 2 || a.M > kMaxMembers || a.nb < 1 || a.nb > kMaxBatch) return hipErrorInvalidValue;
  const bool vec = (a.M % 2 == 0) && aligned16(a.Yp) && aligned16(a.Ye_rec);
  const int nch = sweep_slots(a.M) / 8;
  EFA_NCH_SWITCH(diag_nc, a, vec, s)
}

}  // namespace efa
