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
// Thread layout ("quad per row"): 4 consecutive lanes own one row; lane q of
// the quad holds members {8c+2q, 8c+2q+1 : c = 0..NCH-1} so that every global
// access is a 16-byte load and the 4 lanes of a quad cover 64 contiguous bytes.
// A wave64 therefore streams 16 consecutive rows (contiguous in memory).  The
// M-long dot product is 2*NCH FMAs per lane + a 2-step DPP butterfly over the
// quad; no LDS traffic for the state, LDS only broadcasts the batch's ye rows.
#include "efa_device.h"
#include "efa_internal.h"

namespace efa {

namespace {

constexpr int kThreads = 256;
constexpr int kRowsPerBlock = kThreads / 4;  // 64

template <int NCH, bool VEC>
__device__ __forceinline__ void load_row(const double* __restrict__ p, int M, int q, double (&x)[2 * NCH]) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int m0 = 8 * c + 2 * q;
    if (VEC) {
      if (m0 < M) {
        const double2 v = *reinterpret_cast<const double2*>(p + m0);
        x[2 * c] = v.x;
        x[2 * c + 1] = v.y;
      } else {
        x[2 * c] = 0.0;
        x[2 * c + 1] = 0.0;
      }
    } else {
      x[2 * c] = (m0 < M) ? p[m0] : 0.0;
      x[2 * c + 1] = (m0 + 1 < M) ? p[m0 + 1] : 0.0;
    }
  }
}

template <int NCH, bool VEC>
__device__ __forceinline__ void store_row(double* __restrict__ p, int M, int q, const double (&x)[2 * NCH]) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int m0 = 8 * c + 2 * q;
    if (VEC) {
      if (m0 < M) *reinterpret_cast<double2*>(p + m0) = make_double2(x[2 * c], x[2 * c + 1]);
    } else {
      if (m0 < M) p[m0] = x[2 * c];
      if (m0 + 1 < M) p[m0 + 1] = x[2 * c + 1];
    }
  }
}

// dot(x, ye) over the lane's slots with two accumulators, then the quad total.
template <int NCH>
__device__ __forceinline__ double quad_dot(const double (&x)[2 * NCH], const double (&y)[2 * NCH]) {
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    a0 = __builtin_fma(x[2 * c], y[2 * c], a0);
    a1 = __builtin_fma(x[2 * c + 1], y[2 * c + 1], a1);
  }
  return quad_sum(a0 + a1);
}

// ---------------------------------------------------------------------------
// Sweep: rows x batch.
// LDS: ye_s[nb][8*NCH] (zero padded) then coef_s[nb][4].
// ---------------------------------------------------------------------------
template <int NCH, bool VEC>
__global__ __launch_bounds__(kThreads) void k_sweep(const SweepArgs a) {
  extern __shared__ __align__(16) double smem[];
  constexpr int S = 8 * NCH;
  double* ye_s = smem;
  double* coef_s = smem + (size_t)a.nb * S;
  const int tid = threadIdx.x;
  const int M = a.M;

  for (int i = tid; i < a.nb * S; i += kThreads) {
    const int k = i / S, m = i - k * S;
    ye_s[i] = (m < M) ? a.Ye[(size_t)k * M + m] : 0.0;
  }
  for (int i = tid; i < a.nb * kCoefStride; i += kThreads) coef_s[i] = a.coef[i];
  __syncthreads();

  const int q = tid & 3;
  const int r = tid >> 2;
  const double rM1 = 1.0 / (double)(M - 1);
  const long nblocks = (a.nrows + kRowsPerBlock - 1) / kRowsPerBlock;

  for (long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const long row = blk * kRowsPerBlock + r;
    const bool live = (row < a.nrows) && !(row >= a.skip_lo && row < a.skip_hi);
    double x[2 * NCH];
    double xm = 0.0;
    if (live) {
      load_row<NCH, VEC>(a.Xin + (size_t)row * M, M, q, x);
      xm = a.xin[row];
    } else {
#pragma unroll
      for (int c = 0; c < 2 * NCH; ++c) x[c] = 0.0;
    }
    long col = 0;
    double rlat = 0.0, rlon = 0.0;
    const bool obs_taper = (a.taper_mode == kTaperObs) && live && (row < a.taper_rows);
    if (a.taper_mode == kTaperTable) col = live ? (row % a.ncol) : 0;
    if (obs_taper) {
      rlat = a.row_lat[row];
      rlon = a.row_lon[row];
    }

    for (int k = 0; k < a.nb; ++k) {
      const double* ck = coef_s + k * kCoefStride;
      if (ck[3] == 0.0) continue;  // not assimilated (uniform)
      double w = 1.0;
      if (a.taper_mode == kTaperTable) {
        w = live ? a.W[(size_t)k * a.ncol + col] : 0.0;
        if (__ballot(w != 0.0) == 0ull) continue;  // whole wave outside 2*halfwidth
      } else if (a.taper_mode == kTaperObs) {
        if (obs_taper) w = gaspari_cohn(haversine_km(a.ob_lat[k], a.ob_lon[k], rlat, rlon), a.ob_hw[k]);
      }
      double y[2 * NCH];
      const double* yk = ye_s + k * S + 2 * q;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const double2 v = *reinterpret_cast<const double2*>(yk + 8 * c);
        y[2 * c] = v.x;
        y[2 * c + 1] = v.y;
      }
      const double dot = quad_dot<NCH>(x, y);
      double kc = dot * rM1;   // kcov = dot/(Nens-1)
      kc = w * kc;             // localisation
      const double km = kc * ck[1];  // kmat = kcov/kdenom
      xm = xm + km * ck[0];    // xam = xbm + kmat*innov
      const double kb = ck[2] * km;  // beta*kmat
#pragma unroll
      for (int c = 0; c < 2 * NCH; ++c) x[c] = __builtin_fma(-kb, y[c], x[c]);
    }

    if (live) {
      store_row<NCH, VEC>(a.Xout + (size_t)row * M, M, q, x);
      if (q == 0) a.xout[row] = xm;
    }
  }
}

// ---------------------------------------------------------------------------
// Diag: the batch's own nb obs rows, serial in k.  One workgroup, quad per row.
// LDS: ye_cur[2][S], mye[2], tw[nb][nb] (GC only).
// ---------------------------------------------------------------------------
template <int NCH, bool VEC>
__global__ __launch_bounds__(kThreads) void k_diag(const DiagArgs a) {
  extern __shared__ __align__(16) double smem[];
  constexpr int S = 8 * NCH;
  double* ye_cur = smem;            // [2][S]
  double* mye_s = smem + 2 * S;     // [2] (+2 pad)
  double* tw = smem + 2 * S + 4;    // [nb][nb]
  const int tid = threadIdx.x;
  const int q = tid & 3;
  const int r = tid >> 2;
  const int M = a.M;
  const int nb = a.nb;
  const bool live = r < nb;
  const long row = a.b0 + r;
  const double rM1 = 1.0 / (double)(M - 1);
  const double invM = (double)M;

  if (a.loc_mode != 0) {
    // taper of ob k (row index k) against ob j of the batch: observation.py:68-83
    for (int i = tid; i < nb * nb; i += kThreads) {
      const int k = i / nb, j = i - k * nb;
      const double d = haversine_km(a.ob_lat[a.b0 + k], a.ob_lon[a.b0 + k], a.ob_lat[a.b0 + j],
                                    a.ob_lon[a.b0 + j]);
      tw[i] = gaspari_cohn(d, a.ob_hw[a.b0 + k]);
    }
  }

  double x[2 * NCH];
  double xm = 0.0;
  if (live) {
    load_row<NCH, VEC>(a.Yp + (size_t)row * M, M, q, x);
    xm = a.ym[row];
  } else {
#pragma unroll
    for (int c = 0; c < 2 * NCH; ++c) x[c] = 0.0;
  }
  // validity mask of this lane's slots (padding slots must not enter np.var)
  bool valid[2 * NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    valid[2 * c] = (8 * c + 2 * q) < M;
    valid[2 * c + 1] = (8 * c + 2 * q + 1) < M;
  }

  for (int k = 0; k < nb; ++k) {
    double* yb = ye_cur + (k & 1) * S;
    if (r == k) {
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        *reinterpret_cast<double2*>(yb + 8 * c + 2 * q) = make_double2(x[2 * c], x[2 * c + 1]);
      if (q == 0) mye_s[k & 1] = xm;
    }
    __syncthreads();
    double y[2 * NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const double2 v = *reinterpret_cast<const double2*>(yb + 8 * c + 2 * q);
      y[2 * c] = v.x;
      y[2 * c + 1] = v.y;
    }
    const double mye = mye_s[k & 1];
    // varye = np.var(ye): two-pass, ddof = 0 (ensrf.py:69)
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 2 * NCH; ++c) s += y[c];
    const double ymean = quad_sum(s) / invM;
    double ss = 0.0;
#pragma unroll
    for (int c = 0; c < 2 * NCH; ++c) {
      const double d = valid[c] ? (y[c] - ymean) : 0.0;
      ss = __builtin_fma(d, d, ss);
    }
    const double varye = quad_sum(ss) / invM;
    const long ob = a.b0 + k;
    const bool owner = (r == k);
    if (owner) {
      // Ye_rec row + prior diagnostics (ensrf.py:66,70)
      double* yr = a.Ye_rec + (size_t)ob * M;
      store_row<NCH, VEC>(yr, M, q, y);
      if (q == 0) {
        a.prior_mean[ob] = mye;
        a.prior_var[ob] = varye;
      }
    }
    const bool assim = a.ob_assim[ob] != 0;
    if (!assim) {  // ensrf.py:74-76 (uniform branch)
      if (owner && q == 0) {
        a.assimilated[ob] = 0;
        double* ck = a.coef + (size_t)ob * kCoefStride;
        ck[0] = 0.0; ck[1] = 0.0; ck[2] = 0.0; ck[3] = 0.0;
      }
      continue;
    }
    const double obs_err = a.ob_error[ob];
    const double innov = a.ob_value[ob] - mye;       // :85
    const double kdenom = varye + obs_err;           // :91
    const double rden = 1.0 / kdenom;
    const double beta = 1.0 / (1.0 + sqrt(obs_err / (varye + obs_err)));  // :135
    if (owner && q == 0) {
      double* ck = a.coef + (size_t)ob * kCoefStride;
      ck[0] = innov; ck[1] = rden; ck[2] = beta; ck[3] = 1.0;
    }
    // update every row of the batch (its own row included, taper(k,k) = 1)
    const double dot = quad_dot<NCH>(x, y);
    double kc = dot * rM1;
    if (a.loc_mode != 0) kc = (live ? tw[k * nb + r] : 0.0) * kc;
    const double km = kc * rden;
    xm = xm + km * innov;
    const double kb = beta * km;
#pragma unroll
    for (int c = 0; c < 2 * NCH; ++c) x[c] = __builtin_fma(-kb, y[c], x[c]);
    if (owner) {
      // posterior diagnostics of ob k (ensrf.py:144-149)
      double s2 = 0.0;
#pragma unroll
      for (int c = 0; c < 2 * NCH; ++c) s2 += x[c];
      const double pm = quad_sum(s2) / invM;
      double ss2 = 0.0;
#pragma unroll
      for (int c = 0; c < 2 * NCH; ++c) {
        const double d = valid[c] ? (x[c] - pm) : 0.0;
        ss2 = __builtin_fma(d, d, ss2);
      }
      const double pv = quad_sum(ss2) / invM;
      if (q == 0) {
        a.post_mean[ob] = xm;
        a.post_var[ob] = pv;
        a.assimilated[ob] = 1;
      }
    }
  }

  if (live) {
    store_row<NCH, VEC>(a.Yp + (size_t)row * M, M, q, x);
    if (q == 0) a.ym[row] = xm;
  }
}

template <int NCH>
hipError_t sweep_nch(const SweepArgs& a, bool vec, hipStream_t s) {
  const long nblocks = (a.nrows + kRowsPerBlock - 1) / kRowsPerBlock;
  const size_t lds = ((size_t)a.nb * 8 * NCH + (size_t)a.nb * kCoefStride) * sizeof(double);
  long grid = nblocks < 256L * 8 ? nblocks : 256L * 8;
  if (grid < 1) grid = 1;
  if (vec)
    hipLaunchKernelGGL((k_sweep<NCH, true>), dim3((unsigned)grid), dim3(kThreads), lds, s, a);
  else
    hipLaunchKernelGGL((k_sweep<NCH, false>), dim3((unsigned)grid), dim3(kThreads), lds, s, a);
  return hipGetLastError();
}

template <int NCH>
hipError_t diag_nch(const DiagArgs& a, bool vec, hipStream_t s) {
  const size_t lds = (2 * 8 * NCH + 4 + (a.loc_mode ? (size_t)a.nb * a.nb : 0)) * sizeof(double);
  if (vec)
    hipLaunchKernelGGL((k_diag<NCH, true>), dim3(1), dim3(kThreads), lds, s, a);
  else
    hipLaunchKernelGGL((k_diag<NCH, false>), dim3(1), dim3(kThreads), lds, s, a);
  return hipGetLastError();
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

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

static int round_nch(int M) {
  int nch = (M + 7) / 8;
  if (nch > 16) nch = (nch <= 20) ? 20 : (nch <= 24) ? 24 : 32;
  return nch;
}

hipError_t launch_sweep(const SweepArgs& a, hipStream_t s) {
  if (a.M < 2 || a.M > kMaxMembers || a.nb < 1 || a.nb > kMaxBatch) return hipErrorInvalidValue;
  if (a.nrows <= 0) return hipSuccess;
  const bool vec = (a.M % 2 == 0) && aligned16(a.Xin) && aligned16(a.Xout);
  const int nch = round_nch(a.M);
  EFA_NCH_SWITCH(sweep_nch, a, vec, s)
}

hipError_t launch_diag(const DiagArgs& a, hipStream_t s) {
  if (a.M < 2 || a.M > kMaxMembers || a.nb < 1 || a.nb > kMaxBatch) return hipErrorInvalidValue;
  const bool vec = (a.M % 2 == 0) && aligned16(a.Yp) && aligned16(a.Ye_rec);
  const int nch = round_nch(a.M);
  EFA_NCH_SWITCH(diag_nch, a, vec, s)
}

}  // namespace efa
