// Persistent Phase-A pipeline: the whole serial obs-space loop (ensrf.py:50-149
// restricted to the P obs rows of the augmented state) in ONE launch.
//
// Why: the loop is serial in k (ob k+1's prior depends on ob k's update), so its time is
// P x (latency of one step).  Launching a diag + a sweep kernel per 64-ob batch put two
// kernel boundaries, a 53 KB LDS re-staging and a full re-read of the obs block on that
// chain for every batch.  Here every workgroup keeps its 64 obs rows in registers for the
// entire loop and the steps are chained inside the launch:
//
//  - workgroup b owns rows [64b, 64b+64) (8 lanes per row, 512 threads).  It is the LEADER
//    for obs 64b..64b+63 and a FOLLOWER for every other ob;
//  - the leader's group that owns ob k publishes ye_k and the scalar gain factors
//    (a) into the workgroup's LDS ring -- its own waves continue after one barrier --
//    (b) into the global trajectory record traj[k] with agent-scope 8-byte stores;
//  - a follower's wave 0 keeps kPrefetch records in flight (agent-scope 8-byte loads,
//    compiler-tracked so no wait sits on the loop), validates every element against the
//    sentinel the record was pre-filled with, re-polls until complete, and hands the row
//    to its workgroup through the LDS ring;
//  - no flags, fences or ordering between elements are needed: each 8-byte granule is
//    self-validating (MI355X_MICROARCH.md "R2 granule"), loads/stores are agent scope
//    (per-XCD L2s are not coherent for plain accesses);
//  - all workgroups are co-resident (grid <= 256 CUs, one 512-thread workgroup each);
//    every spin is bounded and a global abort word makes all workgroups leave; the host
//    then falls back to the per-batch kernels (the obs block is only written at the end).
//
// The trajectory records double as the Phase-B input (ye rows + coefficients).
#include "efa_device.h"
#include "efa_internal.h"
#include "efa_rows.h"

namespace efa {
namespace {

constexpr int kPT = 512;      // threads per workgroup
constexpr int PL = 8;         // lanes per row
constexpr int kRing = 4;      // LDS ring slots
constexpr int kPrefetch = 4;  // trajectory records in flight per follower

typedef unsigned long long u64;

__device__ __forceinline__ u64 traj_load(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void traj_store(u64* p, double v) {
  __hip_atomic_store(p, (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// workgroup barrier that waits for LDS traffic only: outstanding global loads (prefetch) and
// stores (publication) must stay in flight across it (__syncthreads() would add vmcnt(0))
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int NC>
__global__ __launch_bounds__(kPT) void k_pipe(const PipeArgs a) {
  constexpr int PAD = 16 * NC;          // ye slots of a record
  constexpr int TS = PAD + kTrajScalars;
  constexpr int EPL = (TS + 63) / 64;   // record elements per lane of the loader wave
  __shared__ __align__(16) double ring[kRing * TS];
  __shared__ int bail;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int j = tid & (PL - 1);
  const int r = tid / PL;  // 0..63
  const int M = a.M;
  const long P = a.P, R = a.R;
  const long own0 = (long)blockIdx.x * kPipeRowsPerWG;
  const long own1 = (own0 + kPipeRowsPerWG < P) ? own0 + kPipeRowsPerWG : P;  // owned obs [own0, own1)
  const long row = own0 + r;
  const bool live = row < R;
  const bool is_ob = row < P;
  const double rM1 = 1.0 / (double)(M - 1);
  const double dM = (double)M;
  const double invM = 1.0 / dM;
  const bool vec = (M % 2 == 0);

  if (tid == 0) bail = 0;

  double x[2 * NC];
  double xm = 0.0, my_val = 0.0, my_err = 1.0;
  bool my_asm = false;
  if (live) {
    if (vec) load_row<PL, NC, true>(a.Yp + (size_t)row * M, M, j, x);
    else load_row<PL, NC, false>(a.Yp + (size_t)row * M, M, j, x);
    xm = a.ym[row];
    if (is_ob) {
      my_val = a.ob_value[row];
      my_err = a.ob_error[row];
      my_asm = a.ob_assim[row] != 0;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 2 * NC; ++c) x[c] = 0.0;
  }
  double rmean = group_rowsum<PL, NC>(x) / dM;

  double o_prior_mean = 0.0, o_prior_var = 0.0, o_innov = 0.0, o_rden = 0.0, o_beta = 0.0;
  double o_post_mean = 0.0, o_post_var = 0.0;
  bool o_done = false;

  // publish(k): by the 8 lanes that own row k, whose row is current through ob k-1
  auto publish = [&](long k) {
    double* slot = ring + (size_t)(k % kRing) * TS;
    u64* rec = a.traj + (size_t)k * TS;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int m0 = 2 * PL * c + 2 * j;
      *reinterpret_cast<double2*>(slot + m0) = make_double2(x[2 * c], x[2 * c + 1]);
      if (!(a.debug & 1)) {
        traj_store(rec + m0, x[2 * c]);
        traj_store(rec + m0 + 1, x[2 * c + 1]);
      }
    }
    // np.var, ddof=0 (ensrf.py:69); padding slots hold 0 and add mean^2 each: remove it
    const double ss = group_sumsq_about<PL, NC>(x, rmean) - (double)(PAD - M) * (rmean * rmean);
    const double varye = ss * invM;
    const double innov = my_val - xm;                       // :85
    const double kdenom = varye + my_err;                   // :91
    const double rden = 1.0 / kdenom;
    const double beta = 1.0 / (1.0 + sqrt(my_err * rden));  // :135
    double sv;  // lane j writes scalar j of the record
    switch (j) {
      case 0: sv = xm; break;
      case 1: sv = rmean; break;
      case 2: sv = innov; break;
      case 3: sv = rden; break;
      case 4: sv = beta; break;
      case 5: sv = my_asm ? 1.0 : 0.0; break;
      case 6: sv = varye; break;
      default: sv = 0.0; break;
    }
    slot[PAD + j] = sv;
    if (!(a.debug & 1)) traj_store(rec + PAD + j, sv);
    o_prior_mean = xm;    // :66
    o_prior_var = varye;  // :70
    o_innov = innov;
    o_rden = rden;
    o_beta = beta;
  };

  // ---- loader state (wave 0): kPrefetch records in flight ------------------------------
  u64 pend[kPrefetch][EPL];
  auto issue = [&](long k, u64 (&dst)[EPL]) {
    const long kk = (k < P) ? k : P - 1;  // clamp: harmless re-read at the tail
    const u64* rec = a.traj + (size_t)kk * TS;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const int idx = lane + 64 * e;
      dst[e] = traj_load(rec + (idx < TS ? idx : TS - 1));
    }
  };
  if (wave == 0) {
#pragma unroll
    for (int d = 0; d < kPrefetch; ++d) issue(d, pend[d]);
  }
  if (own0 == 0 && r == 0 && P > 0) publish(0);

  // GC taper of ob k against this lane's row, prefetched two obs ahead
  double wq0 = 1.0, wq1 = 1.0;
  const bool use_tw = (a.loc_mode != 0) && live;
  if (use_tw) {
    wq0 = a.tw[(size_t)0 * R + row];
    wq1 = (P > 1) ? a.tw[(size_t)1 * R + row] : 1.0;
  }

  long spins_left = a.spin_limit;
  for (long k = 0; k < P; ++k) {
    const bool mine = (k >= own0) && (k < own1);
    double* slot = ring + (size_t)(k % kRing) * TS;
    if (wave == 0) {
      u64 cur[EPL];
#pragma unroll
      for (int e = 0; e < EPL; ++e) cur[e] = pend[0][e];
#pragma unroll
      for (int d = 0; d + 1 < kPrefetch; ++d)
#pragma unroll
        for (int e = 0; e < EPL; ++e) pend[d][e] = pend[d + 1][e];
      if (!(a.debug & 2)) issue(k + kPrefetch, pend[kPrefetch - 1]);
      if (!mine) {
        bool ok = true;
#pragma unroll
        for (int e = 0; e < EPL; ++e) ok = ok && (cur[e] != kTrajSentinel);
        while (!__all(ok)) {  // not yet published: re-poll (bounded)
          if (--spins_left <= 0 || __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            if (lane == 0) {
              __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              bail = 1;
            }
            break;
          }
          __builtin_amdgcn_s_sleep(2);
          issue(k, cur);
          ok = true;
#pragma unroll
          for (int e = 0; e < EPL; ++e) ok = ok && (cur[e] != kTrajSentinel);
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          const int idx = lane + 64 * e;
          if (idx < TS) slot[idx] = __longlong_as_double((long long)cur[e]);
        }
      }
    }
    lds_barrier();
    if (bail) break;

    const bool active = slot[PAD + 5] != 0.0;  // uniform
    const double w = wq0;
    wq0 = wq1;
    if (use_tw) wq1 = a.tw[(size_t)((k + 2 < P) ? k + 2 : P - 1) * R + row];
    if (active) {
      double y[2 * NC];
      lds_read_row<PL, NC>(slot, j, y);
      const double dot = group_dot<PL, NC>(x, y);
      double kc = dot * rM1;                      // :95
      if (a.loc_mode != 0) kc = (live ? w : 0.0) * kc;  // :115
      const double km = kc * slot[PAD + 3];       // :119
      xm = xm + km * slot[PAD + 2];               // :130
      const double kb = slot[PAD + 4] * km;       // :136
      rmean = __builtin_fma(-kb, slot[PAD + 1], rmean);
#pragma unroll
      for (int c = 0; c < 2 * NC; ++c) x[c] = __builtin_fma(-kb, y[c], x[c]);  // :141
      if (mine && r == (int)(k - own0)) {
        const double f = 1.0 - kb;  // the ob's own row was scaled by (1 - kb)  (:144-149)
        o_post_var = (f * f) * o_prior_var;
        o_post_mean = xm;
        o_done = true;
      }
    }
    const long kn = k + 1;
    if (kn >= own0 && kn < own1 && r == (int)(kn - own0)) publish(kn);
  }

  if (bail) {
    if (tid == 0) a.status[1] = 1;
    return;  // nothing written back: the host re-runs Phase A with the per-batch kernels
  }
  if (live) {
    if (vec) store_row<PL, NC, true>(a.Yp + (size_t)row * M, M, j, x);
    else store_row<PL, NC, false>(a.Yp + (size_t)row * M, M, j, x);
    if (j == 0) {
      a.ym[row] = xm;
      if (is_ob) {
        a.prior_mean[row] = o_prior_mean;
        a.prior_var[row] = o_prior_var;
        double* ck = a.coef + (size_t)row * kCoefStride;
        ck[0] = my_asm ? o_innov : 0.0;
        ck[1] = my_asm ? o_rden : 0.0;
        ck[2] = my_asm ? o_beta : 0.0;
        ck[3] = my_asm ? 1.0 : 0.0;
        a.assimilated[row] = o_done ? 1 : 0;  // :74-76, :149
        if (o_done) {
          a.post_mean[row] = o_post_mean;
          a.post_var[row] = o_post_var;
        }
      }
    }
  }
}

__global__ void k_fill_u64(u64* p, size_t n, u64 v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// tw[k][row] = GC(haversine(ob_k, ob_row) / halfwidth_k): observation.py:68-83 for every pair
__global__ __launch_bounds__(256) void k_obs_taper_matrix(long P, long R, const double* __restrict__ lat,
                                                          const double* __restrict__ lon,
                                                          const double* __restrict__ hw, double* __restrict__ tw) {
  const size_t total = (size_t)P * R;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const long k = (long)(i / R);
    const long rr = (long)(i - (size_t)k * R);
    double w = 1.0;
    if (rr < P) w = gaspari_cohn(haversine_km(lat[k], lon[k], lat[rr], lon[rr]), hw[k]);
    tw[i] = w;
  }
}

template <int NC>
hipError_t pipe_launch(const PipeArgs& a, hipStream_t s) {
  const long grid = (a.R + kPipeRowsPerWG - 1) / kPipeRowsPerWG;
  hipLaunchKernelGGL((k_pipe<NC>), dim3((unsigned)grid), dim3(kPT), 0, s, a);
  return hipGetLastError();
}

}  // namespace

bool pipeline_supported(int M, long R) {
  return M >= 2 && M <= kMaxMembers && R > 0 && (R + kPipeRowsPerWG - 1) / kPipeRowsPerWG <= kPipeMaxWGs;
}

hipError_t launch_pipeline(const PipeArgs& a, hipStream_t s) {
  if (!pipeline_supported(a.M, a.R) || a.P <= 0) return hipErrorInvalidValue;
  switch ((a.M + 15) / 16) {
    case 1: return pipe_launch<1>(a, s);
    case 2: return pipe_launch<2>(a, s);
    case 3: return pipe_launch<3>(a, s);
    case 4: return pipe_launch<4>(a, s);
    case 5: return pipe_launch<5>(a, s);
    case 6: return pipe_launch<6>(a, s);
    case 7: return pipe_launch<7>(a, s);
    case 8: return pipe_launch<8>(a, s);
    case 9: return pipe_launch<9>(a, s);
    case 10: return pipe_launch<10>(a, s);
    case 11: return pipe_launch<11>(a, s);
    case 12: return pipe_launch<12>(a, s);
    case 13: return pipe_launch<13>(a, s);
    case 14: return pipe_launch<14>(a, s);
    case 15: return pipe_launch<15>(a, s);
    case 16: return pipe_launch<16>(a, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_fill_u64(unsigned long long* p, size_t n, unsigned long long v, hipStream_t s) {
  if (n == 0) return hipSuccess;
  size_t g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(k_fill_u64, dim3((unsigned)g), dim3(256), 0, s, p, n, v);
  return hipGetLastError();
}

hipError_t launch_obs_taper_matrix(long P, long R, const double* ob_lat, const double* ob_lon, const double* ob_hw,
                                   double* tw, hipStream_t s) {
  if (P <= 0 || R <= 0) return hipSuccess;
  size_t g = ((size_t)P * R + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  hipLaunchKernelGGL(k_obs_taper_matrix, dim3((unsigned)g), dim3(256), 0, s, P, R, ob_lat, ob_lon, ob_hw, tw);
  return hipGetLastError();
}

}  // namespace efa
