// Persistent Phase-A pipeline: the whole serial obs-space loop (ensrf.py:50-149
// restricted to the P obs rows of the augmented state) in ONE launch.
//
// Why: the loop is serial in k (ob k+1's prior depends on ob k's update), so its time is
// P x (latency of one step).  A diag + a sweep kernel per 64-ob batch puts two kernel
// boundaries, a 53 KB LDS re-staging and a full re-read of the obs block on that chain for
// every batch.  Here every workgroup keeps its 64 obs rows in registers for the entire loop
// and the steps are chained inside the launch.
//
// Roles.  Workgroup b owns rows [64b, 64b+64): 8 compute waves (8 lanes per row; consecutive
// obs live in DIFFERENT waves so the work of adjacent steps overlaps) + 1 loader wave.  The
// workgroup is the LEADER for its own 64 obs and a FOLLOWER for every other ob.
//
// Hand-off inside a workgroup: an LDS ring of records (ye row + 8 scalars) and two LDS
// counter, ready.  No workgroup barrier in the loop: waves spin on the
// counters (LDS operations of one wave are performed in order, so "write data, then write
// counter" / "read counter, then read data" needs no wait states beyond the poll itself).
// The owner starts the long scalar chain (rsq/rcp + Newton) in the same straight-line block as
// the independent row update, so both are finished together and ONE counter publishes the whole
// record: consumers fetch ye and the scalars in a single LDS round trip.  Ring slots are recycled only after every
// compute wave has reported (prog[w]) that it consumed the slot's previous record.
//
// Hand-off between workgroups: the leader's records are also written to global memory with
// agent-scope 8-byte stores; every element was pre-filled with a sentinel NaN payload, so a
// follower's loader wave validates each granule on its own -- no flag, fence or ordering
// (MI355X_MICROARCH.md "R2 granule"; agent scope because per-XCD L2s are not coherent).  The
// loader polls 4 records per round trip and is the only wave that waits on global memory;
// while its workgroup leads, the same wave forwards the finished records from the LDS ring
// to global memory, so publication costs the serial chain nothing.
//
// The leader's serial chain per step: poll -> ye from LDS -> dot -> 3-step DPP butterfly ->
// gain -> {FMA update -> publish ye} || {variance by a one-step recurrence -> rsq/rcp with
// Newton refinement -> publish scalars}.  The variance of row k+1 after ob k is
//     var' = var - 2 kb cov(y, ye_k) + kb^2 var_k,   cov = dot/M - mean(y) mean(ye_k)
// from a FRESH two-pass variance that the row's owner computed one step earlier, off the
// chain; if var' < 1% of var (cancellation) it is recomputed from the updated row.
//
// All workgroups are co-resident (grid <= 256, one 320-thread workgroup per CU, checked against the occupancy query
// before the launch); every
// spin is bounded; a global abort word releases everyone; the kernel then writes nothing
// back and the host re-runs Phase A with the per-batch kernels.
#include "efa_device.h"
#include "efa_internal.h"
#include "efa_rows.h"

namespace efa {
namespace {

constexpr int kCW = kPipeLanes;        // compute waves per workgroup (one per SIMD when 4)
constexpr int kPT = 64 * (kCW + 1);    // + one loader wave
constexpr int PL = kPipeLanes;         // lanes per row
constexpr int kRing = 32;              // LDS ring slots
constexpr int kGuardEvery = 8;         // the slot-recycling guard is evaluated once per this many records
constexpr int kPoll = 4;               // records fetched per loader round trip

typedef unsigned long long u64;

__device__ __forceinline__ u64 traj_load(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void traj_store(u64* p, double v) {
  __hip_atomic_store(p, (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// LDS control words: plain in-order LDS accesses + a compiler barrier (see header comment)
__device__ __forceinline__ int ctl_load_lane(const int* p) {  // per-lane address
  const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
  return v;
}
// every lane reads the same word: hand the compiler a wave-uniform (SGPR) value so that the
// spin / bail logic compiles to scalar branches instead of exec-mask bookkeeping
__device__ __forceinline__ int ctl_load(const int* p) { return __builtin_amdgcn_readfirstlane(ctl_load_lane(p)); }
// minimum over the 8 lanes of a row group (DPP, no LDS round trip), same value in all 8 lanes
__device__ __forceinline__ int group8_min(int v) {  // over the PL lanes of a row group
  v = min(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true));
  v = min(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true));
  if (PL >= 8) v = min(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true));
  return v;
}
__device__ __forceinline__ void ctl_store(int* p, int v) {
  asm volatile("" ::: "memory");
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// v_rsq_f64 / v_rcp_f64 are accurate to 2e-8 relative on gfx950 (measured, tools/latency_probe.hip):
// one Newton step squares that (rsq: 1.5 e^2, rcp: e^2) -- below double rounding; the second-order
// term of the rsq step is added so that both land within ~1 ulp.
__device__ __forceinline__ double fast_rsq(double a) {  // 1/sqrt(a)
  const double q = __builtin_amdgcn_rsq(a);
  const double e = __builtin_fma(-a * q, q, 1.0);           // 1 - a q^2
  const double p = __builtin_fma(0.375, e, 0.5);            // 1/2 + 3/8 e
  return __builtin_fma(q * e, p, q);                        // q (1 + e/2 + 3 e^2/8)
}
__device__ __forceinline__ double fast_rcp(double b) {  // 1/b
  const double r = __builtin_amdgcn_rcp(b);
  const double e = __builtin_fma(-b, r, 1.0);
  return __builtin_fma(r, __builtin_fma(e, e, e), r);        // r (1 + e + e^2)
}

enum { kReadyYe = 0, kBail = 1, kReadySc = 2, kFwd = 3, kProg = 4 };  // ctl[] indices; prog[w] = ctl[kProg+w]

template <int NC>
__global__ __launch_bounds__(kPT) void k_pipe(const PipeArgs a) {
  constexpr int PAD = 2 * PL * NC;  // ye slots of a record
  constexpr int TS = PAD + kTrajScalars;
  constexpr int EPL = (TS + 63) / 64;  // record elements per lane of the loader wave
  __shared__ __align__(16) double ring[kRing * TS];
  __shared__ int ctl[16];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int M = a.M;
  const long P = a.P, R = a.R;
  const long own0 = (long)blockIdx.x * kPipeRowsPerWG;
  const long own1 = (own0 + kPipeRowsPerWG < P) ? own0 + kPipeRowsPerWG : (own0 < P ? P : own0);

  if (tid < 16) ctl[tid] = (tid >= kProg) ? -1 : (tid == kFwd ? (int)(own0 - 1) : 0);
  __syncthreads();
  // spins are bounded by a poll budget AND by wall time (s_memrealtime, 100 MHz), the latter looked at on the slow
  // side of a poll loop only
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#define EFA_TIMED_OUT() (a.spin_ticks > 0 && (long)(__builtin_amdgcn_s_memrealtime() - t_start) > a.spin_ticks)

  // =====================================================================================
  // loader wave
  // =====================================================================================
  if (wave == kCW) {
    // Three phases, each its own loop: follow the records before the own block, forward the own block,
    // follow the rest.  (As ONE loop the compiler merged the poll path's pending-load state into the
    // forward path and waited vmcnt(0) per forwarded record -- i.e. for the previous record's
    // write-through store to complete, ~1400 cycles per ob, which bounded the whole kernel.)
    long spins_left = a.spin_limit;
    bool failed = false;
    auto follow = [&](long next, const long limit) {
      while (next < limit && !failed) {
        const int nrec = (int)((limit - next < kPoll) ? (limit - next) : kPoll);
        u64 v[kPoll][EPL];
#pragma unroll
        for (int d = 0; d < kPoll; ++d) {
          const long kk = next + ((d < nrec) ? d : nrec - 1);
          const u64* rec = a.traj + (size_t)kk * TS;
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            const int idx = lane + 64 * e;
            v[d][e] = traj_load(rec + (idx < TS ? idx : TS - 1));
          }
        }
        int cnt = 0;
#pragma unroll
        for (int d = 0; d < kPoll; ++d) {
          bool ok = true;
#pragma unroll
          for (int e = 0; e < EPL; ++e) ok = ok && (v[d][e] != kTrajSentinel);
          if (cnt == d && d < nrec && __all(ok)) cnt = d + 1;
        }
        if (cnt == 0) {
          if (--spins_left <= 0 || __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
              ((spins_left & 15) == 0 && EFA_TIMED_OUT()))
            failed = true;
          __builtin_amdgcn_s_sleep(2);
          continue;
        }
        // ring slots may be recycled only when every compute wave consumed their old record
        const long need = next + cnt - 1 - kRing;  // all prog[w] must be >= need
        if (need >= 0) {
          for (;;) {
            const int mn = __builtin_amdgcn_readfirstlane(group8_min(ctl_load_lane(&ctl[kProg + (lane & (kCW - 1))])));
            if (mn >= (int)need) break;
            if (--spins_left <= 0 || ((spins_left & 15) == 0 && EFA_TIMED_OUT())) {
              failed = true;
              break;
            }
          }
          if (failed) break;
        }
#pragma unroll
        for (int d = 0; d < kPoll; ++d) {
          if (d < cnt) {
            double* slot = ring + (size_t)((next + d) % kRing) * TS;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
              const int idx = lane + 64 * e;
              if (idx < TS) slot[idx] = __longlong_as_double((long long)v[d][e]);
            }
          }
        }
        next += cnt;
        if (lane == 0) ctl_store(&ctl[kReadyYe], (int)next);
      }
    };
    follow(0, (own0 < P) ? own0 : P);
    if (own0 < P && !failed) {
      // leader phase: the owners publish into the LDS ring only; this wave forwards each
      // finished record to global memory (agent-scope granules) for the other workgroups
      for (long f = own0; f < own1; ++f) {
        while (ctl_load(&ctl[kReadyYe]) <= (int)f) {
          if (--spins_left <= 0 || ctl_load(&ctl[kBail]) != 0 || ((spins_left & 15) == 0 && EFA_TIMED_OUT())) {
            failed = true;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (failed) break;
        const double* slot = ring + (size_t)(f % kRing) * TS;
        u64* rec = a.traj + (size_t)f * TS;
        if (!(a.debug & 1)) {
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            const int idx = lane + 64 * e;
            if (idx < TS) traj_store(rec + idx, slot[idx]);
          }
        }
        if (lane == 0) ctl_store(&ctl[kFwd], (int)f);
      }
      if (!failed) follow(own1, P);
    }
    if (failed && lane == 0) {
      __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.status + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ctl_store(&ctl[kBail], 1);
    }
    return;
  }

  // =====================================================================================
  // compute waves
  // =====================================================================================
  const int j = lane & (PL - 1);
  const int grp = lane / PL;
  const int i_loc = wave + kCW * grp;  // local row / ob index 0..63: consecutive obs in different waves
  const long row = own0 + i_loc;
  const bool live = row < R;
  const bool is_ob = row < P;
  const double rM1 = 1.0 / (double)(M - 1);
  const double dM = (double)M;
  const double invM = 1.0 / dM;
  const double padc = (double)(PAD - M);
  const bool vec = (M % 2 == 0);

  double x[2 * NC];
  double xm = 0.0, my_val = 0.0, my_err = 1.0, my_sqrt_err = 1.0;
  bool my_asm = false;
  if (live) {
    if (vec) load_row<PL, NC, true>(a.Yp + (size_t)row * M, M, j, x);
    else load_row<PL, NC, false>(a.Yp + (size_t)row * M, M, j, x);
    xm = a.ym[row];
    if (is_ob) {
      my_val = a.ob_value[row];
      my_err = a.ob_error[row];
      my_sqrt_err = sqrt(my_err);
      my_asm = a.ob_assim[row] != 0;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 2 * NC; ++c) x[c] = 0.0;
  }
  double rmean = group_rowsum<PL, NC>(x) / dM;
  // fresh two-pass variance of the row as it stands (np.var, ddof = 0, ensrf.py:69)
  auto fresh_var = [&]() {
    return (group_sumsq_about<PL, NC>(x, rmean) - padc * (rmean * rmean)) * invM;
  };
  double vfresh = fresh_var();

  double o_prior_mean = 0.0, o_prior_var = 0.0, o_innov = 0.0, o_rden = 0.0, o_beta = 0.0;
  double o_post_mean = 0.0, o_post_var = 0.0;
  bool o_done = false;
  long spins_left = a.spin_limit;
  bool bailed = false;
  auto give_up = [&]() {  // a bounded spin expired: release every workgroup, report to the host
    __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.status + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ctl_store(&ctl[kBail], 1);
  };

  // Scalar gain factors of this lane's row as a prospective observation (ensrf.py:85,91,135),
  // with q = 1/sqrt(kdenom): rden = q^2, beta = 1/(1 + sqrt(err/kdenom)) = 1/(1 + sqrt(err) q).
  // It is a ~25-deep dependent chain, so the chain wave starts it BEFORE the row update and the
  // LDS publication of ye: by the time the other waves have read ye and finished their dot
  // products the scalars are already in the ring.
  struct Gain {
    double innov, rden, beta, varye;
  };
  auto gain_of = [&](double mean_now, double varye) {
    Gain g;
    g.varye = varye;
    g.innov = my_val - mean_now;
    const double kdenom = varye + my_err;
    const double q = fast_rsq(kdenom);
    g.rden = q * q;
    g.beta = fast_rcp(1.0 + my_sqrt_err * q);
    return g;
  };
  // publication of record kn into the LDS ring: ye first (raises ready_ye), then the scalars
  auto write_record = [&](long kn, const Gain& g, bool pub) {
    double* slot = ring + (size_t)(kn % kRing) * TS;
    if (pub) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
        *reinterpret_cast<double2*>(slot + 2 * PL * c + 2 * j) = make_double2(x[2 * c], x[2 * c + 1]);
      if (j == 0) {
        double2* sc = reinterpret_cast<double2*>(slot + PAD);
        sc[0] = make_double2(xm, rmean);
        sc[1] = make_double2(g.innov, g.rden);
        sc[2] = make_double2(g.beta, my_asm ? 1.0 : 0.0);
        sc[3] = make_double2(g.varye, 0.0);
        ctl_store(&ctl[kReadyYe], (int)(kn + 1));  // ye AND scalars of record kn are in the ring
      }
      o_prior_mean = xm;         // :66
      o_prior_var = g.varye;     // :70
      o_innov = g.innov;
      o_rden = g.rden;
      o_beta = g.beta;
    }
  };
  // wave-uniform wait until the ring slot of record kn may be recycled: every compute wave
  // consumed, and the forwarder forwarded, record kn - kRing
  auto wait_slot_free = [&](long kn) {
    for (;;) {
      const int mn = __builtin_amdgcn_readfirstlane(
          group8_min(min(ctl_load_lane(&ctl[kProg + j]), ctl_load_lane(&ctl[kFwd]))));
      if (mn >= (int)(kn - kRing)) return;
      if (ctl_load(&ctl[kBail]) != 0) {
        bailed = true;
        return;
      }
      if (--spins_left <= 0 || ((spins_left & 15) == 0 && EFA_TIMED_OUT())) {
        give_up();
        bailed = true;
        return;
      }
    }
  };

  if (own0 == 0 && wave == 0 && P > 0) write_record(0, gain_of(xm, vfresh), i_loc == 0);

  // GC taper of ob k against this lane's row, prefetched two obs ahead
  double wq0 = 1.0, wq1 = 1.0;
  const bool use_tw = (a.loc_mode != 0) && live;
  if (use_tw) {
    wq0 = a.tw[(size_t)0 * R + row];
    wq1 = (P > 1) ? a.tw[(size_t)1 * R + row] : 1.0;
  }

  // cycle stamps for tools/pipe_stamps.py: diagnostic builds only (make STAMPS=1), see efa_pipeline_gram.hip
#ifdef EFA_PIPE_STAMPS
#define EFA_STAMP(i)                                                                     \
  do {                                                                                   \
    if (a.dbg != nullptr && pub && j == 0) a.dbg[(size_t)k * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define EFA_STAMP(i) \
  do {                \
  } while (0)
#endif
  for (long k = 0; k < P && !bailed; ++k) {
    const double* slot = ring + (size_t)(k % kRing) * TS;
    const long kn = k + 1, k2 = k + 2;
    // wave-uniform roles: the wave holding the owner of ob k+1 carries this step's serial chain;
    // the wave holding the owner of ob k+2 refreshes that row's variance off the chain
    const bool chain_wave = (kn >= own0) && (kn < own1) && ((int)((kn - own0) & (kCW - 1)) == wave);
    const bool fresh_wave = (k2 >= own0) && (k2 < own1) && ((int)((k2 - own0) & (kCW - 1)) == wave);
    const bool pub = chain_wave && (i_loc == (int)(kn - own0));
    if (chain_wave) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(0);
    EFA_STAMP(0);
    {  // wait for record k: the chain wave polls back to back, the others doze between polls;
       // the bail word and the spin budget are looked at once per 16 polls
      int polls = 0;
      while (ctl_load(&ctl[kReadyYe]) <= (int)k) {
        if ((++polls & 15) == 0) {
          if (ctl_load(&ctl[kBail]) != 0) {
            bailed = true;
            break;
          }
          spins_left -= 16;
          if (spins_left <= 0 || EFA_TIMED_OUT()) {
            give_up();
            bailed = true;
            break;
          }
        }
        if (!chain_wave) __builtin_amdgcn_s_sleep(1);
      }
    }
    if (bailed) break;
    EFA_STAMP(1);
    double y[2 * NC];
    lds_read_row<PL, NC>(slot, j, y);
    const double2 s01 = *reinterpret_cast<const double2*>(slot + PAD);      // mye, mean(ye)
    const double2 s23 = *reinterpret_cast<const double2*>(slot + PAD + 2);  // innov, rden
    const double2 s45 = *reinterpret_cast<const double2*>(slot + PAD + 4);  // beta, active
    const double var_k = slot[PAD + 6];
    const double w = wq0;
    wq0 = wq1;
    if (use_tw) wq1 = a.tw[(size_t)((k + 2 < P) ? k + 2 : P - 1) * R + row];
    const double dot = group_dot<PL, NC>(x, y);
    EFA_STAMP(2);
    if (lane == 0) ctl_store(&ctl[kProg + wave], (int)k);  // record k consumed by this wave
    EFA_STAMP(3);
    const bool active = __builtin_amdgcn_readfirstlane((int)(s45.y != 0.0)) != 0;  // wave-uniform
    const bool own_k = (k >= own0) && (k < own1) && (i_loc == (int)(k - own0));
    if (chain_wave) {
      // ---- the serial chain of this step: ONE straight-line block so that the scheduler can
      // interleave the long scalar chain (rsq/rcp + Newton) with the independent row update
      double kc = dot * rM1;                              // :95
      if (a.loc_mode != 0) kc = (live ? w : 0.0) * kc;    // :115
      const double km = active ? kc * s23.y : 0.0;        // :119 (not assimilated: no update)
      const double kb = s45.x * km;                       // :136
      const double cov = __builtin_fma(dot, invM, -(rmean * s01.y));
      const double var_rec = __builtin_fma(kb * kb, var_k, __builtin_fma(-2.0 * kb, cov, vfresh));
      const double var_next = active ? var_rec : vfresh;  // one-step recurrence (see header)
      xm = xm + km * s23.x;                               // :130
      Gain g;
      if (!__any(pub && active && !(var_next > 0.01 * vfresh))) {
        // fast path, one basic block: the gain chain is issued first and the independent
        // row update fills its latency
        g = gain_of(xm, var_next);
        rmean = __builtin_fma(-kb, s01.y, rmean);
#pragma unroll
        for (int c = 0; c < 2 * NC; ++c) x[c] = __builtin_fma(-kb, y[c], x[c]);  // :141
      } else {
        // cancellation guard (rare): the recurrence lost digits, take the variance of the updated row
        rmean = __builtin_fma(-kb, s01.y, rmean);
#pragma unroll
        for (int c = 0; c < 2 * NC; ++c) x[c] = __builtin_fma(-kb, y[c], x[c]);
        g = gain_of(xm, fresh_var());
      }
      EFA_STAMP(4);
      // recycling guard, amortised: once per kGuardEvery records, for that many records ahead
      if ((kn & (kGuardEvery - 1)) == 0 && kn + kGuardEvery > kRing) wait_slot_free(kn + kGuardEvery - 1);
      if (bailed) break;
      write_record(kn, g, pub);
    } else if (active) {
      double kc = dot * rM1;                              // :95
      if (a.loc_mode != 0) kc = (live ? w : 0.0) * kc;    // :115
      const double km = kc * s23.y;                       // :119
      xm = xm + km * s23.x;                               // :130
      const double kb = s45.x * km;                       // :136
      rmean = __builtin_fma(-kb, s01.y, rmean);
#pragma unroll
      for (int c = 0; c < 2 * NC; ++c) x[c] = __builtin_fma(-kb, y[c], x[c]);  // :141
      if (own_k) {
        const double f = 1.0 - kb;  // the ob's own row was scaled by (1 - kb)  (:144-149)
        o_post_var = (f * f) * o_prior_var;
        o_post_mean = xm;
        o_done = true;
      }
    }
    EFA_STAMP(5);
    if (fresh_wave) vfresh = fresh_var();  // off the chain (uniform over the wave)
    EFA_STAMP(6);
  }
#undef EFA_STAMP
  __builtin_amdgcn_s_setprio(0);

  if (bailed) return;  // nothing written back: the host re-runs Phase A with the per-batch kernels
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

// per ob: cos/sin of latitude and longitude, and s_lim = sin^2(|halfwidth| / R): the haversine argument at which the
// taper reaches 0 (see efa_gcsweep.hip: the same trig-free rejection, here for the P x P obs-obs pairs)
constexpr int kObTrigP = 6;
__global__ void k_obs_trig(long P, const double* __restrict__ lat, const double* __restrict__ lon,
                           const double* __restrict__ hw, double* __restrict__ tab) {
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= P) return;
  const double pl = radians(lat[k]), po = radians(lon[k]);
  const double ang = fabs(hw[k]) / kEarthRadiusKm;
  double slim = sin(ang);
  slim = slim * slim;
  if (!(ang < 1.5)) slim = 4.0;  // cut-off beyond a quarter of the globe (or a NaN radius): nothing is rejected cheaply
  double* t = tab + k * kObTrigP;
  t[0] = cos(pl);
  t[1] = sin(pl);
  t[2] = cos(po);
  t[3] = sin(po);
  t[4] = slim;
  t[5] = 0.0;
}

// tw[k][row] = GC(haversine(ob_k, ob_row) / halfwidth_k): observation.py:68-83 for every pair.  Pairs clearly beyond
// 2 halfwidths (trig-free haversine argument against s_lim, relative margin 1e-6) are 0 without evaluating anything else;
// every other pair goes through the reference's formula.
// Round 3: the survivors are a few per cent of the pairs but sit in four waves out of five, so evaluating them where they
// are found made nearly every wave run the trigonometry for a handful of lanes (1.7 ms at 1e4 obs).  A wave now owns 64
// columns for kTwRows rows: it writes the zeros (and the ones of the carried rows) at once, collects the survivors in an LDS
// list (ballot + prefix count) and evaluates them afterwards 64 at a time.
constexpr int kTwRows = 32;
__global__ __launch_bounds__(256) void k_obs_taper_matrix(long P, long R, const double* __restrict__ lat,
                                                          const double* __restrict__ lon,
                                                          const double* __restrict__ hw, const double* __restrict__ tab,
                                                          double* __restrict__ tw) {
  __shared__ unsigned int list[4][kTwRows * 64];  // per wave: (row in the chunk) << 6 | lane
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long rr = ((long)blockIdx.x * 4 + wave) * 64 + lane;  // this lane's column
  const bool col_ok = rr < R, is_ob = rr < P;
  double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0, latr = 0.0, lonr = 0.0;
  if (is_ob) {
    const double* tr = tab + rr * kObTrigP;
    c0 = tr[0];
    c1 = tr[1];
    c2 = tr[2];
    c3 = tr[3];
    latr = lat[rr];
    lonr = lon[rr];
  }
  for (long k0 = (long)blockIdx.y * kTwRows; k0 < P; k0 += (long)gridDim.y * kTwRows) {
    int count = 0;  // wave-uniform
    for (int i = 0; i < kTwRows; ++i) {
      const long k = k0 + i;
      if (k >= P) break;
      const double* tk = tab + k * kObTrigP;   // wave-uniform loads
      const double cc = tk[0] * c0;
      const double h = 0.5 * (1.0 - (cc + tk[1] * c1)) + cc * (0.5 * (1.0 - (tk[2] * c2 + tk[3] * c3)));
      const bool near = is_ob && !(h > tk[4] * (1.0 + 1e-6) + 1e-13);
      const unsigned long long m = __ballot(near);
      if (col_ok && !near) tw[(size_t)k * R + rr] = is_ob ? 0.0 : 1.0;
      if (near) list[wave][count + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = ((unsigned)i << 6) | (unsigned)lane;
      count += __builtin_popcountll(m);
    }
    // (LDS operations of one wave are performed in order: no barrier between the list's writes and reads)
    for (int e = lane; e < count; e += 64) {
      const unsigned int v = list[wave][e];
      const long k = k0 + (v >> 6);
      const long col = rr - lane + (v & 63);
      tw[(size_t)k * R + col] = gaspari_cohn(haversine_km(lat[k], lon[k], lat[col], lon[col]), hw[k]);
    }
  }
  (void)latr;
  (void)lonr;
}

template <int NC>
hipError_t pipe_launch(const PipeArgs& a, hipStream_t s) {
  const long grid = (a.R + kPipeRowsPerWG - 1) / kPipeRowsPerWG;
  // every workgroup waits for records of every other: the grid must fit the device at once
  int per_cu = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&k_pipe<NC>), kPT, 0);
  if (e != hipSuccess) return e;
  if (grid > (long)per_cu * a.cu_count) return hipErrorCooperativeLaunchTooLarge;
  hipLaunchKernelGGL((k_pipe<NC>), dim3((unsigned)grid), dim3(kPT), 0, s, a);
  return hipGetLastError();
}

}  // namespace

bool pipeline_supported(int M, long R) {
  return M >= 2 && M <= 16 * 2 * kPipeLanes && M <= 128 && R > 0 && (R + kPipeRowsPerWG - 1) / kPipeRowsPerWG <= kPipeMaxWGs;
}

hipError_t launch_pipeline(const PipeArgs& a, hipStream_t s) {
  if (!pipeline_supported(a.M, a.R) || a.P <= 0) return hipErrorInvalidValue;
  switch ((a.M + 2 * PL - 1) / (2 * PL)) {
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
                                   double* trig_scratch, double* tw, hipStream_t s) {
  if (P <= 0 || R <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_obs_trig, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, P, ob_lat, ob_lon, ob_hw, trig_scratch);
  const long gx = (R + 255) / 256;                                   // 4 waves x 64 columns per workgroup
  long gy = (P + kTwRows - 1) / kTwRows;
  if (gy > 4096) gy = 4096;
  hipLaunchKernelGGL(k_obs_taper_matrix, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, s, P, R, ob_lat, ob_lon, ob_hw, trig_scratch, tw);
  return hipGetLastError();
}

}  // namespace efa
