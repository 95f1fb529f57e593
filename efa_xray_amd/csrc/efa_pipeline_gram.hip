// Persistent Phase-A pipeline, Gram-space leader ("k_pipe_gram").
//
// Same launch structure and inter-workgroup protocol as efa_pipeline.hip (one workgroup per 64
// obs rows, rows resident in registers, sentinel-validated agent-scope trajectory records), but
// the LEADER no longer carries vectors on its serial chain.  In efa_pipeline.hip every step of
// the chain contains an M-long LDS read, a dot product with a cross-lane reduction, the gain
// scalars and an M-long LDS write.  Here, when a workgroup's turn comes:
//
//   1. its 64 rows are written once to an LDS tile and G = Y Y^T (the 64 x 64 matrix of their
//      dot products) is formed on the matrix cores (v_mfma_f64_16x16x4_f64, ~3 us);
//      the two helper waves keep the pending rows of G in registers;
//   2. ONE "pivot" wave runs the serial recurrence in Gram space, lane j = row j: for ob k
//         var_k   = G_kk/M - mean_k^2                       (np.var, ddof 0: ensrf.py:69)
//         kmat_j  = w_jk * G_kj/(M-1) / kdenom_k            (:95,:115,:119) for all 64 rows at once
//         kb_j    = beta_k * kmat_j                         (:136)
//         G_ij   -= kb_j G_ki + kb_i (G_kj - kb_j G_kk)     (what y_i -= kb_i ye_k does to the dots)
//      no dot products, no reductions, no vectors: lane k's values are fetched with v_readlane.
//      Two helper waves apply the rank-one downdate to the rows that become pivots later;
//   3. the four vector waves follow behind: with kb_i known, a step is a pure axpy
//      y_i -= kb_i ye_k, and ye_k itself is just row k after its own axpys.  For the block the rows
//      sit in matrix-core accumulator tiles: the next rows to be published are updated step by
//      step, all others by one rank-4 MFMA per tile and four steps;
//   4. the loader wave forwards (ye_k from the vector ring, scalars from the pivot wave) to
//      global memory for the other workgroups, which consume them exactly as before.
//
// The dot products of a block are thus taken from ONE fresh Gram matrix per 64 obs and downdated
// in between; if a pivot's G_kk has shrunk below 1e-3 of its value at the start of the block
// (the downdate may then have lost more than three digits) the kernel bails out through the
// usual abort word and the host re-runs Phase A with efa_pipeline.hip.
#include <type_traits>

#include "efa_device.h"
#include "efa_internal.h"
#include "efa_rows.h"

namespace efa {
namespace {

typedef unsigned long long u64;
typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int kVW = 4;        // vector waves (quad per row, one per SIMD)
constexpr int kGT = 512;      // threads: 4 vector + pivot + 2 helpers + loader
constexpr int PLg = kPipeLanes;  // lanes per row: same record layout as efa_pipeline.hip
constexpr int kRingG = 16;    // LDS ring slots for ye rows
constexpr int kPollG = 4;
constexpr int kRowsWG = kPipeRowsPerWG;  // 64
static_assert(PLg == 4 && kRowsWG == 64, "layout assumptions of the Gram kernel");

enum { cReady = 0, cBail = 1, cSReady = 2, cFwd = 3, cProg = 4, cHProg = 8 };
constexpr int kScStride = 8;  // doubles per step in the pivot's scalar records

__device__ __forceinline__ u64 g_traj_load(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void g_traj_store(u64* p, double v) {
  __hip_atomic_store(p, (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int g_ctl_lane(const int* p) {
  const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
  return v;
}
__device__ __forceinline__ int g_ctl(const int* p) { return __builtin_amdgcn_readfirstlane(g_ctl_lane(p)); }
__device__ __forceinline__ void g_ctl_set(int* p, int v) {
  asm volatile("" ::: "memory");
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ double g_rsq(double a) {  // see efa_pipeline.hip: one Newton step suffices
  const double q = __builtin_amdgcn_rsq(a);
  const double e = __builtin_fma(-a * q, q, 1.0);
  const double p = __builtin_fma(0.375, e, 0.5);
  return __builtin_fma(q * e, p, q);
}
__device__ __forceinline__ double g_rcp(double b) {
  const double r = __builtin_amdgcn_rcp(b);
  const double e = __builtin_fma(-b, r, 1.0);
  return __builtin_fma(r, __builtin_fma(e, e, e), r);
}
__device__ __forceinline__ double rl(double v, int lane) {  // value held by `lane` (wave-uniform index)
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

template <int NC>
struct GramShape {
  static constexpr int PAD = 2 * PLg * NC;
  static constexpr int TS = PAD + kTrajScalars;
  static constexpr int SP = PAD + ((2 - PAD % 32) + 32) % 32;  // tile row stride == 2 (mod 32): conflict-free operand reads
  static constexpr int UREG = (kRowsWG * SP > kRowsWG * (2 * kRowsWG + 8)) ? kRowsWG * SP : kRowsWG * (2 * kRowsWG + 8);
  static size_t lds_bytes(bool gc) {
    return ((size_t)kRingG * TS + kRowsWG * kRowsWG + UREG + 5 * kRowsWG + (gc ? kRowsWG * kRowsWG : 0)) * sizeof(double) +
           16 * sizeof(int);
  }
};

template <int NC>
__global__ __launch_bounds__(kGT) void k_pipe_gram(const PipeArgs a) {
  using Sh = GramShape<NC>;
  constexpr int PAD = Sh::PAD, TS = Sh::TS, SP = Sh::SP, UREG = Sh::UREG;
  constexpr int EPL = (TS + 63) / 64;
  constexpr int NJ = (PAD + 15) / 16;  // accumulator tiles per wave in the block's matrix-core layout
  extern __shared__ __align__(16) double lds[];
  double* ring = lds;                          // [kRingG][TS]   ye rows (+ scalars in follower mode)
  double* G_s = ring + kRingG * TS;            // [64][64]       Gram matrix of the block
  double* U = G_s + kRowsWG * kRowsWG;         // union: Yt[64][SP]  then  {g, kb}[64][64], sc[64][4]
  double* pm = U + UREG;                       // [2][64]        parked row means / obs-space means
  double* pv = pm + 2 * kRowsWG;               // [3][64]        ob value / error / sqrt(error) of the block
  double* tw_s = pv + 3 * kRowsWG;             // [64][64]       taper corner (GC only)
  int* ctl = reinterpret_cast<int*>(tw_s + (a.loc_mode != 0 ? kRowsWG * kRowsWG : 0));  // [16]
  double* Yt = U;
  // per-step records of the pivot wave (every LDS instruction on that wave costs the chain ~30 cycles of
  // issue, so a step publishes as little as possible): one b128 per lane and two by lane 0
  double2* s_gk = reinterpret_cast<double2*>(U);  // [step][row] = {G_kj, kb_j}
  double* s_sc = U + 2 * kRowsWG * kRowsWG;       // [step][8]   = innov, rden, beta, active, prior mean, prior var

  const int tid = threadIdx.x;
  // wave roles: 0-3 vector, 4 pivot, 5-6 helpers, 7 loader (pairing the pivot with the loader on one SIMD
  // instead, by permuting the roles, measured 3% slower)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int M = a.M;
  const long P = a.P, R = a.R;
  const long own0 = (long)blockIdx.x * kRowsWG;
  const long own1 = (own0 + kRowsWG < P) ? own0 + kRowsWG : (own0 < P ? P : own0);
  const int nb = (int)(own1 - own0);  // obs this workgroup leads (0: it only follows)
  const bool leads = nb > 0;
  const double rM1 = 1.0 / (double)(M - 1);
  const double invM = 1.0 / (double)M;

  if (tid < 16) ctl[tid] = (tid >= cProg) ? -1 : (tid == cFwd ? (int)(own0 - 1) : 0);
  __syncthreads();

  // Cycle stamps for tools/gram_stamps*.py exist only in diagnostic builds (make STAMPS=1): an s_memtime
  // anywhere in a loop makes the compiler wait for ALL outstanding LDS traffic (lgkmcnt(0)) wherever it
  // waits at all, because scalar-memory results may return out of order with LDS results.
#ifdef EFA_PIPE_BLOCKTIME  /* make EXTRA=-DEFA_PIPE_BLOCKTIME: three stamps per block, none inside a loop */
#define EFA_BLOCKSTAMP(cond, slot)                                                                  \
  do {                                                                                              \
    if (a.dbg != nullptr && (cond)) a.dbg[(size_t)own0 * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define EFA_BLOCKSTAMP(cond, slot) \
  do {                              \
  } while (0)
#endif
#if defined(EFA_PIPE_STAMPS) || defined(EFA_PIPE_BLOCKTIME)
#define EFA_EXP(bit) ((a.debug & (bit)) != 0)  /* timing experiments of tools/gram_exp*.py: results are wrong */
#else
#define EFA_EXP(bit) false
#endif
#ifdef EFA_PIPE_STAMPS
#define EFA_GSTAMP(cond, kidx, slot)                                                                          \
  do {                                                                                                          \
    if (a.dbg != nullptr && (!(a.debug & 128) || (slot) == 0 || (slot) == 7) && (cond))                         \
      a.dbg[(size_t)(kidx) * 8 + (slot)] = __builtin_amdgcn_s_memtime();                                        \
  } while (0)
#else
#define EFA_GSTAMP(cond, kidx, slot) \
  do {                                \
  } while (0)
#endif
  long budget = a.spin_limit;
  // every spin is bounded twice: by a poll budget and by wall time (s_memrealtime, 100 MHz), looked at only on the
  // slow side of a poll loop; the time bound is what limits the price of a launch that cannot become fully resident
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#define EFA_TIMED_OUT() (a.spin_ticks > 0 && (long)(__builtin_amdgcn_s_memrealtime() - t_start) > a.spin_ticks)
  int polls = 0;
  auto give_up = [&]() {
    __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.status + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    g_ctl_set(&ctl[cBail], 1);
  };
  // wait until *word > thr; false if the kernel is being abandoned
  auto wait_gt = [&](const int* word, int thr, bool doze) {
    while (g_ctl(word) <= thr) {
      if ((++polls & 15) == 0) {
        if (g_ctl(&ctl[cBail]) != 0) return false;
        budget -= 16;
        if (budget <= 0 || EFA_TIMED_OUT()) {
          give_up();
          return false;
        }
      }
      if (EFA_EXP(512)) __builtin_amdgcn_s_sleep(6);
      else if (doze) __builtin_amdgcn_s_sleep(1);
    }
    return true;
  };

  auto wait2_gt = [&](const int* wa, int ta, const int* wb, int tb, bool doze) {  // both words in one LDS round trip
    for (;;) {
      const int va = g_ctl_lane(wa), vb = g_ctl_lane(wb);
      if (__builtin_amdgcn_readfirstlane(va) > ta && __builtin_amdgcn_readfirstlane(vb) > tb) return true;
      if ((++polls & 15) == 0) {
        if (g_ctl(&ctl[cBail]) != 0) return false;
        budget -= 16;
        if (budget <= 0 || EFA_TIMED_OUT()) {
          give_up();
          return false;
        }
      }
      if (EFA_EXP(512)) __builtin_amdgcn_s_sleep(6);
      else if (doze) __builtin_amdgcn_s_sleep(1);
    }
  };

  // G = Y Y^T for the pivot rows, every wave takes two 16 x 16 tiles (between the block-start barriers)
  auto form_gram = [&]() {
    const int I = wave >> 1, J0 = (wave & 1) * 2;
    if (16 * I < nb) {
      v4f64 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
      const double* pa = Yt + (size_t)(16 * I + (lane & 15)) * SP + (lane >> 4);
      const double* pb0 = Yt + (size_t)(16 * J0 + (lane & 15)) * SP + (lane >> 4);
      const double* pb1 = pb0 + 16 * SP;
      for (int s = 0; s < PAD / 4; ++s) {
        const double av = pa[4 * s], b0 = pb0[4 * s], b1 = pb1[4 * s];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b1, acc1, 0, 0, 0);
      }
      // D layout (probed, tools/mfma_probe.hip): row = 4 v + lane/16, col = lane%16
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int gi = 16 * I + 4 * v + (lane >> 4);
        G_s[gi * kRowsWG + 16 * J0 + (lane & 15)] = acc0[v];
        G_s[gi * kRowsWG + 16 * (J0 + 1) + (lane & 15)] = acc1[v];
      }
    }
  };

  // ======================================================================================
  // wave 7: loader (follower mode) / forwarder (leader mode)
  // ======================================================================================
  if (wave == 7) {
    // Three phases, each its own loop: follow the records before the block, forward the block, follow the
    // rest.  (As ONE loop the compiler merged the pending-load state of the poll path into the forward
    // path and waited vmcnt(0) per forwarded record, i.e. for the previous record's store to reach
    // memory: ~1400 cycles per ob, which bounded the whole kernel.)
    bool failed = false;
    int barriers_left = leads ? 3 : 0;
    auto follow = [&](long next, const long limit) {
      while (next < limit && !failed) {
        const int nrec = (int)((limit - next < kPollG) ? (limit - next) : kPollG);
        u64 v[kPollG][EPL];
#pragma unroll
        for (int d = 0; d < kPollG; ++d) {
          const long kk = next + ((d < nrec) ? d : nrec - 1);
          const u64* rec = a.traj + (size_t)kk * TS;
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            const int idx = lane + 64 * e;
            v[d][e] = g_traj_load(rec + (idx < TS ? idx : TS - 1));
          }
        }
        int cnt = 0;
#pragma unroll
        for (int d = 0; d < kPollG; ++d) {
          bool ok = true;
#pragma unroll
          for (int e = 0; e < EPL; ++e) ok = ok && (v[d][e] != kTrajSentinel);
          if (cnt == d && d < nrec && __all(ok)) cnt = d + 1;
        }
        if (cnt == 0) {
          if (--budget <= 0 || __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
              ((budget & 15) == 0 && EFA_TIMED_OUT()))
            failed = true;
          __builtin_amdgcn_s_sleep(2);
          continue;
        }
        const long need = next + cnt - 1 - kRingG;  // slots are recycled only once every vector wave consumed them
        if (need >= 0) {
          for (;;) {
            int mn = g_ctl_lane(&ctl[cProg + (lane & 3)]);
            mn = min(mn, __builtin_amdgcn_mov_dpp(mn, 0xB1, 0xF, 0xF, true));
            mn = min(mn, __builtin_amdgcn_mov_dpp(mn, 0x4E, 0xF, 0xF, true));
            if (__builtin_amdgcn_readfirstlane(mn) >= (int)need) break;
            if (--budget <= 0 || g_ctl(&ctl[cBail]) != 0 || ((budget & 15) == 0 && EFA_TIMED_OUT())) {
              failed = true;
              break;
            }
          }
          if (failed) break;
        }
#pragma unroll
        for (int d = 0; d < kPollG; ++d) {
          if (d < cnt) {
            double* slot = ring + (size_t)((next + d) % kRingG) * TS;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
              const int idx = lane + 64 * e;
              if (idx < TS) slot[idx] = __longlong_as_double((long long)v[d][e]);
            }
          }
        }
        next += cnt;
        if (lane == 0) g_ctl_set(&ctl[cReady], (int)next);
      }
    };
    follow(0, (own0 < P) ? own0 : P);
    EFA_BLOCKSTAMP(lane == 0 && leads, 3);
    if (leads && !failed) {
      __syncthreads();  // B1: the vector waves have parked their rows in the tile
      form_gram();
      if (a.loc_mode != 0) {  // this block's 64 x 64 corner of the obs-obs taper
        for (int i = lane; i < kRowsWG * kRowsWG; i += 64) {
          const long kg = own0 + (i >> 6), rg = own0 + (i & 63);
          tw_s[i] = (kg < P && rg < R) ? a.tw[(size_t)kg * R + rg] : 1.0;
        }
      }
      __syncthreads();  // B2: G and the taper corner are complete
      if (EFA_EXP(65536)) return;  // TIMING EXPERIMENT (single workgroup only): the pivot wave alone
      barriers_left = 1;
      for (long f = own0; f < own1; ++f) {
        if (!wait2_gt(&ctl[cReady], (int)f, &ctl[cSReady], (int)(f - own0), true)) {
          failed = true;
          break;
        }
        const double* slot = ring + (size_t)(f % kRingG) * TS;
        const int st = (int)(f - own0);
        const double* sc = s_sc + (size_t)st * kScStride;
        u64* rec = a.traj + (size_t)f * TS;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          const int idx = lane + 64 * e;  // record = ye, then 8 scalars of which the followers read [2..5]
          const int si = idx - PAD - 2;
          if (idx < TS) g_traj_store(rec + idx, idx < PAD ? slot[idx] : ((si >= 0 && si < 4) ? sc[si] : 0.0));
        }
        if (lane == 0) g_ctl_set(&ctl[cFwd], (int)f);
        // the ob's diagnostics and sweep coefficients, derived from the pivot's record (the pivot wave itself
        // keeps nothing per ob: every instruction there is on the chain).  The same operations in the same
        // order as the pivot's own lane k: km = kc rden with kc = G_kk/(M-1) (x the taper of an ob with itself,
        // exactly 1), and the ob's own row is scaled by (1 - kb_k)  (:144-149).
        if (lane == 0) {
          const double2 s01 = *reinterpret_cast<const double2*>(sc);      // innov, rden
          const double2 s23 = *reinterpret_cast<const double2*>(sc + 2);  // beta, active
          const double2 s45 = *reinterpret_cast<const double2*>(sc + 4);  // prior mean, prior var
          const double2 gk = s_gk[st * kRowsWG + st];                     // G_kk, kb_k
          const bool act = s23.y != 0.0;
          a.prior_mean[f] = s45.x;                                        // :66
          a.prior_var[f] = s45.y;                                         // :70
          double* ck = a.coef + (size_t)f * kCoefStride;
          ck[0] = act ? s01.x : 0.0;
          ck[1] = act ? s01.y : 0.0;
          ck[2] = act ? s23.x : 0.0;
          ck[3] = act ? 1.0 : 0.0;
          a.assimilated[f] = act ? 1 : 0;                                 // :74-76, :149
          if (act) {
            double kc = gk.x * rM1;                                       // :95 (taper of the ob with itself: 1)
            if (a.loc_mode != 0) kc = tw_s[st * kRowsWG + st] * kc;       // :115
            const double km = kc * s01.y;                                 // :119
            const double fsc = 1.0 - gk.y;
            a.post_mean[f] = s45.x + km * s01.x;                          // :130
            a.post_var[f] = (fsc * fsc) * s45.y;
          }
        }
        EFA_GSTAMP(lane == 0, f, 7);
      }
      EFA_BLOCKSTAMP(lane == 0, 2);
      __syncthreads();  // B3: done with the pivot's records
      barriers_left = 0;
      if (!failed) follow(own1, P);
    }
    if (failed && lane == 0) give_up();
    for (; barriers_left > 0; --barriers_left) __syncthreads();  // never leave the others at a barrier
    return;
  }

  // ======================================================================================
  // wave 4 (pivot) and waves 5, 6 (helpers): the Gram-space recurrence of this workgroup's block
  // ======================================================================================
  if (wave >= kVW) {
    if (!leads) return;
    // the block's ob constants are fetched now, while the wave waits for its block anyway (after B2 the
    // global-memory round trip would sit on the hand-over between workgroups)
    const bool pre_ob = lane < nb;
    const double pre_err = (wave == kVW && pre_ob) ? a.ob_error[own0 + lane] : 1.0;
    const double pre_val = (wave == kVW && pre_ob) ? a.ob_value[own0 + lane] : 0.0;
    const bool pre_asm = (wave == kVW && pre_ob) ? (a.ob_assim[own0 + lane] != 0) : false;
    const double pre_sq = sqrt(pre_err);
    __syncthreads();  // B1
    form_gram();
    __syncthreads();  // B2
    if (EFA_EXP(65536) && wave != kVW) return;
    if (wave == kVW) {
      // ---------------- pivot wave: lane j <-> row j of the workgroup ----------------
      // The loop below is the serial chain of the whole filter, and one wave issues in order: every
      // instruction in it costs issue slots, so per-step work is kept to the recurrence itself.  The gain
      // factors are arranged for a short dependent chain:
      //   kdenom -> q0 = rsq(kdenom) -> { Newton step of q  ||  beta0 = 1/(1 + sqrt(err) q0) } -> beta
      // with beta = beta0 - beta0^2 sqrt(err) q0 d for q = q0 (1 + d)  (d ~ 1e-8: the d^2 term is < 1 ulp).
      double mu = pm[lane], xmv = pm[kRowsWG + lane];
      const bool my_asm = pre_asm;
      const double err_l = pre_err, sq_l = pre_sq, val_l = pre_val;   // this lane's ob constants: fetched by v_readlane
      const u64 asm_mask = __ballot(my_asm);
      // cancellation guard: an assimilated pivot whose G_kk fell below 1e-3 of its value at block start.  It is
      // only ACCUMULATED in the loop (one compare + two scalar ops per step) and acted upon after the block: a
      // tripped guard abandons the whole launch, so the numbers produced meanwhile are never used.
      const double thr_ld = my_asm ? 1e-3 * G_s[lane * kRowsWG + lane] : -1.0;
      u64 bad = 0ull;
      bool bailed = false;
#ifdef EFA_PIPE_BLOCKTIME
      int slow_rows = 0;  // steps whose hand-over row was late
#endif
      double g = G_s[lane];             // row 0
      double g1 = G_s[kRowsWG + lane];  // row 1
      // values that came from LDS are pinned here: a wait left at their first use inside the step code would be
      // executed every step and drain that step's own (exec-masked, hence uncounted) record stores
      double thr_p = thr_ld;
      asm volatile("" : "+v"(mu), "+v"(xmv), "+v"(thr_p), "+v"(g), "+v"(g1));
      const double thr = thr_p;
      const bool gc = a.loc_mode != 0;
      // slow path of the hand-over (the helper is late): poll flag and row together
      auto wait_row = [&](int kk, double& r2) {
        const int* flag = &ctl[cHProg + (kk & 1)];
        for (;;) {
          const int f = g_ctl_lane(flag);
          r2 = G_s[(kk + 2) * kRowsWG + lane];
          if (__builtin_amdgcn_readfirstlane(f) >= kk + 2) return true;
          if ((++polls & 15) == 0) {
            budget -= 16;
            if (g_ctl(&ctl[cBail]) != 0) return false;
            if (budget <= 0 || EFA_TIMED_OUT()) {
              if (lane == 0) give_up();
              return false;
            }
          }
        }
      };
      // One step of the recurrence.  has1 / has2: rows kk+1 / kk+2 exist; poll: row kk+2 comes from a
      // helper (kk >= 1).  The block loop below calls it with constants, so the steady-state body is
      // straight-line code.  What one step costs this wave is its INSTRUCTION COUNT (one wave issues in order,
      // an LDS instruction costs it 15-35 cycles, a taken branch a dozen arithmetic instructions), so:
      //   - per-ob constants live in the ob's lane and are fetched with v_readlane (no LDS read, no latency);
      //   - the ob's diagnostics are not kept here: the forwarder wave derives them from the step's record;
      //   - the record is published at the END of the step, after the hand-over row has been consumed, so the
      //     wait for that row never has the record's own stores in front of it.
      auto pivot_step = [&](const int kk, const bool has1, const bool has2, const bool poll, auto gc_tag) {
        constexpr bool GC = decltype(gc_tag)::value;  // compile-time: a run-time test costs the chain a taken branch per step
        // row kk+2 from its helper, read speculatively (flag first, then the row): the helper puts the
        // row a pivot needs next into G_s before anything else, right after the previous record appears,
        // so it is normally there by now and the round trip is hidden behind the whole gain chain
        EFA_GSTAMP(lane == 0, own0 + kk, 0);
        int f_early = 0;
        double r2 = 0.0;
        if (has2) {
          f_early = poll ? g_ctl_lane(&ctl[cHProg + (kk & 1)]) : 0;
          r2 = G_s[(kk + 2) * kRowsWG + lane];
        }
        const double twk = GC ? tw_s[kk * kRowsWG + lane] : 1.0;       // consumed after the gain chain
        const bool act = ((asm_mask >> kk) & 1) != 0;
        if (!EFA_EXP(0x70)) bad |= __ballot(!(g > thr)) & (1ull << kk);
        const double Gkk = rl(g, kk), muk = rl(mu, kk), xmk = rl(xmv, kk);
        const double errk = rl(err_l, kk), sqk = rl(sq_l, kk), valk = rl(val_l, kk);
        const double mu2 = muk * muk;
        const double kdenom = __builtin_fma(Gkk, invM, errk - mu2);   // var + err  (:69, :91)
        const double q0 = __builtin_amdgcn_rsq(kdenom);
        const double e = __builtin_fma(-kdenom * q0, q0, 1.0);
        const double d = e * __builtin_fma(0.375, e, 0.5);            // q = q0 (1 + d)
        const double q = __builtin_fma(q0, d, q0);
        const double rden = q * q;                                    // 1 / kdenom
        const double sq0 = sqk * q0;
        const double b0 = 1.0 + sq0;
        const double r0 = __builtin_amdgcn_rcp(b0);
        const double eb = __builtin_fma(-b0, r0, 1.0);
        const double beta0 = __builtin_fma(r0, __builtin_fma(eb, eb, eb), r0);
        const double beta = __builtin_fma(-((beta0 * beta0) * sq0), d, beta0);  // 1 / (1 + sqrt(err / kdenom))  (:135)
        double kc = g * rM1;                                          // :95
        if (GC) kc = twk * kc;                                        // :115
        const double km = act ? kc * rden : 0.0;                      // :119
        const double kb = beta * km;                                  // :136
        const double innov = valk - xmk;                              // :85
        const double gpub = g;
        xmv = xmv + km * innov;                                       // :130
        mu = __builtin_fma(-kb, muk, mu);
        bool ok = true;
        if (has1) {
          // g1 = row kk+1 through step kk-1; row kk+2 through step kk-1 comes from its helper (handed
          // over through G_s during the helper's step kk-1)
          const double t = __builtin_fma(-kb, Gkk, g);
          const double gi = rl(g, kk + 1), ai = rl(kb, kk + 1);
          const double gnew = __builtin_fma(-ai, t, __builtin_fma(-kb, gi, g1));
          if (has2) {
            if (__builtin_expect(poll && !EFA_EXP(2048) && __builtin_amdgcn_readfirstlane(f_early) < kk + 2, 0)) {
#ifdef EFA_PIPE_BLOCKTIME
              ++slow_rows;
#endif
              ok = wait_row(kk, r2);
            }
            const double gi2 = rl(g, kk + 2), ai2 = rl(kb, kk + 2);
            g1 = __builtin_fma(-ai2, t, __builtin_fma(-kb, gi2, r2));
          }
          g = gnew;
        }
        // everything that consumes the hand-over row is computed BEFORE the record's stores are issued
        asm volatile("" : "+v"(g), "+v"(g1)::"memory");
        // the step's record: {G_kj, kb_j} per row, then (lane 0) the scalars and the flag
        s_gk[kk * kRowsWG + lane] = make_double2(gpub, kb);
        if (lane == 0) {
          double2* sc = reinterpret_cast<double2*>(s_sc + (size_t)kk * kScStride);
          sc[0] = make_double2(innov, rden);
          sc[1] = make_double2(beta, act ? 1.0 : 0.0);
          sc[2] = make_double2(xmk, __builtin_fma(Gkk, invM, -mu2));   // prior mean (:66), np.var ddof = 0 (:69, :70)
          g_ctl_set(&ctl[cSReady], kk + 1);
        }
        EFA_GSTAMP(lane == 0, own0 + kk, 1);
        return ok;
      };
      EFA_BLOCKSTAMP(lane == 0, 0);
#ifdef EFA_PIPE_BLOCKTIME
      if (a.dbg != nullptr && lane == 0) a.dbg[(size_t)own0 * 8 + 6] = __builtin_amdgcn_s_memrealtime();
#endif
      {
        int kk = 0;
        bool ok = true;
        auto run_block = [&](auto gc_tag) {
          if (nb >= 4) {
            ok = pivot_step(0, true, true, false, gc_tag);
            for (kk = 1; ok && kk < nb - 2; ++kk) ok = pivot_step(kk, true, true, true, gc_tag);  // steady state
          }
          for (; ok && kk < nb; ++kk) ok = pivot_step(kk, kk + 1 < nb, kk + 2 < nb, kk >= 1, gc_tag);
        };
        if (gc) run_block(std::true_type());
        else run_block(std::false_type());
        if (ok && bad != 0ull) {  // the downdate may have cancelled: abandon the launch (status[2]: the host
          if (lane == 0) {        // re-runs Phase A with the vector-chain kernel)
            give_up();
            __hip_atomic_store(a.status + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          ok = false;
        }
        bailed = !ok;
      }
      EFA_BLOCKSTAMP(lane == 0, 1);
#ifdef EFA_PIPE_BLOCKTIME
      if (a.dbg != nullptr && lane == 0) a.dbg[(size_t)own0 * 8 + 7] = __builtin_amdgcn_s_memrealtime();
      if (a.dbg != nullptr && lane == 0) a.dbg[(size_t)own0 * 8 + 4] = (u64)slow_rows;
#endif
      pm[kRowsWG + lane] = xmv;  // obs-space means of all 64 rows after the block, back to the vector waves
      (void)bailed;
      __syncthreads();  // B3
      return;
    }
    // ---------------- helper waves: rows that become pivots later ----------------
    // helper h keeps the 32 rows of its parity in registers (lane = column) and applies each step's
    // rank-one downdate to those at least three rows ahead of the pivot; row kk+3 is handed to the
    // pivot wave through G_s at the start of step kk, a full step before the pivot needs it (the pivot
    // applies the last two steps to it itself)
    const int h = wave - kVW - 1;  // 0, 1
    __builtin_amdgcn_s_setprio(2);  // measured: 2% faster Phase A with it, 3% slower if the pivot wave is raised too
    double gr[kRowsWG / 2];
#pragma unroll
    for (int r = 0; r < kRowsWG / 2; ++r) gr[r] = (2 * r + h >= 3) ? G_s[(2 * r + h) * kRowsWG + lane] : 0.0;
    auto downdate = [&](const double2* rec, double kb, double t, int kk, auto half) {
      constexpr int H = decltype(half)::value;  // rows 32 H .. 32 H + 31, sixteen of them here
      double2 ga[16];  // {G_ki, kb_i} of the rows: uniform-address LDS reads, all in flight together
#pragma unroll
      for (int q = 0; q < 16; ++q) ga[q] = rec[32 * H + 2 * q + h];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {  // four rows at a time, the two dependent FMAs of a row kept apart
#pragma unroll
        for (int q = 4 * q4; q < 4 * q4 + 4; ++q) gr[16 * H + q] = __builtin_fma(-kb, ga[q].x, gr[16 * H + q]);
#pragma unroll
        for (int q = 4 * q4; q < 4 * q4 + 4; ++q) gr[16 * H + q] = __builtin_fma(-ga[q].y, t, gr[16 * H + q]);
      }
    };
    for (int kk = 0; kk + 3 < nb; ++kk) {
      if (!wait_gt(&ctl[cSReady], kk, false)) break;
      EFA_GSTAMP(lane == 0 && h == 0 && (a.debug & 8), own0 + kk, 3);
#ifdef EFA_PIPE_STAMPS
      if ((a.debug & 4096) && a.dbg != nullptr && h == 1) {  // LDS round-trip latency under the real load
        unsigned long long t0, t1, t2;
        int vv;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(vv) : "v"((int)(size_t)0) : "memory");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
        if (lane == 0) {
          a.dbg[(size_t)(own0 + kk) * 8 + 1] = t1 - t0;  // LDS round trip + one s_memtime round trip
          a.dbg[(size_t)(own0 + kk) * 8 + 2] = t2 - t1;  // one s_memtime round trip
        }
        asm volatile("" ::"v"(vv));
      }
#endif
      const double2* rec = s_gk + kk * kRowsWG;
      const double2 own = rec[lane];
      const double Gkk = rl(own.x, kk);
      const double kb = own.y;
      const double t = own.x - kb * Gkk;
      EFA_GSTAMP(lane == 0 && h == ((kk + 1) & 1) && !(a.debug & 8), own0 + kk, 5);
      EFA_GSTAMP(lane == 0 && h == 0 && (a.debug & 8), own0 + kk, 4);
      // The row the pivot needs next (kk+3) first and on its own -- one uniform read, the same two FMAs the
      // bulk applies to it below (so the register copy ends up identical) -- so that it is in G_s well
      // before the pivot's next step starts and the pivot's early read of it never has to be repeated.
      if (((kk + 3) & 1) == h) {
        const int hr = kk + 3;
        const double2 gu = rec[hr];
        double cur = 0.0;
        switch (hr >> 1) {
#define EFA_HR_CASE(R) \
  case R:              \
    cur = gr[R];       \
    break;
          EFA_HR_CASE(1) EFA_HR_CASE(2) EFA_HR_CASE(3) EFA_HR_CASE(4) EFA_HR_CASE(5) EFA_HR_CASE(6) EFA_HR_CASE(7)
          EFA_HR_CASE(8) EFA_HR_CASE(9) EFA_HR_CASE(10) EFA_HR_CASE(11) EFA_HR_CASE(12) EFA_HR_CASE(13)
          EFA_HR_CASE(14) EFA_HR_CASE(15) EFA_HR_CASE(16) EFA_HR_CASE(17) EFA_HR_CASE(18) EFA_HR_CASE(19)
          EFA_HR_CASE(20) EFA_HR_CASE(21) EFA_HR_CASE(22) EFA_HR_CASE(23) EFA_HR_CASE(24) EFA_HR_CASE(25)
          EFA_HR_CASE(26) EFA_HR_CASE(27) EFA_HR_CASE(28) EFA_HR_CASE(29) EFA_HR_CASE(30) EFA_HR_CASE(31)
#undef EFA_HR_CASE
          default: break;
        }
        cur = __builtin_fma(-kb, gu.x, cur);
        cur = __builtin_fma(-gu.y, t, cur);
        G_s[hr * kRowsWG + lane] = cur;
        if (lane == 0) g_ctl_set(&ctl[cHProg + h], hr);
        EFA_GSTAMP(lane == 0 && !(a.debug & 8), own0 + kk, 6);
      }
      // rows behind the pivot may be updated too (their registers are dead)
      if (EFA_EXP(16)) continue;  // timing experiment: hand-over only
      if (kk + 3 < 32) downdate(rec, kb, t, kk, std::integral_constant<int, 0>());
      EFA_GSTAMP(lane == 0 && h == 0 && (a.debug & 8), own0 + kk, 5);
      downdate(rec, kb, t, kk, std::integral_constant<int, 1>());
      EFA_GSTAMP(lane == 0 && h == 0 && (a.debug & 8), own0 + kk, 6);
    }
    __syncthreads();  // B3
    return;
  }

  // ======================================================================================
  // waves 0-3: vector waves, rows in registers for the whole kernel
  // ======================================================================================
  const int j = lane & (PLg - 1);
  const int grp = lane / PLg;
  const int i_loc = wave + kVW * grp;  // consecutive obs in different waves
  const long row = own0 + i_loc;
  const bool live = row < R;
  const bool vec = (M % 2 == 0);
  double x[2 * NC];
  double xm = 0.0;
  if (live) {
    if (vec) load_row<PLg, NC, true>(a.Yp + (size_t)row * M, M, j, x);
    else load_row<PLg, NC, false>(a.Yp + (size_t)row * M, M, j, x);
    xm = a.ym[row];
  } else {
#pragma unroll
    for (int c = 0; c < 2 * NC; ++c) x[c] = 0.0;
  }
  const bool use_tw = (a.loc_mode != 0) && live;
  double wq0 = 1.0, wq1 = 1.0;  // GC taper of ob k against this row, prefetched two obs ahead (follower mode)
  auto prime_tw = [&](long k0) {
    if (use_tw) {
      wq0 = (k0 < P) ? a.tw[(size_t)k0 * R + row] : 1.0;
      wq1 = (k0 + 1 < P) ? a.tw[(size_t)(k0 + 1) * R + row] : 1.0;
    }
  };
  prime_tw(0);
  int barriers_left = leads ? 3 : 0;
  bool bailed = false;
  auto min_prog = [&]() {  // least-advanced consumer of the ring: the 4 vector waves and the forwarder
    int mn = min(g_ctl_lane(&ctl[cProg + j]), g_ctl_lane(&ctl[cFwd]));
    mn = min(mn, __builtin_amdgcn_mov_dpp(mn, 0xB1, 0xF, 0xF, true));
    mn = min(mn, __builtin_amdgcn_mov_dpp(mn, 0x4E, 0xF, 0xF, true));
    return __builtin_amdgcn_readfirstlane(mn);
  };
  long k = 0;
  while (k < P && !bailed) {
    if (leads && k == own0) {
      // ---------------- this workgroup's block ----------------
#pragma unroll
      for (int c = 0; c < NC; ++c)
        *reinterpret_cast<double2*>(Yt + (size_t)i_loc * SP + 2 * PLg * c + 2 * j) = make_double2(x[2 * c], x[2 * c + 1]);
      const double rmean = group_rowsum<PLg, NC>(x) * invM;
      if (j == 0) {
        pm[i_loc] = rmean;
        pm[kRowsWG + i_loc] = xm;
      }
      __syncthreads();  // B1: tile and parked means complete
      form_gram();
      // For the block the rows change layout.  Wave w re-reads the 16 rows it has just parked (rows
      // 4 rho + w, rho = 0..15) as NJ accumulator tiles of v_mfma_f64_16x16x4_f64: register v of tile J in
      // lane l is member 16 J + (l & 15) of tile row rho = 4 v + (l >> 4).  With the gains kb known from
      // the pivot wave, the rows are then updated
      //   - by ONE matrix-core instruction per tile for every four steps (rank-4 update, A = -kb of the
      //     four steps, B = their four ye), off the chain, and
      //   - step by step only inside a window of the next eight rows to be published (a per-lane
      //     coefficient, zero outside the window; those (row, step) pairs are masked out of the rank-4
      //     update), so that the row that becomes ye of the next ob is always current.
      // The quad layout's 13 replicated ds_read_b128 per wave and step (the LDS pipe is shared by all
      // eight waves) and its 26 FMAs per wave and step shrink to 7 ds_read_b64 and <= 14 FMAs.
      v4f64 xt[NJ];
      const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
      for (int J = 0; J < NJ; ++J)
#pragma unroll
        for (int v = 0; v < 4; ++v) xt[J][v] = Yt[(size_t)(16 * v + 4 * lr + wave) * SP + 16 * J + lc];
      __syncthreads();  // B2: G complete; the tile region now belongs to the pivot wave's records
      if (EFA_EXP(65536)) return;
      barriers_left = 1;
      auto publish_row = [&](int r) {  // block row r (held by this wave: (r & 3) == wave) IS ye of ob own0 + r
        double* slot = ring + (size_t)((own0 + r) % kRingG) * TS;
        if (lr == ((r >> 2) & 3)) {
          switch (r >> 4) {
#define EFA_PUB_CASE(V)                                                      \
  case V:                                                                    \
    _Pragma("unroll") for (int J = 0; J < NJ; ++J)                           \
      if (16 * J + lc < PAD) slot[16 * J + lc] = xt[J][V];                   \
    break;
            EFA_PUB_CASE(0)
            EFA_PUB_CASE(1)
            EFA_PUB_CASE(2)
            EFA_PUB_CASE(3)
#undef EFA_PUB_CASE
          }
        }
        if (lane == 0) g_ctl_set(&ctl[cReady], (int)(own0 + r + 1));
      };
      if (wave == 0) publish_row(0);
      for (int st = 0; st < nb; ++st) {
        const long kg = own0 + st;
        const int hi = 4 * ((st + 1) >> 2) + 7;                         // window: rows st < r <= hi
        const int nu = st + 1;                                          // the row that is ye of the next ob
        const bool mine = ((nu & (kVW - 1)) == wave) && nu < nb;        // ... is held by this wave
        const bool batch = ((st & 3) == 2) || st == nb - 1;             // a group of four steps is complete
        const bool inwin_any = (60 + wave > st) && (wave <= hi);        // some row of this wave is in the window
        if (!inwin_any && !batch) continue;
        if (!wait2_gt(&ctl[cReady], (int)kg, &ctl[cSReady], st, !mine)) {
          bailed = true;
          break;
        }
        EFA_GSTAMP(mine && lane == 0 && !(a.debug & 8), kg, 3);
        if (inwin_any && EFA_EXP(64)) {
          if (mine) publish_row(nu);
        } else if (inwin_any) {
          const double* slot = ring + (size_t)(kg % kRingG) * TS;
          double y[NJ];
#pragma unroll
          for (int J = 0; J < NJ; ++J) y[J] = slot[16 * J + lc];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            if (16 * v + 12 + wave > st && 16 * v + wave <= hi) {       // uniform: tile rows 4v..4v+3 touch the window
              const int row = 16 * v + 4 * lr + wave;
              double kb = s_gk[st * kRowsWG + row].y;                   // :136
              if (!(row > st && row <= hi)) kb = 0.0;
#pragma unroll
              for (int J = 0; J < NJ; ++J) xt[J][v] = __builtin_fma(-kb, y[J], xt[J][v]);  // :141
            }
          }
          if (mine) {  // the chain: publish the next ye
            if ((nu & 3) == 0) {  // recycling guard, amortised over four records
              while (min_prog() < (int)(kg + 1 + 3 - kRingG)) {
                if (g_ctl(&ctl[cBail]) != 0 || --budget <= 0 || ((budget & 15) == 0 && EFA_TIMED_OUT())) {
                  bailed = true;
                  break;
                }
              }
              if (bailed) break;
            }
            publish_row(nu);
            EFA_GSTAMP(lane == 0 && !(a.debug & 8), kg, 4);
          }
        }
        if (batch && EFA_EXP(32)) {
          if (lane == 0) g_ctl_set(&ctl[cProg + wave], (int)kg);
        } else if (batch) {
          // rank-4 update with the steps 4 tp - 1 .. 4 tp + 2 (those that exist), minus what the window did
          const int tp = (st + 1) >> 2;
          const int sa = 4 * tp - 1 + lr;                               // this lane's k slice
          const bool valid = sa >= 0 && sa <= st;
          const int sc = valid ? sa : st;
          const int row = 4 * lc + wave;                                // A: tile row rho = l & 15
          double av = s_gk[sc * kRowsWG + row].y;
          if (!valid || (row > sa && row <= 4 * tp + 7)) av = 0.0;
          av = -av;
          const double* bs = ring + (size_t)((own0 + sc) % kRingG) * TS;
#pragma unroll
          for (int J = 0; J < NJ; ++J) xt[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bs[16 * J + lc], xt[J], 0, 0, 0);
          if (lane == 0) g_ctl_set(&ctl[cProg + wave], (int)kg);       // ring slots up to kg consumed
        }
      }
      __syncthreads();  // B3: every wave is done with the pivot's records; the tile region is free again
      barriers_left = 0;
      if (bailed) break;
      // back to the follower layout through the tile (each wave reads only rows it wrote itself)
#pragma unroll
      for (int J = 0; J < NJ; ++J)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (16 * J + lc < PAD) Yt[(size_t)(16 * v + 4 * lr + wave) * SP + 16 * J + lc] = xt[J][v];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const double2 v = *reinterpret_cast<const double2*>(Yt + (size_t)i_loc * SP + 2 * PLg * c + 2 * j);
        x[2 * c] = v.x;
        x[2 * c + 1] = v.y;
      }
      xm = pm[kRowsWG + i_loc];  // the pivot wave carried the obs-space means through the block (:130)
      k = own1;
      prime_tw(k);
      continue;
    }
    // ---------------- follower step: consume record k from the ring ----------------
    if (!wait_gt(&ctl[cReady], (int)k, true)) {
      bailed = true;
      break;
    }
    const double* slot = ring + (size_t)(k % kRingG) * TS;
    double y[2 * NC];
    lds_read_row<PLg, NC>(slot, j, y);
    const double2 s23 = *reinterpret_cast<const double2*>(slot + PAD + 2);  // innov, rden
    const double2 s45 = *reinterpret_cast<const double2*>(slot + PAD + 4);  // beta, active
    if (lane == 0) g_ctl_set(&ctl[cProg + wave], (int)k);
    const double w = wq0;
    wq0 = wq1;
    if (use_tw) wq1 = a.tw[(size_t)((k + 2 < P) ? k + 2 : P - 1) * R + row];
    if (EFA_EXP(1024) && own0 >= 128) {  // TIMING EXPERIMENT: followers far from the leader skip their work
      ++k;
      continue;
    }
    if (__builtin_amdgcn_readfirstlane((int)(s45.y != 0.0)) != 0) {
      const double dot = group_dot<PLg, NC>(x, y);
      double kc = dot * rM1;                              // :95
      if (a.loc_mode != 0) kc = (live ? w : 0.0) * kc;    // :115
      const double km = kc * s23.y;                       // :119
      xm = xm + km * s23.x;                               // :130
      const double kb = s45.x * km;                       // :136
#pragma unroll
      for (int c = 0; c < 2 * NC; ++c) x[c] = __builtin_fma(-kb, y[c], x[c]);  // :141
    }
    ++k;
  }
  for (; barriers_left > 0; --barriers_left) __syncthreads();
  if (bailed || g_ctl(&ctl[cBail]) != 0) return;  // nothing written back: the host re-runs Phase A
  if (live) {
    if (vec) store_row<PLg, NC, true>(a.Yp + (size_t)row * M, M, j, x);
    else store_row<PLg, NC, false>(a.Yp + (size_t)row * M, M, j, x);
    if (j == 0) a.ym[row] = xm;
  }
}

template <int NC>
hipError_t gram_launch(const PipeArgs& a, hipStream_t s) {
  const long grid = (a.R + kRowsWG - 1) / kRowsWG;
  const size_t lds = GramShape<NC>::lds_bytes(a.loc_mode != 0);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pipe_gram<NC>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  // every workgroup waits for records of every other: the grid must fit the device at once
  int per_cu = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&k_pipe_gram<NC>), kGT, lds);
  if (e != hipSuccess) return e;
  if (grid > (long)per_cu * a.cu_count) return hipErrorCooperativeLaunchTooLarge;
  hipLaunchKernelGGL((k_pipe_gram<NC>), dim3((unsigned)grid), dim3(kGT), lds, s, a);
  return hipGetLastError();
}

}  // namespace

bool pipeline_gram_supported(int M, long R, int loc_mode) {
  if (!(M >= 2 && M <= 128 && R > 0 && (R + kRowsWG - 1) / kRowsWG <= kPipeMaxWGs)) return false;
  const int nc = (M + 2 * PLg - 1) / (2 * PLg);
  const int pad = 2 * PLg * nc, ts = pad + kTrajScalars;
  const int sp = pad + ((2 - pad % 32) + 32) % 32;
  const size_t ureg = (size_t)kRowsWG * (sp > 2 * kRowsWG + 8 ? sp : 2 * kRowsWG + 8);
  const size_t bytes = ((size_t)kRingG * ts + kRowsWG * kRowsWG + ureg + 5 * kRowsWG + (loc_mode ? kRowsWG * kRowsWG : 0)) * 8 + 64;
  return bytes <= 160 * 1024;
}

hipError_t launch_pipeline_gram(const PipeArgs& a, hipStream_t s) {
  if (!pipeline_gram_supported(a.M, a.R, a.loc_mode) || a.P <= 0) return hipErrorInvalidValue;
  switch ((a.M + 2 * PLg - 1) / (2 * PLg)) {
    case 1: return gram_launch<1>(a, s);
    case 2: return gram_launch<2>(a, s);
    case 3: return gram_launch<3>(a, s);
    case 4: return gram_launch<4>(a, s);
    case 5: return gram_launch<5>(a, s);
    case 6: return gram_launch<6>(a, s);
    case 7: return gram_launch<7>(a, s);
    case 8: return gram_launch<8>(a, s);
    case 9: return gram_launch<9>(a, s);
    case 10: return gram_launch<10>(a, s);
    case 11: return gram_launch<11>(a, s);
    case 12: return gram_launch<12>(a, s);
    case 13: return gram_launch<13>(a, s);
    case 14: return gram_launch<14>(a, s);
    case 15: return gram_launch<15>(a, s);
    case 16: return gram_launch<16>(a, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace efa
