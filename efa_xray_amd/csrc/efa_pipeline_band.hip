// Persistent Phase-A pipeline, BAND leader ("k_pipe_band"): unlocalised cycles and, since the end of round 2,
// Gaspari-Cohn cycles (template parameter GC: the obs-obs taper of the block in LDS, two-term downdates).
//
// Same launch structure, follower code and inter-workgroup protocol as efa_pipeline_gram.hip (one
// workgroup per 64 obs rows, rows resident in registers, sentinel-validated agent-scope trajectory
// records).  What changes is how a workgroup LEADS its own 64 observations.
//
// In efa_pipeline_gram.hip every step hands data between waves through LDS: the pivot wave waits
// for a row from a helper, two helper waves read 32 per-row operand pairs each, the four vector waves
// pass ye_k from wave to wave (write, flag, poll, read) before ye_{k+1} can be formed.  Stage
// timings with the production code (tools/gram_blocktime.py, profiles/r02_phase_a_stages.txt):
// pivot alone 1 060 cycles per step, pivot waiting for its helper row 1 520 (late in 61 of 64
// steps), vector chain 1 650, followers 950.  Here the block is cut into BANDS of 4 observations and
// the waves exchange data once per band:
//
//   pivot wave   keeps the band's 4 rows of G (lane = column) in registers and runs the 4 steps of
//                the Gram-space recurrence alone: the chain carries ONE gain constant per ob,
//                c = beta / ((M-1) kdenom) (gain_c), the band's later rows are downdated in registers, the
//                ob constants arrive by wave-uniform loads one band ahead.  Per step it publishes {G_kj, kb_j};
//                per band the inverse of the band's unit lower triangular factor
//                (ye_{r0+s} = y_{r0+s} - sum_{t<s} kb^{(t)}_{r0+s} ye_{r0+t}  <=>  YE = L^-1 Y).
//                (Round 3: no mean chain -- a block whose pivot rows are not centred goes to the vector-chain kernel --
//                and no per-ob latches: the forwarder wave recomputes 1/kdenom and beta from the recorded G_kk.)
//   2 G waves    hold G as matrix-core accumulator tiles (two tile columns each) and apply a
//                band's downdates as matrix-core updates per tile (v_mfma_f64_16x16x4_f64, from the step
//                records alone: A = -G_ki, B = gamma G_kj without localisation; A = -G_ki, B = kb_j and
//                A = -kb_i, B = t_j with it).  They run ONE BAND BEHIND the pivot wave: after band b the rows of
//                band b+2 go back to it first (round 3: the pivot <-> G-wave round trip through LDS flags, ~2.4 k
//                cycles, is then off the chain; the pivot applies band b+1 to those rows itself).
//   4 vector waves  hold 16 consecutive block rows each as accumulator tiles.  The band's four rows are one
//                register of ONE wave, already in B-operand layout: that wave forms YE = L^-1 Y for every
//                column tile from its registers, publishes it in the ring and applies the band to its own
//                tile from the same registers; the other three take the band from the ring (rank-4 update,
//                A = -kb, B = YE).  No ye_k -> ye_{k+1} chain across waves, no rendezvous among them.
//   loader wave  carries the obs-space means, forwards the band's records to global memory, derives the
//                obs' diagnostics once per block.
//
// LDS instructions per observation step in the leading workgroup drop from ~140 to ~25, and no wave
// waits for another inside a band.  The Gram-space cancellation guard and the fallbacks are those of
// efa_pipeline_gram.hip.
#include <type_traits>

#include "efa_device.h"
#include "efa_internal.h"
#include "efa_rows.h"

namespace efa {
namespace {

typedef unsigned long long u64;
typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int kVW = 4;        // vector waves (quad per row, one per SIMD)
constexpr int kGT = 512;      // threads: 4 vector + pivot + 2 G waves + loader
constexpr int PLg = kPipeLanes;  // lanes per row: same record layout as efa_pipeline.hip
constexpr int kRingG = 16;    // LDS ring slots for ye rows (four bands; 8 and 32 slots measure the same)
constexpr int kPollGMax = 8;  // records per poll of the loader wave: two bands without localisation (it must be able to catch up: a poll is a
                              // global round trip), one band with (each record also brings its row of the obs-obs taper table)
constexpr int kRowsWG = kPipeRowsPerWG;  // 64
#ifndef EFA_BAND
#define EFA_BAND 4
#endif
constexpr int kBand = EFA_BAND;  // observations per band (4 or 8)
static_assert(kBand == 4 || kBand == 8, "band size");
constexpr int kNBands = kRowsWG / kBand;
static_assert(PLg == 4 && kRowsWG == 64, "layout assumptions of the band kernel");

// control words (ints in LDS)
static_assert(true, "");
enum { cReady = 0, cBail = 1, cSReady = 2, cFwd = 3, cProg = 4 /* ..7 */, cBandH = 8 /* ..9 */, cLinv = 10, cPark = 11, cYe = 12, cHalf = 13, cDef = 14 };
#ifndef EFA_EARLY
#define EFA_EARLY 0
#endif
#ifndef EFA_G_DEFER
#define EFA_G_DEFER 0  // 1: the G waves owe a band's trailing update until after the next band's early rows (measured: slower)
#endif
// kEarly >= 1: the G waves hand the NEXT band's rows to the pivot wave once kEarly steps of the current band are applied
//   to them (the pivot applies the band's other steps itself): one pivot -> G wave -> pivot round trip through LDS flags
//   per band, ~2.4 k cycles, which the band's remaining steps do not cover (profiles/r03_phase_a_pivot_loop.txt).
// kEarly == 0 (default): the G waves run one whole band behind.  After band b they hand over the rows of band b + 2,
//   current through band b; the pivot wave, which needs them a full band later, applies band b + 1 to them itself (16 row
//   updates per band instead of 12).  The round trip has a band's time to complete: the pivot never waits.
// With Gaspari-Cohn localisation a row update of the pivot wave is two-term (two v_readlane pairs): the 16 updates per band of
// the one-band-behind scheme then cost what the wait did, and the hand-over after ONE step measures best (configs[3]'s
// Phase A 6.98 ms against 7.14).
#ifndef EFA_EARLY_GC
#define EFA_EARLY_GC 1
#endif
constexpr int kEarlyNone = EFA_EARLY, kEarlyGC = EFA_EARLY_GC;
static_assert(kEarlyNone >= 0 && kEarlyNone < kBand && kEarlyGC >= 0 && kEarlyGC < kBand, "early hand-over");
constexpr int kScStride = 4;  // doubles per ob: rden, beta (latched by the pivot wave), innov, active (added by the forwarder): the record's scalars

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
// The gain chain of one observation from its obs-space variance G_kk / M and error variance (ensrf.py:91, :119, :135):
//   kdenom -> q0 = rsq(kdenom) -> { Newton step of q  ||  beta0 = 1/(1 + sqrt(err) q0) } -> rden = 1/kdenom, beta.
// The pivot wave (on the serial chain) and the forwarder wave (for the records and diagnostics) both call it on the same
// inputs, so they hold the same bits without handing them to each other.
__device__ __forceinline__ void gain_scalars(double Gkk, double invM, double errk, double sqk, double& rden, double& beta) {
  const double kdenom = __builtin_fma(Gkk, invM, errk);         // var + err  (:69, :91)
  const double q0 = __builtin_amdgcn_rsq(kdenom);
  const double e = __builtin_fma(-kdenom * q0, q0, 1.0);
  const double d = e * __builtin_fma(0.375, e, 0.5);            // q = q0 (1 + d)
  const double q = __builtin_fma(q0, d, q0);
  rden = q * q;                                                 // 1 / kdenom
  const double sq0 = sqk * q0;
  const double b0 = 1.0 + sq0;
  const double r0c = __builtin_amdgcn_rcp(b0);
  const double eb = __builtin_fma(-b0, r0c, 1.0);
  const double beta0 = __builtin_fma(r0c, __builtin_fma(eb, eb, eb), r0c);
  beta = __builtin_fma(-((beta0 * beta0) * sq0), d, beta0);     // 1 / (1 + sqrt(err / kdenom))  (:135)
}
// What the serial chain itself needs of an observation is one number, c = beta / ((M-1) kdenom): every gain is
// kb_j = w_kj c G_kj and the downdate of G is gamma g g^T with gamma = c (2 - c G_kk).  With beta = 1/(1 + sqrt(err/kdenom)),
//   c = 1 / ((M-1) (kdenom + sqrt(err) sqrt(kdenom))):
// one rsq with a Newton step for sqrt(kdenom), one rcp with a cubic step -- 12 dependent operations instead of the 20 that
// form 1/kdenom and beta separately (the forwarder wave still forms those two, off the chain, for the records).
__device__ __forceinline__ double gain_c(double Gkk, double invM, double errk, double sqk, double rM1) {
  const double v = __builtin_fma(Gkk, invM, errk);              // kdenom = var + err  (:69, :91)
  const double q0 = __builtin_amdgcn_rsq(v);
  const double r0 = v * q0;                                     // ~ sqrt(v)
  const double er = __builtin_fma(-r0, r0, v);
  const double r = __builtin_fma(er, 0.5 * q0, r0);             // sqrt(v), one Newton step
  const double u = __builtin_fma(sqk, r, v);                    // kdenom (1 + sqrt(err / kdenom)) = kdenom / beta
  const double c0 = __builtin_amdgcn_rcp(u);
  const double eu = __builtin_fma(-u, c0, 1.0);
  return __builtin_fma(c0, __builtin_fma(eu, eu, eu), c0) * rM1;
}
__device__ __forceinline__ double rl(double v, int lane) {  // value held by `lane` (wave-uniform index)
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}


#ifndef EFA_DEFER_GRAM
#define EFA_DEFER_GRAM 1
#endif
template <int NC, bool GC = false>
struct BandShape {
  static constexpr int PAD = 2 * PLg * NC;
  static constexpr int TS = PAD + kTrajScalars;
  static constexpr int TSR = TS + (GC ? kRowsWG : 0);  // ring slot: the record, then (GC) the taper of its ob against this workgroup's 64 rows
  static constexpr int SP = PAD + ((2 - PAD % 32) + 32) % 32;  // park tile row stride == 2 (mod 32): conflict-free operand reads
  static constexpr int UREG = (kRowsWG * SP > kRowsWG * (2 * kRowsWG + kScStride + 1)) ? kRowsWG * SP : kRowsWG * (2 * kRowsWG + kScStride + 1);
  static constexpr int kLinv = kNBands * kBand * kBand;  // L^-1 of every band, [band][t][s]
  static constexpr int kTw = GC ? kRowsWG * kRowsWG : 0;  // the block's 64 x 64 corner of the obs-obs taper table (Gaspari-Cohn cycles)
  // DEFERRED GRAM (unlocalised cycles): only tile row 0 of G is formed between the block-start barriers; the pivot wave starts
  // on it while tile rows 1..3 are formed on the other three SIMDs.  The parked tile those products read shares its LDS with
  // the step records, so the records of the first kSgE steps live in an area of their own.
  static constexpr bool kDefer = !GC && NC <= 13 && (EFA_DEFER_GRAM != 0);  // (14 chunks and more: the extra registers would spill)
  static constexpr int kSgE = kDefer ? 16 : 0;               // steps whose records live outside the union
  static size_t lds_doubles() { return (size_t)kRingG * TSR + kRowsWG * kRowsWG + UREG + 3 * kRowsWG + kLinv + kTw + (size_t)kSgE * kRowsWG * 2; }
  static size_t lds_bytes() { return lds_doubles() * sizeof(double) + 32 * sizeof(int); }
};

template <int NC, bool GC>
__global__ __launch_bounds__(kGT) void k_pipe_band(const PipeArgs a) {
  using Sh = BandShape<NC, GC>;
  constexpr int PAD = Sh::PAD, TS = Sh::TS, TSR = Sh::TSR, SP = Sh::SP, UREG = Sh::UREG;
  constexpr int kEarly = GC ? kEarlyGC : kEarlyNone;
  constexpr int EPL = (TS + 63) / 64;
  constexpr int NJ = (PAD + 15) / 16;  // accumulator tiles per vector wave in the block's matrix-core layout
  extern __shared__ __align__(16) double lds[];
  double* ring = lds;                          // [kRingG][TSR]  ye rows (+ scalars, + GC taper column in follower mode)
  double* G_s = ring + kRingG * TSR;            // [64][64]       Gram matrix of the block; later the rows handed to the pivot
  double* U = G_s + kRowsWG * kRowsWG;         // union: Yt[64][SP]  then  {g, kb}[64][64], sc[64][8]
  double* pm = U + UREG;                       // [3][64]        parked row means / obs-space means / squared row norms
  double* LinvA = pm + 3 * kRowsWG;            // [16 bands][4][4]  L^-1 of each band: LinvA[b][t][s] = (L^-1)[s][t]
  double* tw_s = LinvA + Sh::kLinv;            // [64][64]       GC: taper of the block's obs against the block's rows
  double2* sgk_e = reinterpret_cast<double2*>(tw_s + Sh::kTw);  // [kSgE][64]  step records of the first steps (deferred Gram)
  int* ctl = reinterpret_cast<int*>(reinterpret_cast<double*>(sgk_e) + (size_t)Sh::kSgE * kRowsWG * 2);  // [32]
  double* Yt = U;
  double2* s_gk = reinterpret_cast<double2*>(U);  // [step][row] = {G_kj, kb_j}
  constexpr bool DEFER = Sh::kDefer;
  constexpr int kPollG = GC ? 4 : kPollGMax;
  // record of step st (per-lane or uniform st)
  auto SG = [&](int st) -> double2* { return (DEFER && st < Sh::kSgE) ? sgk_e + (size_t)st * kRowsWG : s_gk + (size_t)st * kRowsWG; };

  const int tid = threadIdx.x;
  // wave roles: 0-3 vector, 4 pivot, 5-6 G waves, 7 loader
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int M = a.M;
  const long P = a.P, R = a.R;
  const long own0 = (long)blockIdx.x * kRowsWG;
  const long own1 = (own0 + kRowsWG < P) ? own0 + kRowsWG : (own0 < P ? P : own0);
  const int nb = (int)(own1 - own0);  // obs this workgroup leads (0: it only follows)
  const bool leads = nb > 0;
  const int nbands = (nb + kBand - 1) / kBand;
  const double rM1 = 1.0 / (double)(M - 1);
  const double invM = 1.0 / (double)M;

  if (tid < 32) ctl[tid] = (tid >= cProg && tid < cProg + 4) ? -1 : (tid == cFwd ? (int)(own0 - 1) : 0);
  __syncthreads();

#ifdef EFA_PIPE_BLOCKTIME  /* make diag: three stamps per block, none inside a loop */
#define EFA_BLOCKSTAMP(cond, slot)                                                                  \
  do {                                                                                              \
    if (a.dbg != nullptr && (cond)) a.dbg[(size_t)own0 * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define EFA_BLOCKSTAMP(cond, slot) \
  do {                              \
  } while (0)
#endif
#ifdef EFA_PIPE_BLOCKTIME
#define EFA_WAIT_OUT(cond, row, slot, v)                                         \
  do {                                                                            \
    if (a.dbg != nullptr && (cond)) a.dbg[(size_t)(own0 + (row)) * 8 + (slot)] = (u64)(v); \
  } while (0)
#else
#define EFA_WAIT_OUT(cond, row, slot, v) do { } while (0)
#endif
#if defined(EFA_PIPE_STAMPS) || defined(EFA_PIPE_BLOCKTIME) || defined(EFA_PIPE_PIVSTAMP)
#define EFA_EXP(bit) ((a.debug & (bit)) != 0)  /* timing experiments (diagnostic builds only): results are wrong */
#else
#define EFA_EXP(bit) false
#endif
#ifdef EFA_PIPE_PIVSTAMP
#define EFA_HO(row, slot) do { if (a.dbg != nullptr && lane == 0 && own0 + (row) < P) a.dbg[(size_t)(own0 + (row)) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define EFA_HO_IF(cond, row, slot) do { if (cond) EFA_HO(row, slot); } while (0)
#else
#define EFA_HO(row, slot) do { } while (0)
#define EFA_HO_IF(cond, row, slot) do { } while (0)  /* (nothing of the condition either: dead tests perturbed the register allocation) */
#endif
#if defined(EFA_PIPE_PIVSTAMP) && !defined(EFA_PIPE_HOSTAMP_ONLY)  /* make pivstamp: wait / work accounting of the pivot and G waves only (they have registers to spare) */
#define EFA_PS_NOW() __builtin_amdgcn_s_memtime()
#define EFA_PS(stmt) stmt
#else  /* (make hostamp: only the eight s_memrealtime stamps of the hand-over chain per block -- next to no perturbation) */
#define EFA_PS_NOW() 0ull
#define EFA_PS(stmt)
#endif
  long budget = a.spin_limit;
  // every spin is bounded twice: by a poll budget and by wall time (s_memrealtime, 100 MHz), looked at only on the
  // slow side of a poll loop
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
          EFA_PS(if (a.dbg != nullptr && lane == 0) { u64* d = a.dbg + (size_t)(32 + wave) * 8; d[0] = 1000 + (u64)(word - ctl); d[1] = (u64)thr; d[2] = (u64)g_ctl(word); d[3] = (u64)own0; })
          give_up();
          return false;
        }
      }
      if (doze) __builtin_amdgcn_s_sleep(1);
    }
    return true;
  };
  // both G waves' hand-over words (adjacent ints, 8-byte aligned) in ONE LDS read per poll
  auto wait_gt2 = [&](const int* word2, int thr) {
    for (;;) {
      const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(word2), __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("" ::: "memory");
      const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffull)), hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
      if (lo > thr && hi > thr) return true;
      if ((++polls & 15) == 0) {
        if (g_ctl(&ctl[cBail]) != 0) return false;
        budget -= 16;
        if (budget <= 0 || EFA_TIMED_OUT()) {
          give_up();
          return false;
        }
      }
    }
  };
  // least-advanced consumer of the ye ring: the 4 vector waves and the forwarder
  auto min_prog = [&]() {
    int mn = g_ctl_lane(&ctl[cProg + (lane & 3)]);  // (the forwarder reads the ring no more: the four vector waves are its only consumers)
    mn = min(mn, __builtin_amdgcn_mov_dpp(mn, 0xB1, 0xF, 0xF, true));
    mn = min(mn, __builtin_amdgcn_mov_dpp(mn, 0x4E, 0xF, 0xF, true));
    return __builtin_amdgcn_readfirstlane(mn);
  };

  // one 16 x 16 tile of G = Y Y^T from the parked tile, stored at its place
  auto gram_tile = [&](int I, int J) {
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    const double* pa = Yt + (size_t)(16 * I + (lane & 15)) * SP + (lane >> 4);
    const double* pb = Yt + (size_t)(16 * J + (lane & 15)) * SP + (lane >> 4);
    for (int s = 0; s < PAD / 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * s], pb[4 * s], acc, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 4; ++v) G_s[(16 * I + 4 * v + (lane >> 4)) * kRowsWG + 16 * J + (lane & 15)] = acc[v];
  };
  // deferred Gram: tile rows 1..3 after the second block-start barrier, on the three SIMDs the pivot wave is not on
  // (waves 1, 2: four tiles each; waves 3 and 7: two each -- 104 matrix-core steps per SIMD); cDef counts finished tiles
  auto gram_deferred = [&]() {
    int n = 0;
    if (wave == 1 || wave == 2) {
      for (int J = 0; J < 4; ++J) gram_tile(wave, J);
      n = 4;
    } else if (wave == 3) {
      gram_tile(3, 0);
      gram_tile(3, 1);
      n = 2;
    } else if (wave == 7) {
      gram_tile(3, 2);
      gram_tile(3, 3);
      n = 2;
    }
    asm volatile("" ::: "memory");
    if (n && lane == 0) __hip_atomic_fetch_add(&ctl[cDef], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  // G = Y Y^T for the block's rows, every wave takes two 16 x 16 tiles (between the block-start barriers)
  auto form_gram = [&]() {
    const int I = wave >> 1, J0 = (wave & 1) * 2;
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
  };

  // ======================================================================================
  // wave 7: loader (follower mode) / forwarder (leader mode)
  // ======================================================================================
  if (wave == 7) {
    // Three phases, each its own loop: follow the records before the block, forward the block, follow the
    // rest (as ONE loop the compiler waits vmcnt(0) per forwarded record: see efa_pipeline.hip).
    bool failed = false;
    int barriers_left = leads ? 3 : 0;
    auto follow = [&](long next, const long limit) {
      while (next < limit && !failed) {
        const int nrec = (int)((limit - next < kPollG) ? (limit - next) : kPollG);
        u64 v[kPollG][EPL];
        double twv[kPollG];  // GC: taper of each polled ob against this workgroup's row own0 + lane
#pragma unroll
        for (int d = 0; d < kPollG; ++d) {
          const long kk = next + ((d < nrec) ? d : nrec - 1);
          twv[d] = (GC && own0 + lane < R) ? a.tw[(size_t)kk * R + own0 + lane] : 0.0;
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
          // Nothing yet: wait on ONE word -- the last ye word of the first missing record, which its owner wave stores with its
          // last store instruction -- and poll the full records again when it has arrived.  A full poll is sixteen loads per
          // lane and a global round trip; missing the records by a moment cost a whole such period (light stamps: the next
          // leader's loader saw the last records 5 k cycles after they had been stored).
          const u64* probe = a.traj + (size_t)next * TS + (PAD - 1);
          for (;;) {
            if (--budget <= 0 || __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                ((budget & 15) == 0 && EFA_TIMED_OUT())) {
              failed = true;
              break;
            }
            const u64 w = g_traj_load(probe);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(w & 0xffffffffull));
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(w >> 32));
            if ((((u64)hi << 32) | lo) != kTrajSentinel) break;
            __builtin_amdgcn_s_sleep(1);
          }
          continue;
        }
        EFA_HO_IF(leads && next + cnt == own0 && limit == own0, 8, 7);  // T2a: this (next) leader's loader has SEEN the last foreign record complete
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
            double* slot = ring + (size_t)((next + d) % kRingG) * TSR;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
              const int idx = lane + 64 * e;
              if (idx < TS) slot[idx] = __longlong_as_double((long long)v[d][e]);
            }
            if (GC) slot[TS + lane] = twv[d];
          }
        }
        next += cnt;
        if (lane == 0) g_ctl_set(&ctl[cReady], (int)next);
      }
    };
    follow(0, (own0 < P) ? own0 : P);
    EFA_HO_IF(leads, 8, 0);   // T2: the last foreign record is in this workgroup's ring
    EFA_BLOCKSTAMP(lane == 0 && leads, 3);
    EFA_WAIT_OUT(lane == 0 && leads, 4, 6, __builtin_amdgcn_s_memrealtime());  // 100 MHz, comparable across workgroups
#ifdef EFA_PIPE_BLOCKTIME
    {
      const int mp = min_prog();  // all lanes: the quad minimum goes through DPP
      EFA_WAIT_OUT(lane == 0 && leads, 3, 5, (long)(own0 - 1) - (long)mp);  // records still to be applied when the last one is in the ring
    }
#endif
    if (leads && !failed) {
      __syncthreads();  // B1: the vector waves have parked their rows in the tile
      if (DEFER) gram_tile(0, 3);
      else form_gram();
      __syncthreads();  // B2: G is complete (deferred Gram: its tile row 0)
      barriers_left = 1;
      if (DEFER) gram_deferred();
      // This wave also carries the obs-space MEANS of the block's 64 rows (lane j = row j) and everything that
      // hangs on them -- innovation, mean update (:85, :130), the obs' diagnostics -- from the pivot wave's records:
      // none of it is on the serial chain, and every instruction the pivot wave does not issue shortens a step.
      double xmv = pm[kRowsWG + lane];
      const bool f_ob = lane < nb;
      const double val_l = f_ob ? a.ob_value[own0 + lane] : 0.0;
      const double2 ec_l = f_ob ? *reinterpret_cast<const double2*>(a.ob_errsq + 4 * (own0 + lane)) : make_double2(1.0, 1.0);
      const bool my_asm = f_ob ? (a.ob_assim[own0 + lane] != 0) : false;
      // The pivot wave's two guards are kept HERE, off the serial chain (this wave reads every step's G_kk anyway; a tripped
      // guard abandons the launch, nothing produced meanwhile is used, so it does not matter that it trips a band later).
      // Cancellation: an assimilated pivot whose G_kk fell below 1e-3 of its value at block start.
      // Centring: np.var re-centres (:69), var = G_kk / M - mean^2; the rows are mean-removed perturbations (assimilation.py:47,
      // :147), their means are rounding residue and mean^2 changes no bit of var or kdenom, so the chain does not carry the
      // means; a block whose pivot rows are NOT centred (mean^2 above 1e-22 of var or of the error variance) goes to the
      // vector-chain kernel, which computes np.var as written.
      const double gjj0 = DEFER ? pm[2 * kRowsWG + lane] : G_s[lane * kRowsWG + lane];  // |y_j|^2 at block start
      const double thr = my_asm ? 1e-3 * gjj0 : -1.0;
      bool bad = f_ob && !(pm[lane] * pm[lane] <= 1e-22 * fmin(ec_l.x, gjj0 * invM));  // pm[0][j]: the member mean of row j
      double l_xm = 0.0, l_innov = 0.0;
      double l_var = 0.0, l_rd = 0.0, l_be = 0.0;  // this lane's ob: prior variance, 1/kdenom, beta -- from G_kk at its step
      // Round 3: the ye rows of a record go to global memory straight from the vector wave that forms them (below, "owner"), so
      // this wave stores only the record's four scalars -- and those hang on the pivot's step records alone, not on YE: it runs
      // right behind the pivot (cSReady) instead of behind the YE tiles, touches the ring no more, and publishes a band's scalars
      // with four stores from the band's own lanes.  (The followers validate every word of a record by itself, so its two parts
      // may arrive in either order.)  Before, this wave read every ye row back from the ring and stored it: as slow as the pivot,
      // and its backlog was the first 4-7 k cycles of every hand-over.
      u64* const rec_l = a.traj + (size_t)(own0 + (f_ob ? lane : 0)) * TS + PAD;  // this lane's ob: its record's scalars
      for (int b = 0; b < nbands && !failed; ++b) {
        const int s1 = (nb - kBand * b < kBand) ? nb - kBand * b : kBand;
        if (!wait_gt(&ctl[cSReady], kBand * b + s1 - 1, true)) {  // the pivot has recorded the band's steps
          failed = true;
          break;
        }
        const bool mine = (lane >> 2) == b;  // kBand == 4
        {  // the band's obs, one per lane: the scalars the pivot wave used at their steps, recomputed from the same G_kk
          const double Gkk = SG(lane & 63)[lane].x;
          double rd, be;
          gain_scalars(Gkk, invM, ec_l.x, ec_l.y, rd, be);
          bad = bad || (mine && f_ob && !(Gkk > thr));  // (a block's last band may be partial)
          l_rd = mine ? rd : l_rd;
          l_be = mine ? be : l_be;
          l_var = mine ? Gkk * invM : l_var;                                   // np.var, ddof = 0 (:69, :70): the rows are centred
        }
        // Every LDS operand of the band first (one round trip for its up to four records), then the serial mean chain
        double2 gk4[kBand];
        double tw4[kBand];
#pragma unroll
        for (int s = 0; s < kBand; ++s) {
          const int st = kBand * b + ((s < s1) ? s : 0);
          gk4[s] = SG(st)[lane];                                  // G_kj, kb_j of this lane's row
          tw4[s] = GC ? tw_s[st * kRowsWG + lane] : 1.0;
        }
        const double rd_a = my_asm ? l_rd : 0.0;                  // :74: an ob that is not assimilated moves no mean
#pragma unroll
        for (int s = 0; s < kBand; ++s) {
          if (s >= s1) break;  // wave-uniform
          const int st = kBand * b + s;
          const double dv = val_l - xmv;                                       // :85 in the ob's own lane
          const double innov = rl(dv, st);
          const double rden_a = rl(rd_a, st);
          double kc = gk4[s].x * rM1;                                          // :95
          if (GC) kc = tw4[s] * kc;                                            // :115
          const double km = kc * rden_a;                                       // :119
          l_xm = (lane == st) ? xmv : l_xm;                                    // this lane's ob: its prior mean (:66)
          l_innov = (lane == st) ? dv : l_innov;
          xmv = xmv + km * innov;                                              // :130
        }
        // the records' scalars, from the band's own lanes.  The followers read two: c = beta / ((M-1) kdenom), 0 for an ob that
        // is not assimilated, and m = innov / ((M-1) kdenom): kb_j = w c (y_j . ye), xm_j += w m (y_j . ye)
        // (:95, :115, :119, :130, :136 folded once per ob); then innov and the assimilate flag
        if (mine && f_ob) {
          const double cf = my_asm ? (l_be * l_rd) * rM1 : 0.0, mf = (l_rd * rM1) * l_innov;
          g_traj_store(rec_l + 0, cf);
          g_traj_store(rec_l + 1, mf);
          g_traj_store(rec_l + 2, l_innov);
          const double actv = my_asm ? 1.0 : 0.0;
#pragma unroll
          for (int sj = 3; sj < TS - PAD; ++sj) g_traj_store(rec_l + sj, actv);  // (the followers wait for every word of a record)
        }
      }
      // (not when the launch is being abandoned anyway: then this block may have run on rows that were never parked)
      if (!failed && __ballot(bad) != 0ull && g_ctl(&ctl[cBail]) == 0 &&
          __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {  // abandon the launch (status[2]: the host re-runs
        if (lane == 0) {                                                                  // Phase A with the vector-chain kernel)
          give_up();
          __hip_atomic_store(a.status + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      // The obs' diagnostics and sweep coefficients, once per block and one ob per lane, from what is in LDS anyway:
      // km of the ob's own row is kc rden with kc = G_kk/(M-1), and that row is scaled by (1 - kb_k)  (:144-149).
      if (!failed && f_ob) {
        const long f = own0 + lane;
        const double2 gk = SG(lane)[lane];                                     // G_kk, kb_k at the ob's own step
        const double2 rb = make_double2(l_rd, l_be);                           // rden, beta
        const double var = l_var;
        const bool act = my_asm;
        const double innov = val_l - l_xm;                                     // :85
        a.prior_mean[f] = l_xm;                                                // :66
        a.prior_var[f] = var;                                                  // :70
        double* ck = a.coef + (size_t)f * kCoefStride;
        *reinterpret_cast<double2*>(ck) = make_double2(act ? innov : 0.0, act ? rb.x : 0.0);
        *reinterpret_cast<double2*>(ck + 2) = make_double2(act ? rb.y : 0.0, act ? 1.0 : 0.0);
        a.assimilated[f] = act ? 1 : 0;                                        // :74-76, :149
        if (act) {
          const double km = (gk.x * rM1) * rb.x;                               // :95, :119
          const double fsc = 1.0 - gk.y;
          a.post_mean[f] = l_xm + km * innov;                                  // :130
          a.post_var[f] = (fsc * fsc) * var;
        }
      }
      EFA_HO(8, 1);   // T1: this block's last record has been forwarded
      pm[kRowsWG + lane] = xmv;  // obs-space means of all 64 rows after the block, back to the vector waves
      EFA_BLOCKSTAMP(lane == 0, 2);
      EFA_WAIT_OUT(lane == 0, 4, 5, __builtin_amdgcn_s_memrealtime());
      __syncthreads();  // B3: done with the pivot's records
      barriers_left = 0;
      if (!failed) follow(own1, P);
    }
    if (failed && lane == 0) give_up();
    for (; barriers_left > 0; --barriers_left) __syncthreads();  // never leave the others at a barrier
    return;
  }

  // ======================================================================================
  // wave 4 (pivot) and waves 5, 6 (G waves): the Gram-space recurrence of this workgroup's block
  // ======================================================================================
  if (wave >= kVW) {
    if (!leads) return;
    if (GC) {  // the three waves that only wait here fetch the block's corner of the obs-obs taper table (ensrf.py:99-115 on the obs rows)
      for (int i = (wave - kVW) * 64 + lane; i < kRowsWG * kRowsWG; i += 3 * 64) {
        const long kg = own0 + (i >> 6), rg = own0 + (i & 63);
        tw_s[i] = (kg < P && rg < R) ? a.tw[(size_t)kg * R + rg] : 1.0;
      }
    }
    __syncthreads();  // B1
    EFA_HO_IF(wave == kVW, 8, 2);   // T3: every vector wave has parked its rows
    if (DEFER) gram_tile(0, wave - kVW);
    else form_gram();
    __syncthreads();  // B2
    EFA_HO_IF(wave == kVW, 8, 3);   // T4: G is complete
    if (wave == kVW) {
      // ---------------- pivot wave: lane j <-> column j of G ----------------
      __builtin_amdgcn_s_setprio(3);  // the serial chain: ahead of the vector wave that shares its SIMD
      // One wave issues in order, so what a step costs is its instruction count: per-ob constants live in the
      // ob's lane (v_readlane), the band's later rows are downdated in registers, nothing is read from LDS
      // inside a band, and the step's record is written at its end.  Gain chain as in efa_pipeline_gram.hip:
      //   kdenom -> q0 = rsq(kdenom) -> { Newton step of q  ||  beta0 = 1/(1 + sqrt(err) q0) } -> beta
      // (the cancellation and centring guards of this chain are kept by the forwarder wave, which reads every G_kk anyway)
      bool ok = true;
      double band[kBand];
#pragma unroll
      for (int s = 0; s < kBand; ++s) band[s] = G_s[s * kRowsWG + lane];
      // Every value that came from LDS is pinned HERE: the compiler waits for a load where its result is first used;
      // left inside the step code such a wait would be executed every step.
#define EFA_PIN_BAND(a) _Pragma("unroll") for (int pin_i = 0; pin_i < kBand; ++pin_i) asm volatile("" : "+v"(a[pin_i]))
      EFA_PIN_BAND(band);
      // {error, sqrt(error), assimilate} of the band's four obs: wave-uniform loads one band ahead (no v_readlane on the chain)
      const double* ecp = a.ob_errsq + 4 * own0;
      double ec_nx[kBand][3];
#pragma unroll
      for (int s = 0; s < kBand; ++s) {
        const int o = (s < nb) ? s : 0;
        const double2 e01 = *reinterpret_cast<const double2*>(ecp + 4 * o);
        ec_nx[s][0] = e01.x;
        ec_nx[s][1] = e01.y;
        ec_nx[s][2] = ecp[4 * o + 2];
      }
      double gprev[kBand - kEarly], gamprev[kBand - kEarly];  // rows and gammas of the previous band's late steps
      double kbprev[kBand - kEarly], tprev[kBand - kEarly];   // GC: their gains kb_j and t_j = G_kj - kb_j G_kk
      EFA_BLOCKSTAMP(lane == 0, 0);
      EFA_WAIT_OUT(lane == 0, 5, 5, __builtin_amdgcn_s_memrealtime());
      EFA_HO(8, 4);   // T5: the pivot starts band 0
      EFA_PS(u64 ps_wait = 0; const u64 ps_t0 = EFA_PS_NOW(); u64 ps_ld = 0; u64 ps_ap = 0; u64 ps_st = 0; u64 ps_en = 0; u64 ps_hd = 0;)
      for (int b = 0; b < nbands && ok; ++b) {
        EFA_PS(const u64 ps_top = EFA_PS_NOW();)
        const int r0 = kBand * b;
        const int s1 = (nb - r0 < kBand) ? nb - r0 : kBand;  // steps of this band
        double ec[kBand][3];
#pragma unroll
        for (int s = 0; s < kBand; ++s) {
          ec[s][0] = ec_nx[s][0];
          ec[s][1] = ec_nx[s][1];
          ec[s][2] = ec_nx[s][2];
          const int o = (r0 + kBand + s < nb) ? r0 + kBand + s : 0;
          const double2 e01 = *reinterpret_cast<const double2*>(ecp + 4 * o);
          ec_nx[s][0] = e01.x;
          ec_nx[s][1] = e01.y;
          ec_nx[s][2] = ecp[4 * o + 2];
        }
        if (DEFER && r0 == Sh::kSgE) {  // the step records move into the union from here: the parked tile must be done with
          ok = wait_gt(&ctl[cDef], 11, false);
          if (!ok) break;
        }
        if (b > 0) {  // the band's rows, current through the previous band, from the two G waves
          EFA_PS(const u64 ps_a = EFA_PS_NOW();)
          if (!EFA_EXP(2048) && (kEarly > 0 || b >= 2)) ok = wait_gt2(&ctl[cBandH], b - 1);  // kEarly == 0: band 1's rows are the initial ones
          if (!ok) break;
#pragma unroll
          for (int s = 0; s < kBand; ++s) band[s] = G_s[(r0 + s) * kRowsWG + lane];
          EFA_PS(const u64 ps_r = EFA_PS_NOW(); ps_wait += ps_r - ps_a;
                 if (a.dbg != nullptr && lane == 0 && own0 + 64 + b < P) { a.dbg[(size_t)(own0 + 64 + b) * 8 + 1] = ps_a; a.dbg[(size_t)(own0 + 64 + b) * 8 + 2] = ps_r; })
          EFA_PIN_BAND(band);
          EFA_PS(const u64 ps_l = EFA_PS_NOW(); ps_ld += ps_l - ps_r; ps_hd += ps_a - ps_top;)
          // the G waves handed these rows over EARLY, current through the first kEarly steps of the previous band
          // (so that this wave never waits for them); the rest of that band is applied here, in order.
#pragma unroll
          for (int o = 0; o < kBand - kEarly; ++o) {
            // the four scalars of a step first, then the four updates: a v_readlane result needs two wait states before a
            // vector instruction may read it, and back to back (readlane pair, FMA, readlane pair, FMA ...) every FMA paid them
            double gi[kBand], kbi[kBand];
#pragma unroll
            for (int s2 = 0; s2 < kBand; ++s2) {
              gi[s2] = rl(gprev[o], r0 + s2);
              kbi[s2] = GC ? rl(kbprev[o], r0 + s2) : 0.0;
            }
#pragma unroll
            for (int s2 = 0; s2 < kBand; ++s2) {
              if (GC) {  // the taper makes the downdate two-term: G_ij -= kb_j G_ki + kb_i t_j
                band[s2] = __builtin_fma(-gi[s2], kbprev[o], band[s2]);
                band[s2] = __builtin_fma(-kbi[s2], tprev[o], band[s2]);
              } else {
                band[s2] = __builtin_fma(-gi[s2], gamprev[o], band[s2]);  // gamprev holds the VECTOR gamma g: one product per step, not per row
              }
            }
          }
          EFA_PS(EFA_PIN_BAND(band); ps_ap += EFA_PS_NOW() - ps_l;)
        }
        EFA_PS(const u64 ps_s0 = EFA_PS_NOW();)
        // (L^-1)[s][t] for this band, lane t holds column t (lanes >= 8 carry zeros): right-looking,
        // linv[s'] -= L[s'][s] linv[s] once row s is final
        double linv[kBand];
#pragma unroll
        for (int s = 0; s < kBand; ++s) linv[s] = (lane == s) ? 1.0 : 0.0;
        double tw4[kBand];  // GC: taper of the band's obs against this lane's row, fetched once per band
#pragma unroll
        for (int s = 0; s < kBand; ++s) tw4[s] = GC ? tw_s[(r0 + (s < s1 ? s : 0)) * kRowsWG + lane] : 1.0;
        if (GC) EFA_PIN_BAND(tw4);
        // One wave issues in order and every instruction -- scalar ones too -- takes an issue slot: half of what this loop
        // issued was scalar bookkeeping (the assimilate bit and the guard's bit mask per step, the record's address, the
        // tests for a partial band).  The assimilate flag now comes with the ob's constants, the guards live in the forwarder
        // wave, the band's records have one base address, and a full band (every band but a block's last, at most) runs a
        // copy of the steps without the partial-band tests.
        double2* const recb = SG(r0) + lane;  // step r0 + s writes recb[s * kRowsWG] (a band never straddles the records' two areas)
        auto run_steps = [&](auto full_tag) {
          constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
          for (int s = 0; s < kBand; ++s) {
            if (FULL || s < s1) {  // wave-uniform
              const int kk = r0 + s;
              const double g = band[s];
              const double Gkk = rl(g, kk);
              double cc = gain_c(Gkk, invM, ec[s][0], ec[s][1], rM1);       // beta / ((M-1) kdenom)  (:95, :119, :135, :136)
              cc = (ec[s][2] != 0.0) ? cc : 0.0;                            // :74: an ob that is not assimilated changes nothing
              double kb = cc * g;                                           // the rows' gains (:136)
              if (GC) kb = tw4[s] * kb;                                     // :115
              // this step's rank-one downdate of G is gamma g g^T, gamma = c (2 - c G_kk) with c = kb_j / G_kj: the form
              // the band's later rows (and L, below) use -- one v_readlane pair per row instead of two
              const double gam = cc * __builtin_fma(-cc, Gkk, 2.0);
              const double tj = GC ? __builtin_fma(-kb, Gkk, g) : 0.0;      // t_j = G_kj - kb_j G_kk
              const double gg = gam * g;                                    // the step's downdate of row i is G_ki (gamma g)
              double gi[kBand], kbi[kBand];                                  // (all the scalars first: see the note on wait states above)
#pragma unroll
              for (int s2 = s + 1; s2 < kBand; ++s2) {
                gi[s2] = rl(g, r0 + s2);                                    // G_k,i of row i = r0 + s2
                kbi[s2] = GC ? rl(kb, r0 + s2) : 0.0;                       // GC: kb_j = w_kj c G_kj is no longer a multiple of G_kj
              }
#pragma unroll
              for (int s2 = s + 1; s2 < kBand; ++s2) {
                if (GC) {  // the two-term form, kb_i by v_readlane
                  band[s2] = __builtin_fma(-gi[s2], kb, band[s2]);
                  band[s2] = __builtin_fma(-kbi[s2], tj, band[s2]);
                  linv[s2] = __builtin_fma(-kbi[s2], linv[s], linv[s2]);    // L[s2][s] = kb_i
                } else {
                  band[s2] = __builtin_fma(-gi[s2], gg, band[s2]);
                  linv[s2] = __builtin_fma(-(cc * gi[s2]), linv[s], linv[s2]);  // L[s2][s] = kb_i = c G_k,i
                }
              }
              recb[s * kRowsWG] = make_double2(g, kb);         // the step's record: {G_kj, kb_j} per row
              if (s == kEarly - 1 && lane == 0) g_ctl_set(&ctl[cHalf], 2 * b + 1);  // the G waves may start on the next band's rows
              EFA_PS(if (s == kEarly - 1 && a.dbg != nullptr && lane == 0 && own0 + 64 + b < P) a.dbg[(size_t)(own0 + 64 + b) * 8 + 0] = EFA_PS_NOW();)
              if (s >= kEarly) {
                gprev[s - kEarly] = g;
                gamprev[s - kEarly] = gg;
                if (GC) {
                  kbprev[s - kEarly] = kb;
                  tprev[s - kEarly] = tj;
                }
              }
            }
          }
          // per band: L^-1 (LinvA[b][t][s], zero where s < t or s >= 8)
          if (lane < kBand) {
            double* dst = LinvA + ((size_t)b * kBand + lane) * kBand;
#pragma unroll
            for (int s = 0; s < kBand; ++s) dst[s] = (FULL || s < s1) ? linv[s] : 0.0;
          }
        };
        if (s1 == kBand) run_steps(std::true_type());
        else run_steps(std::false_type());
        EFA_PS(EFA_PIN_BAND(band); const u64 ps_s1 = EFA_PS_NOW(); ps_st += ps_s1 - ps_s0;)
        if (lane == 0) {
          g_ctl_set(&ctl[cSReady], r0 + s1);
          g_ctl_set(&ctl[cLinv], b + 1);
        }
        EFA_HO_IF((b & 3) == 3, 14, b >> 2);  // the pivot is through with bands 3, 7, 11, 15
        EFA_PS(ps_en += EFA_PS_NOW() - ps_s1;)
      }
      EFA_PS(if (a.dbg != nullptr && lane == 0 && own0 + 16 < P) {
        u64* d = a.dbg + (size_t)(own0 + 16) * 8;
        d[0] = ps_ld; d[1] = ps_ap; d[2] = ps_st; d[3] = ps_en; d[4] = ps_hd;
      })
      EFA_HO(8, 5);   // T0: the pivot has finished the block's last step
      EFA_PS(if (a.dbg != nullptr && lane == 0) {
        a.dbg[(size_t)own0 * 8 + 0] = EFA_PS_NOW() - ps_t0;
        a.dbg[(size_t)own0 * 8 + 1] = ps_wait;
      })
      EFA_BLOCKSTAMP(lane == 0, 1);
      __syncthreads();  // B3
      return;
    }
    // ---------------- G waves: G as accumulator tiles, one rank-8 update per band ----------------
    // G wave h holds tile columns J = 2h, 2h+1 of the 4 x 4 tile grid: acc[I][jj][v] in lane l is
    // G[16 I + 4 v + l/16][16 (2h + jj) + l%16].  After band b (steps r0 .. r0+s1-1):
    //   G_ij -= sum_s kb^(s)_j G^(s)_ki + kb^(s)_i t^(s)_j,   t_j = G_kj - kb_j G_kk   (what y_i -= kb_i ye_k does to the dots)
    // Only tile rows that still contain rows of later bands are kept current, the tile row of the next band
    // first: its 8 rows go back to G_s for the pivot wave.
    const int h = wave - kVW - 1;  // 0, 1
    const int lr = lane >> 4, lc = lane & 15;
    v4f64 acc[4][2];
    auto load_acc = [&](int I0, int I1) {
#pragma unroll
      for (int I = 0; I < 4; ++I)
        if (I >= I0 && I < I1)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[I][jj][v] = G_s[(16 * I + 4 * v + lr) * kRowsWG + 16 * (2 * h + jj) + lc];
    };
    static_assert(!DEFER || kEarly == 0, "the deferred Gram is written for the one-band-behind G waves");
    if (DEFER) __builtin_amdgcn_s_setprio(2);  // ahead of the vector wave on this SIMD, which forms deferred Gram tiles back to back
    load_acc(0, DEFER ? 1 : 4);  // deferred Gram: tile rows 1..3 are not there yet
    bool rows_ready = !DEFER;
    EFA_PS(u64 ps_w1 = 0; u64 ps_p1 = 0; u64 ps_w2 = 0; u64 ps_p2 = 0; u64 ps_l1 = 0;)
    // From the records {G_kj, kb_j} alone: G_ij -= kb_j G_ki + kb_i (G_kj - kb_j G_kk), two products per K slice of
    // four steps: (A1 = -G_ki, B1 = kb_j) and (A2 = -kb_i, B2 = t_j).  Steps outside [lo, hi) of the band are masked.
    // Without localisation kb_j = c G_kj for every row, and the two terms collapse into ONE product per K slice:
    // G_ij -= gamma G_ki G_kj, gamma = c (2 - c G_kk), c = kb_k / G_kk from the step's own diagonal record.
    double b1[kBand / 4][2], b2[kBand / 4][2];
    auto load_b = [&](int r0, int lo, int hi) {
#pragma unroll
      for (int q = 0; q < kBand / 4; ++q) {
        const int sb = 4 * q + lr;                            // this lane's K slot: step r0 + sb
        const bool on = sb >= lo && sb < hi;
        const int st = r0 + (on ? sb : 0);
        const double2 dg = SG(st)[st];                        // G_kk, kb_k
        const double Gkk = dg.x;
        double gam = 0.0;
        if (!GC) {
          const double cc = (Gkk > 0.0) ? dg.y * g_rcp(Gkk) : 0.0;  // kb_k = 0 for an ob that is not assimilated
          gam = cc * __builtin_fma(-cc, Gkk, 2.0);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const double2 r = SG(st)[16 * (2 * h + jj) + lc];
          if (GC) {
            b1[q][jj] = on ? r.y : 0.0;
            b2[q][jj] = on ? __builtin_fma(-r.y, Gkk, r.x) : 0.0;
          } else {
            b1[q][jj] = on ? gam * r.x : 0.0;
            b2[q][jj] = 0.0;
          }
        }
      }
    };
    auto update_row = [&](auto Itag, int r0, int lo, int hi) {
      constexpr int I = decltype(Itag)::value;
#pragma unroll
      for (int q = 0; q < kBand / 4; ++q) {
        const int sb = 4 * q + lr;
        const bool on = sb >= lo && sb < hi;
        const double2 r = SG(r0 + (on ? sb : 0))[16 * I + lc];  // A[i = lc][s = lr]
        const double a1 = on ? -r.x : 0.0, a2 = on ? -r.y : 0.0;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          acc[I][jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1[q][jj], acc[I][jj], 0, 0, 0);
          if (GC) acc[I][jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2[q][jj], acc[I][jj], 0, 0, 0);
        }
      }
    };
    auto hand_over = [&](auto Itag, int rnext) {  // the rows rnext .. rnext + 3 of tile row I (register v0) back to G_s
      constexpr int I = decltype(Itag)::value;
      const int v0 = (rnext & 15) >> 2;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
        for (int vv = 0; vv < kBand / 4; ++vv) {
          const int v = v0 + vv;
          const double val = (v == 0) ? acc[I][jj][0] : (v == 1) ? acc[I][jj][1] : (v == 2) ? acc[I][jj][2] : acc[I][jj][3];
          G_s[(size_t)(16 * I + 4 * v + lr) * kRowsWG + 16 * (2 * h + jj) + lc] = val;
        }
      }
    };
    // the whole of band bb for the tile rows BELOW the one that holds the band after it: nobody needs them before the
    // pivot reaches that tile row, so this runs where the G wave would otherwise wait for the pivot
    auto trailing = [&](int bb) {
      const int r0 = kBand * bb;
      const int In = (r0 + kBand) >> 4;
      if (In >= 3) return;
      load_b(r0, 0, kBand);
      if (In < 1) update_row(std::integral_constant<int, 1>(), r0, 0, kBand);
      if (In < 2) update_row(std::integral_constant<int, 2>(), r0, 0, kBand);
      update_row(std::integral_constant<int, 3>(), r0, 0, kBand);
    };
    if (kEarly == 0) {
      // one band behind: after band b, the rows of band b + 2 first (they go back to the pivot wave), then the tile rows below
      for (int b = 0; b + 2 < nbands; ++b) {
        const int r0 = kBand * b;
        const int I2 = (r0 + 2 * kBand) >> 4;  // tile row of band b + 2
        if (!wait_gt(&ctl[cSReady], r0 + kBand - 1, false)) break;
        if (DEFER && !rows_ready && (I2 > 0 || g_ctl(&ctl[cDef]) >= 12)) {
          // tile rows 1..3 of G have arrived (or are needed now): take them, and give them the bands they owe
          if (!wait_gt(&ctl[cDef], 11, false)) break;
          load_acc(1, 4);
          for (int bb = 0; bb < b; ++bb) {
            load_b(kBand * bb, 0, kBand);
            update_row(std::integral_constant<int, 1>(), kBand * bb, 0, kBand);
            update_row(std::integral_constant<int, 2>(), kBand * bb, 0, kBand);
            update_row(std::integral_constant<int, 3>(), kBand * bb, 0, kBand);
          }
          rows_ready = true;
          __builtin_amdgcn_s_setprio(0);  // (the priority was for the time the vector wave on this SIMD formed deferred tiles back to back)
        }
        load_b(r0, 0, kBand);
        switch (I2) {
          case 0: update_row(std::integral_constant<int, 0>(), r0, 0, kBand); hand_over(std::integral_constant<int, 0>(), r0 + 2 * kBand); break;
          case 1: update_row(std::integral_constant<int, 1>(), r0, 0, kBand); hand_over(std::integral_constant<int, 1>(), r0 + 2 * kBand); break;
          case 2: update_row(std::integral_constant<int, 2>(), r0, 0, kBand); hand_over(std::integral_constant<int, 2>(), r0 + 2 * kBand); break;
          default: update_row(std::integral_constant<int, 3>(), r0, 0, kBand); hand_over(std::integral_constant<int, 3>(), r0 + 2 * kBand); break;
        }
        if (lane == 0) g_ctl_set(&ctl[cBandH + h], b + 2);
        if (rows_ready) {  // (else: owed, see above)
          if (I2 < 1) update_row(std::integral_constant<int, 1>(), r0, 0, kBand);
          if (I2 < 2) update_row(std::integral_constant<int, 2>(), r0, 0, kBand);
          if (I2 < 3) update_row(std::integral_constant<int, 3>(), r0, 0, kBand);
        }
      }
      __syncthreads();  // B3
      return;
    }
    int pending = -1;  // a band whose trailing update is still owed
    for (int b = 0; b + 1 < nbands; ++b) {  // nothing follows the last band (a band before the last one is always full)
      const int r0 = kBand * b;
      const int Inext = (r0 + kBand) >> 4;  // tile row of the next band
      // The tile row of the next band must be current through the previous band before this band's early steps go on
      // top: it is, unless the next band opens a new tile row -- then the owed trailing update comes first.
      if (pending >= 0 && Inext != ((kBand * pending + kBand) >> 4)) {
        trailing(pending);
        pending = -1;
      }
      // phase 1, as soon as the band's first kEarly steps are published: those steps applied to the tile row of the
      // NEXT band, whose rows go back to the pivot wave at once (it applies the band's other steps itself)
      EFA_PS(const u64 ps_a = EFA_PS_NOW();)
      if (!wait_gt(&ctl[cHalf], 2 * b, false)) break;
      EFA_PS(const u64 ps_b = EFA_PS_NOW(); ps_w1 += ps_b - ps_a;
             if (a.dbg != nullptr && lane == 0 && h == 0 && own0 + 64 + b < P) a.dbg[(size_t)(own0 + 64 + b) * 8 + 3] = ps_b;)
      load_b(r0, 0, kEarly);
      EFA_PS(asm volatile("" : "+v"(b1[0][0]), "+v"(b1[0][1])); const u64 ps_b2 = EFA_PS_NOW(); ps_l1 += ps_b2 - ps_b;)
      switch (Inext) {
        case 0: update_row(std::integral_constant<int, 0>(), r0, 0, kEarly); hand_over(std::integral_constant<int, 0>(), r0 + kBand); break;
        case 1: update_row(std::integral_constant<int, 1>(), r0, 0, kEarly); hand_over(std::integral_constant<int, 1>(), r0 + kBand); break;
        case 2: update_row(std::integral_constant<int, 2>(), r0, 0, kEarly); hand_over(std::integral_constant<int, 2>(), r0 + kBand); break;
        default: update_row(std::integral_constant<int, 3>(), r0, 0, kEarly); hand_over(std::integral_constant<int, 3>(), r0 + kBand); break;
      }
      if (lane == 0) g_ctl_set(&ctl[cBandH + h], b + 1);
      EFA_PS(const u64 ps_c = EFA_PS_NOW(); ps_p1 += ps_c - ps_b;
             if (a.dbg != nullptr && lane == 0 && h == 0 && own0 + 64 + b < P) a.dbg[(size_t)(own0 + 64 + b) * 8 + 4] = ps_c;)
      // while the pivot runs the band's other steps: the trailing update owed from the previous band
      if (pending >= 0) {
        trailing(pending);
        pending = -1;
      }
      // phase 2, after the band: its other steps for the next band's tile row; the tile rows below it are owed
      if (!wait_gt(&ctl[cSReady], r0 + kBand - 1, false)) break;
      EFA_PS(const u64 ps_d = EFA_PS_NOW(); ps_w2 += ps_d - ps_c;)
      load_b(r0, kEarly, kBand);
      switch (Inext) {
        case 0: update_row(std::integral_constant<int, 0>(), r0, kEarly, kBand); break;
        case 1: update_row(std::integral_constant<int, 1>(), r0, kEarly, kBand); break;
        case 2: update_row(std::integral_constant<int, 2>(), r0, kEarly, kBand); break;
        default: update_row(std::integral_constant<int, 3>(), r0, kEarly, kBand); break;
      }
#if EFA_G_DEFER
      pending = (Inext < 3) ? b : -1;
#else
      trailing(b);
#endif
      EFA_PS(asm volatile("" : "+v"(acc[3][0][0]), "+v"(acc[3][1][0])); ps_p2 += EFA_PS_NOW() - ps_d;)
    }
    EFA_PS(if (a.dbg != nullptr && lane == 0) {
      a.dbg[(size_t)own0 * 8 + 2 + 4 * h * 0 + (h ? 8 : 0) + 0] = ps_w1;
      a.dbg[(size_t)own0 * 8 + 2 + (h ? 8 : 0) + 1] = ps_p1;
      a.dbg[(size_t)own0 * 8 + 2 + (h ? 8 : 0) + 2] = ps_w2;
      a.dbg[(size_t)own0 * 8 + 2 + (h ? 8 : 0) + 3] = ps_p2;
      a.dbg[(size_t)own0 * 8 + 2 + (h ? 8 : 0) + 4] = ps_l1;
    })
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
  int barriers_left = leads ? 3 : 0;
  bool bailed = false;
  long k = 0;
  while (k < P && !bailed) {
    if (leads && k == own0) {
      // ---------------- this workgroup's block ----------------
#ifdef EFA_PIPE_BLOCKTIME
      if (a.dbg != nullptr && lane == 0) a.dbg[(size_t)(own0 + wave) * 8 + 4] = __builtin_amdgcn_s_memtime();  // per vector wave
      EFA_WAIT_OUT(wave == 0 && lane == 0, 4, 7, __builtin_amdgcn_s_memrealtime());
#endif
#pragma unroll
      for (int c = 0; c < NC; ++c)
        *reinterpret_cast<double2*>(Yt + (size_t)i_loc * SP + 2 * PLg * c + 2 * j) = make_double2(x[2 * c], x[2 * c + 1]);
      const double rmean = group_rowsum<PLg, NC>(x) * invM;
      const double rnorm = DEFER ? group_dot<PLg, NC>(x, x) : 0.0;  // G_jj for the pivot's guard (its tile comes later)
      if (j == 0) {
        pm[i_loc] = rmean;
        pm[kRowsWG + i_loc] = xm;
        if (DEFER) pm[2 * kRowsWG + i_loc] = rnorm;
      }
      __syncthreads();  // B1: tile and parked means complete
      EFA_BLOCKSTAMP(wave == 0 && lane == 0, 5);
      if (!DEFER) form_gram();
      // For the block the rows change layout: wave w takes block rows 16 w .. 16 w + 15 as NJ accumulator tiles of
      // v_mfma_f64_16x16x4_f64: register v of tile J in lane l is member 16 J + (l & 15) of block row
      // 16 w + 4 v + (l >> 4).  A band's four rows are then register v = (band & 3) of ONE wave, already in the
      // B-operand layout (k = lane row, j = lane column): that wave -- the band's OWNER -- forms YE = L^-1 Y for all
      // column tiles straight from its registers, publishes the rows in the ring and applies the band to its own tile
      // from the same registers.  Nothing on this chain waits for another vector wave (no parking, no second counter);
      // the three other waves take the band from the ring when the owner's flag says it is there.
      static_assert(kBand == 4, "band ownership: four bands per 16-row tile");
      // Which 16 rows (bands 4 g .. 4 g + 3) a wave takes: wave 0 the first group, then wave 3, then waves 1 and 2.  The four
      // waves move through the bands in lock step (a band's owner publishes its YE, the others apply it), and with the deferred
      // Gram matrix waves 1 and 2 start the block ~10 k cycles late (four tiles each), wave 3 ~4.5 k (two), wave 0 at once:
      // while wave 1 owned bands 4..7 the whole group stood at band 4 until it arrived -- light stamps: every wave ~10 k cycles
      // behind the pivot from band 7 on, 13 k at the end of the block, the largest piece of the hand-over.  Now the late
      // waves own the late bands and catch up, as consumers, on what the early ones publish.
      const int grpw = DEFER ? ((wave == 0) ? 0 : (wave == 3) ? 1 : (wave == 1) ? 2 : 3) : wave;
      v4f64 xt[NJ];
      const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
      for (int J = 0; J < NJ; ++J)
#pragma unroll
        for (int v = 0; v < 4; ++v) xt[J][v] = Yt[(size_t)(16 * grpw + 4 * v + lr) * SP + 16 * J + lc];
      __syncthreads();  // B2: G complete; the tile region now belongs to the pivot wave's records
      barriers_left = 1;
      if (DEFER) gram_deferred();  // (the tile region stays the parked rows until cDef says all twelve tiles are done)
      for (int b4 = 0; b4 < nbands && !bailed; b4 += 4) {
        const bool owner = (b4 >> 2) == grpw;
#pragma unroll
        for (int vb = 0; vb < 4; ++vb) {  // vb = band & 3: the owner's register holding the band's rows (static index)
          const int b = b4 + vb;
          if (b >= nbands) break;
          const int r0 = kBand * b;
          const int s1 = (nb - r0 < kBand) ? nb - r0 : kBand;
          // ring slots of this band are those of the band kRingG obs earlier: every consumer must be through with them
          // (checked by the owner before it writes; the others only read)
          double ye0[NJ];  // the owner's YE tiles: register 0 of each MFMA result
          double av_own = 0.0;
          if (owner) {
            EFA_HO_IF(b == nbands - 1, 9, 0);  // the owner of the last band is ready for it (has finished the bands before)
            // ONE poll for both conditions -- the pivot has finished the band (cLinv) and the band's ring slots are free (every
            // consumer through with the band four bands earlier): the four vector waves move through the bands in lock step
            // at about the pivot's own pace, so every LDS round trip on this path is a round trip per band of lag.
            {
              const int need = (b >= kRingG / kBand) ? (int)(own0 + kBand * (b - kRingG / kBand) + kBand - 1) : -0x7fffffff;
              for (;;) {
                const int fl = g_ctl_lane(&ctl[cLinv]);
                const int mp = min_prog();
                if (__builtin_amdgcn_readfirstlane(fl) > b && mp >= need) break;
                if ((++polls & 15) == 0) {
                  if (g_ctl(&ctl[cBail]) != 0) {
                    bailed = true;
                    break;
                  }
                  budget -= 16;
                  if (budget <= 0 || EFA_TIMED_OUT()) {
                    give_up();
                    bailed = true;
                    break;
                  }
                }
              }
              if (bailed) break;
            }
            EFA_HO_IF(b == nbands - 1, 9, 2);  // ... the pivot's flag is up and there is ring space
            const double aop = (lc < kBand) ? LinvA[((size_t)b * kBand + lr) * kBand + lc] : 0.0;  // A[s = lc][t = lr], rows s >= 4 are zero
            av_own = SG((lr < s1) ? r0 + lr : r0)[16 * grpw + lc].y;  // (the update's kb operand, fetched in the same LDS round trip)
#pragma unroll
            for (int J = 0; J < NJ; ++J) {
              const v4f64 z = {0.0, 0.0, 0.0, 0.0};
              const v4f64 yt = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, xt[J][vb], z, 0, 0, 0);
              ye0[J] = yt[0];
            }
            // D[s = 4 v + lr][col = lc]: register 0 holds ye_{r0 + lr}, which is also B[k = lr][j = lc] of the update
            // (v_mfma_f64_4x4x4_4b would form exactly these four rows in a quarter of the matrix-core time, operands and result in
            //  these very lanes -- measured 4 % SLOWER overall, 4.53 vs 4.33 ms, although the rows leave 4 k cycles earlier)
            {  // the ye part of the band's records, straight to the trajectory (its scalars come from the forwarder wave)
              u64* grec = a.traj + (size_t)(own0 + r0 + ((lr < s1) ? lr : 0)) * TS;
              if (lr < s1) {
#pragma unroll
                for (int J = 0; J < NJ; ++J)
                  if (16 * J + lc < PAD) g_traj_store(grec + 16 * J + lc, ye0[J]);
              }
            }
            EFA_HO_IF(b == nbands - 1, 8, 6);  // T1b: the ye rows of the block's last band are on their way to global memory
#pragma unroll
            for (int J = 0; J < NJ; ++J)
              if (16 * J + lc < PAD) ring[(size_t)((own0 + r0 + lr) % kRingG) * TSR + 16 * J + lc] = ye0[J];
            if (lane == 0) g_ctl_set(&ctl[cYe], 4 * (b + 1));  // the other vector waves may read the band
          } else {
            if (!wait_gt(&ctl[cYe], 4 * (b + 1) - 1, true)) {
              bailed = true;
              break;
            }
          }
          // the band applied to this wave's 16 rows: X -= KB YE (rank s1 <= 4); A[i][k = lr] = kb of block row 16 w + i
          {
            const int st = r0 + lr;  // this lane's K slice: step st
            const bool valid = lr < s1;
            double av = owner ? av_own : SG(valid ? st : r0)[16 * grpw + lc].y;
            av = valid ? -av : 0.0;
            const double* bs = ring + (size_t)((own0 + (valid ? st : r0)) % kRingG) * TSR;
#pragma unroll
            for (int J = 0; J < NJ; ++J) {
              double bv = owner ? ye0[J] : bs[16 * J + lc];
              bv = valid ? bv : 0.0;
              xt[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, xt[J], 0, 0, 0);
            }
          }
          if (lane == 0) g_ctl_set(&ctl[cProg + wave], (int)(own0 + r0 + s1 - 1));  // ring slots up to here consumed
          EFA_HO_IF((b & 3) == 3, 10 + wave, b >> 2);  // this vector wave is through with bands 3, 7, 11, 15
        }
      }
      EFA_BLOCKSTAMP(wave == 0 && lane == 0, 6);
      __syncthreads();  // B3: every wave is done with the pivot's records; the tile region is free again
      barriers_left = 0;
      if (bailed) break;
      // back to the follower layout through the tile: rows come back from other waves, so the four vector waves
      // meet at an LDS counter between writing and reading (the other waves have left for their follower roles)
#pragma unroll
      for (int J = 0; J < NJ; ++J)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (16 * J + lc < PAD) Yt[(size_t)(16 * grpw + 4 * v + lr) * SP + 16 * J + lc] = xt[J][v];
      if (lane == 0) __hip_atomic_fetch_add(&ctl[cPark], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (!wait_gt(&ctl[cPark], kVW - 1, false)) {
        bailed = true;
        break;
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const double2 v = *reinterpret_cast<const double2*>(Yt + (size_t)i_loc * SP + 2 * PLg * c + 2 * j);
        x[2 * c] = v.x;
        x[2 * c + 1] = v.y;
      }
      xm = pm[kRowsWG + i_loc];  // the pivot wave carried the obs-space means through the block (:130)
      k = own1;
      continue;
    }
    // ---------------- follower steps: consume the records that are in the ring ----------------
    // One poll tells how many records are there (the band leader publishes four at a time); they are then applied
    // without further polls, the LDS reads of record k+1 issued before the arithmetic of record k (two register
    // sets, the loop is unrolled by two), and the progress word is written once per batch: a follower is bound by
    // what one wave can issue, so every instruction per record counts.
    if (!wait_gt(&ctl[cReady], (int)k, true)) {
      bailed = true;
      break;
    }
    long avail = g_ctl(&ctl[cReady]);  // (read again: reusing the value the wait saw measured slower -- more has usually arrived by now)
    const long lim = (leads && k < own0) ? own0 : P;
    if (avail > lim) avail = lim;
    if (avail > k + kRingG / 2) avail = k + kRingG / 2;  // progress is reported at least every half ring (slot recycling)
    double ya[2 * NC], yb2[2 * NC];
    double2 a01, b01;
    double wa = 1.0, wb = 1.0;  // GC: taper of the record's ob against this row (obs-obs table, fetched with the record)
    auto fetch = [&](long kk, double (&y)[2 * NC], double2& s01, double& w) {
      const double* slot = ring + (size_t)(kk % kRingG) * TSR;
      if (GC) w = slot[TS + i_loc];  // put there by the loader wave together with the record
      lds_read_row<PLg, NC>(slot, j, y);
      s01 = *reinterpret_cast<const double2*>(slot + PAD);      // c (0: not assimilated), m
    };
    auto apply = [&](const double (&y)[2 * NC], const double2 s01, const double w) {
      if (EFA_EXP(4096)) return;  // timing experiment: followers do no arithmetic
      // c = 0 marks an ob that is not assimilated (the same value in every lane); with Gaspari-Cohn a record whose taper is 0 on all
      // sixteen rows of this wave changes nothing either (ensrf.py:115: kcov x 0) -- most records, at a cut-off of a few thousand km
      if (__ballot(s01.x != 0.0 && (!GC || w != 0.0)) != 0ull) {
        double dot = group_dot<PLg, NC>(x, y);              // :95
        if (GC) dot = w * dot;                              // :115
        xm = __builtin_fma(s01.y, dot, xm);                 // :119, :130
        const double kb = s01.x * dot;                      // :119, :136
#pragma unroll
        for (int c = 0; c < 2 * NC; ++c) x[c] = __builtin_fma(-kb, y[c], x[c]);  // :141
      }
    };
    // (the prefetch is UNCONDITIONAL -- past the batch it re-reads the batch's last record: behind a conditional prefetch the
    //  compiler cannot count the LDS reads in flight and waits for all of them, the next record's included, before every apply)
    fetch(k, ya, a01, wa);
    while (k < avail) {
      fetch((k + 1 < avail) ? k + 1 : avail - 1, yb2, b01, wb);
      apply(ya, a01, wa);
      ++k;
      if (k >= avail) break;
      fetch((k + 1 < avail) ? k + 1 : avail - 1, ya, a01, wa);
      apply(yb2, b01, wb);
      ++k;
    }
    if (lane == 0) g_ctl_set(&ctl[cProg + wave], (int)(k - 1));
  }
  for (; barriers_left > 0; --barriers_left) __syncthreads();
  if (bailed || g_ctl(&ctl[cBail]) != 0) return;  // nothing written back: the host re-runs Phase A
  if (live) {
    if (vec) store_row<PLg, NC, true>(a.Yp + (size_t)row * M, M, j, x);
    else store_row<PLg, NC, false>(a.Yp + (size_t)row * M, M, j, x);
    if (j == 0) a.ym[row] = xm;
  }
}

template <int NC, bool GC>
hipError_t band_launch_gc(const PipeArgs& a, hipStream_t s) {
  const long grid = (a.R + kRowsWG - 1) / kRowsWG;
  const size_t lds = BandShape<NC, GC>::lds_bytes();
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pipe_band<NC, GC>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  // every workgroup waits for records of every other: the grid must fit the device at once
  int per_cu = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&k_pipe_band<NC, GC>), kGT, lds);
  if (e != hipSuccess) return e;
  if (grid > (long)per_cu * a.cu_count) return hipErrorCooperativeLaunchTooLarge;
  hipLaunchKernelGGL((k_pipe_band<NC, GC>), dim3((unsigned)grid), dim3(kGT), lds, s, a);
  return hipGetLastError();
}

template <int NC>
hipError_t band_launch(const PipeArgs& a, hipStream_t s) {
  return a.loc_mode != 0 ? band_launch_gc<NC, true>(a, s) : band_launch_gc<NC, false>(a, s);
}

}  // namespace

long band_traj_stride(int M) { return 2 * PLg * ((M + 2 * PLg - 1) / (2 * PLg)) + kTrajScalars; }

bool pipeline_band_supported(int M, long R, int loc_mode) {
  if (loc_mode != 0 && loc_mode != 1) return false;
  if (!(M >= 2 && M <= 128 && R > 0 && (R + kRowsWG - 1) / kRowsWG <= kPipeMaxWGs)) return false;
  const int nc = (M + 2 * PLg - 1) / (2 * PLg);
  const int pad = 2 * PLg * nc, ts = pad + kTrajScalars;
  const int sp = pad + ((2 - pad % 32) + 32) % 32;
  const size_t ureg = (size_t)kRowsWG * (sp > 2 * kRowsWG + kScStride + 1 ? sp : 2 * kRowsWG + kScStride + 1);
  const size_t dbl = (size_t)kRingG * (ts + (loc_mode != 0 ? kRowsWG : 0)) + kRowsWG * kRowsWG + ureg + 3 * kRowsWG + kNBands * kBand * kBand +
                     (loc_mode != 0 ? (size_t)kRowsWG * kRowsWG : ((EFA_DEFER_GRAM && nc <= 13) ? (size_t)16 * kRowsWG * 2 : 0));
  return dbl * 8 + 128 <= 160 * 1024;
}

hipError_t launch_pipeline_band(const PipeArgs& a, hipStream_t s) {
  if (!pipeline_band_supported(a.M, a.R, a.loc_mode) || a.P <= 0) return hipErrorInvalidValue;
  switch ((a.M + 2 * PLg - 1) / (2 * PLg)) {
    case 1: return band_launch<1>(a, s);
    case 2: return band_launch<2>(a, s);
    case 3: return band_launch<3>(a, s);
    case 4: return band_launch<4>(a, s);
    case 5: return band_launch<5>(a, s);
    case 6: return band_launch<6>(a, s);
    case 7: return band_launch<7>(a, s);
    case 8: return band_launch<8>(a, s);
    case 9: return band_launch<9>(a, s);
    case 10: return band_launch<10>(a, s);
    case 11: return band_launch<11>(a, s);
    case 12: return band_launch<12>(a, s);
    case 13: return band_launch<13>(a, s);
    case 14: return band_launch<14>(a, s);
    case 15: return band_launch<15>(a, s);
    case 16: return band_launch<16>(a, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace efa
