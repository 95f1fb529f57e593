// Localised (Gaspari-Cohn) state sweep, one pass, ROW PER LANE ("k_gc_rows").
//
// Same job and same inputs as k_sweep_gc (efa_gcsweep.hip): every state row is loaded once, the
// observations of its column block's active list are applied in order (ensrf.py:95-141 with the
// taper of :99-115), and the row is stored once.  What differs is the thread layout.
//
// rocprofv3 counters of k_sweep_gc on configs[2] (profiles/r02_cfg3_summary.txt) show a kernel bound
// by vector-instruction issue in which only 57 % of the vector instructions are the dot / update
// FMAs: with four lanes per row every (row, observation) pair pays a cross-lane reduction, the
// (s0+s1)+(s2+s3) tree and the gain scalars in all four lanes -- about 16 instructions per 40 FMAs --
// and the ye row comes through LDS (staging, barriers, 10 ds_read_b128 per wave and observation).
//
// Here ONE LANE owns a whole row (M doubles in registers: M <= 128, even):
//   - the dot product is in-lane: no reduction, the gain scalars once per row: ~9 instructions per
//     2 M FMAs;
//   - ye_k, the coefficients and the list entry are WAVE-UNIFORM, so they are scalar loads
//     (s_load_dwordx16 from the scalar cache / L2) and enter the FMAs as SGPR operands: no LDS, no
//     staging, no barrier -- the kernel has no __syncthreads and every wave is independent;
//   - a wave owns 16 columns x 4 variable-x-time slabs = 64 rows; lane addresses are 8 M bytes apart,
//     so a load instruction touches 64 cache lines, but consecutive instructions walk the same lines,
//     every byte fetched is used and there are ~200 observations' worth of arithmetic per row loaded:
//     the access pattern costs a few per cent of the kernel.
// Bound: fp64 vector FMA issue (4 M flop per (row, observation) pair against 16 M bytes per row).
#include "efa_device.h"
#include "efa_internal.h"

namespace efa {
namespace {

constexpr int kBlkCols = 16;
constexpr int kTileLeads = 4;  // slabs per wave tile: 16 columns x 4 slabs = 64 rows

template <int MV, bool FUSED>
__global__ __launch_bounds__(256) void k_gc_rows(const long ncol, const long n_lead, const long nblk, const long ngrp,
                                                 const long ntile, const long* __restrict__ off,
                                                 const int* __restrict__ idx, const double* __restrict__ wts,
                                                 const double* __restrict__ coef, const double* __restrict__ Ye,
                                                 const long ye_stride, const double* __restrict__ Xin,
                                                 const double* __restrict__ xin, double* __restrict__ Xout,
                                                 double* __restrict__ xout) {
  constexpr int M = 2 * MV;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so workgroups w, w+8, w+16 ...
  // (one XCD) take consecutive tiles: the ~10 workgroups that walk one column block's active list read
  // it, and its ye rows, through the same L2
  const long nwg = gridDim.x;
  const long per = (nwg + 7) / 8;
  const long vwg = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const long tile = vwg * 4 + wave;
  if (tile >= ntile) return;
  const long b = tile / ngrp;
  const long g = tile - b * ngrp;
  const int c = lane & 15, lo = lane >> 4;
  const long col = b * kBlkCols + c;
  const long lead = g * kTileLeads + lo;
  const bool live = col < ncol && lead < n_lead;
  const long row = lead * ncol + col;
  const double rM1 = 1.0 / (double)(M - 1);

  double x[M];
  double xm = 0.0;
  if (live) {
    const double2* p = reinterpret_cast<const double2*>(Xin + (size_t)row * M);
#pragma unroll
    for (int v = 0; v < MV; ++v) {
      const double2 t = p[v];
      x[2 * v] = t.x;
      x[2 * v + 1] = t.y;
    }
    if (!FUSED) xm = xin[row];
  } else {
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = 0.0;
  }
  if (FUSED) {  // prior members in: remove the ensemble mean (assimilation.py:146-147)
    double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int m = 0; m < M; ++m) s[m & 3] += x[m];
    xm = ((s[0] + s[1]) + (s[2] + s[3])) / (double)M;
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] -= xm;
  }

  const long e0 = off[b], e1 = off[b + 1];
  // the list entry and this lane's taper are fetched one entry ahead: their latency (a dependent scalar
  // load and a vector load) hides behind the previous observation's 2 M FMAs
  long k_nx = 0;
  double w_nx = 0.0;
  if (e0 < e1) {
    k_nx = idx[e0];
    w_nx = wts[(size_t)e0 * kBlkCols + c];
  }
  for (long e = e0; e < e1; ++e) {
    const long k = k_nx;
    const double w = live ? w_nx : 0.0;                           // taper of this lane's column (:99-111)
    if (e + 1 < e1) {
      k_nx = idx[e + 1];
      w_nx = wts[(size_t)(e + 1) * kBlkCols + c];
    }
    if (__ballot(w != 0.0) == 0ull) continue;                     // none of the wave's 64 rows
    const double* __restrict__ ye = Ye + (size_t)k * ye_stride;   // wave-uniform: scalar loads
    const double* __restrict__ ck = coef + (size_t)k * kCoefStride;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int m = 0; m < M; ++m) s[m & 3] = __builtin_fma(x[m], ye[m], s[m & 3]);
    const double dot = (s[0] + s[1]) + (s[2] + s[3]);
    double kc = dot * rM1;            // :95
    kc = w * kc;                      // :115
    const double km = kc * ck[1];     // :119
    xm = xm + km * ck[0];             // :130
    const double kb = ck[2] * km;     // :136
    // the update reads ye again through an index the optimiser cannot equate with k: kept live across the
    // dot product the row would need 2 M scalar registers (102 exist) and they would be spilled lane by lane.
    // (readfirstlane of a uniform value: no inline asm, which would turn every scalar load into a vector load)
    const long k2 = __builtin_amdgcn_readfirstlane((int)k);
    const double* __restrict__ ye2 = Ye + (size_t)k2 * ye_stride;
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = __builtin_fma(-kb, ye2[m], x[m]);  // :141
  }

  if (live) {
    double2* q = reinterpret_cast<double2*>(Xout + (size_t)row * M);
#pragma unroll
    for (int v = 0; v < MV; ++v) {
      if (FUSED) q[v] = make_double2(x[2 * v] + xm, x[2 * v + 1] + xm);  // posterior members out (assimilation.py:168)
      else q[v] = make_double2(x[2 * v], x[2 * v + 1]);
    }
    if (!FUSED) xout[row] = xm;
  }
}

template <int MV>
hipError_t rows_launch(const GcSweepArgs& a, hipStream_t s) {
  const long ngrp = (a.n_lead + kTileLeads - 1) / kTileLeads;
  const long ntile = a.nblk * ngrp;
  long nwg = (ntile + 3) / 4;
  nwg = (nwg + 7) / 8 * 8;  // a multiple of the XCD count: the XCD-aware order covers every tile
  const dim3 grid((unsigned)nwg), block(256);
  if (a.fused_members)
    hipLaunchKernelGGL((k_gc_rows<MV, true>), grid, block, 0, s, a.ncol, a.n_lead, a.nblk, ngrp, ntile, a.off, a.idx, a.wts,
                       a.coef, a.Ye, a.ye_stride, a.Xin, a.xin, a.Xout, a.xout);
  else
    hipLaunchKernelGGL((k_gc_rows<MV, false>), grid, block, 0, s, a.ncol, a.n_lead, a.nblk, ngrp, ntile, a.off, a.idx, a.wts,
                       a.coef, a.Ye, a.ye_stride, a.Xin, a.xin, a.Xout, a.xout);
  return hipGetLastError();
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

bool gc_rows_supported(const GcSweepArgs& a) {
  return a.M >= 2 && a.M <= 128 && (a.M % 2 == 0) && aligned16(a.Xin) && aligned16(a.Xout) && a.nblk > 0 &&
         a.nblk * ((a.n_lead + kTileLeads - 1) / kTileLeads) / 4 + 8 < (1L << 31);
}

hipError_t launch_gc_rows(const GcSweepArgs& a, hipStream_t s) {
  if (!gc_rows_supported(a)) return hipErrorInvalidValue;
  if (a.n_lead <= 0) return hipSuccess;
  switch (a.M / 2) {
#define EFA_ROWS_CASE(V) \
  case V:                \
    return rows_launch<V>(a, s);
    EFA_ROWS_CASE(1) EFA_ROWS_CASE(2) EFA_ROWS_CASE(3) EFA_ROWS_CASE(4) EFA_ROWS_CASE(5) EFA_ROWS_CASE(6) EFA_ROWS_CASE(7)
    EFA_ROWS_CASE(8) EFA_ROWS_CASE(9) EFA_ROWS_CASE(10) EFA_ROWS_CASE(11) EFA_ROWS_CASE(12) EFA_ROWS_CASE(13)
    EFA_ROWS_CASE(14) EFA_ROWS_CASE(15) EFA_ROWS_CASE(16) EFA_ROWS_CASE(17) EFA_ROWS_CASE(18) EFA_ROWS_CASE(19)
    EFA_ROWS_CASE(20) EFA_ROWS_CASE(21) EFA_ROWS_CASE(22) EFA_ROWS_CASE(23) EFA_ROWS_CASE(24) EFA_ROWS_CASE(25)
    EFA_ROWS_CASE(26) EFA_ROWS_CASE(27) EFA_ROWS_CASE(28) EFA_ROWS_CASE(29) EFA_ROWS_CASE(30) EFA_ROWS_CASE(31)
    EFA_ROWS_CASE(32) EFA_ROWS_CASE(33) EFA_ROWS_CASE(34) EFA_ROWS_CASE(35) EFA_ROWS_CASE(36) EFA_ROWS_CASE(37)
    EFA_ROWS_CASE(38) EFA_ROWS_CASE(39) EFA_ROWS_CASE(40) EFA_ROWS_CASE(41) EFA_ROWS_CASE(42) EFA_ROWS_CASE(43)
    EFA_ROWS_CASE(44) EFA_ROWS_CASE(45) EFA_ROWS_CASE(46) EFA_ROWS_CASE(47) EFA_ROWS_CASE(48) EFA_ROWS_CASE(49)
    EFA_ROWS_CASE(50) EFA_ROWS_CASE(51) EFA_ROWS_CASE(52) EFA_ROWS_CASE(53) EFA_ROWS_CASE(54) EFA_ROWS_CASE(55)
    EFA_ROWS_CASE(56) EFA_ROWS_CASE(57) EFA_ROWS_CASE(58) EFA_ROWS_CASE(59) EFA_ROWS_CASE(60) EFA_ROWS_CASE(61)
    EFA_ROWS_CASE(62) EFA_ROWS_CASE(63) EFA_ROWS_CASE(64)
#undef EFA_ROWS_CASE
    default: return hipErrorInvalidValue;
  }
}

}  // namespace efa
