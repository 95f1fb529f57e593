// Localised (Gaspari-Cohn) state sweep in ONE pass over the state: every row is loaded once,
// all observations whose taper is non-zero for it are applied in order, and it is stored once.
//
// Reference semantics (ensrf.py:99-115): kcov is multiplied by a taper that depends only on the
// (y, x) column (distance_to_point + gaspari_cohn, broadcast over variable x time), and is
// exactly 0 beyond 2 x halfwidth, where the update leaves the row bit-unchanged.  So
//   1. k_gc_build evaluates haversine + Gaspari-Cohn ONCE per (column, observation) -- not per
//      state row -- and keeps, per block of 16 columns, the ascending list of observations with
//      any non-zero weight together with the 16 weights (off / cnt / idx / wts).  One pass over the
//      trigonometry: a latitude-only upper bound per block (k_gc_bound) and a device prefix sum
//      (k_gc_scan) place the lists; k_gc_order sorts the blocks longest list first;
//   2. k_sweep_gc gives each workgroup one column block; its 4 waves (4 columns each) walk the n_lead
//      variable x time slabs of those columns (quad per row), and for each group of slabs loop over the
//      block's active list only, staged through LDS 32 observations at a time.
// Work drops from P passes' worth to (active fraction) x P; HBM traffic to one read + one write.
#include "efa_device.h"
#include "efa_internal.h"
#include "efa_rows.h"

#include <cstdlib>
#include <utility>

namespace efa {
namespace {

constexpr int kBlkCols = 16;  // columns per block == rows per wave in the quad layout

// One WAVE per column block: lane l = (column c = l & 15, observation slot o = l >> 4), four
// observations per step in ascending order.  Everything the list needs is wave-local (ballot +
// a running count in a scalar register), so the loop has no workgroup barrier; the exact
// latitude rejection keeps the trigonometry to the few observations near the block.
//
// The lists are built in ONE pass over the trigonometry: k_gc_bound counts, per block, the observations
// that survive the latitude test alone (an upper bound of the list length, no trigonometry), k_gc_scan
// turns the bounds into offsets on the device, and k_gc_build writes each block's entries at its offset
// and records the true length in cnt[b] (the space between a block's end and the next offset stays unused).
constexpr int kBuildWaves = 4;  // column blocks per workgroup

// the block's latitude range, the same in every lane
__device__ __forceinline__ void block_lat_range(bool col_ok, double la, double& la_lo, double& la_hi) {
  la_lo = col_ok ? la : 1e300;
  la_hi = col_ok ? la : -1e300;
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    la_lo = fmin(la_lo, __shfl_xor(la_lo, m, 64));
    la_hi = fmax(la_hi, __shfl_xor(la_hi, m, 64));
  }
}

// 64 observations at once, lane <-> observation: the great-circle distance to ANY column of the
// block is at least R * (latitude gap to the block's range); beyond 2 x halfwidth the
// Gaspari-Cohn weight is exactly 0, so most observations never reach the trigonometry
__device__ __forceinline__ unsigned long long lat_candidates(long k, long P, double la_lo, double la_hi,
                                                             const double* __restrict__ ob_lat,
                                                             const double* __restrict__ ob_hw,
                                                             const double* __restrict__ coef) {
  bool cnd = false;
  if (k < P && coef[k * kCoefStride + 3] != 0.0) {
    const double hw = ob_hw[k], olat = ob_lat[k];
    const double gap = fmax(0.0, fmax(olat - la_hi, la_lo - olat));
    cnd = (kEarthRadiusKm * radians(gap) <= 2.0 * fabs(hw) * (1.0 + 1e-9)) || !(hw == hw);
  }
  return __ballot(cnd);
}

__global__ __launch_bounds__(64 * kBuildWaves) void k_gc_bound(long ncol, long nblk, long P,
                                                               const double* __restrict__ glat,
                                                               const double* __restrict__ ob_lat,
                                                               const double* __restrict__ ob_hw,
                                                               const double* __restrict__ coef, int* __restrict__ ub) {
  const int lane = threadIdx.x & 63;
  const long b = (long)blockIdx.x * kBuildWaves + (threadIdx.x >> 6);
  if (b >= nblk) return;
  const long col = b * kBlkCols + (lane & 15);
  const bool col_ok = col < ncol;
  double la_lo, la_hi;
  block_lat_range(col_ok, col_ok ? glat[col] : 0.0, la_lo, la_hi);
  int n = 0;
  for (long k0 = 0; k0 < P; k0 += 64) n += __builtin_popcountll(lat_candidates(k0 + lane, P, la_lo, la_hi, ob_lat, ob_hw, coef));
  if (lane == 0) ub[b] = n;
}

// off[b] = sum of ub[0..b), off[nblk] = total; one workgroup (nblk is ncol / 16: tens of thousands)
constexpr int kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void k_gc_scan(long nblk, const int* __restrict__ ub, long* __restrict__ off) {
  __shared__ long part[kScanThreads];
  const int t = threadIdx.x;
  const long per = (nblk + kScanThreads - 1) / kScanThreads;
  const long lo = (long)t * per, hi = (lo + per < nblk) ? lo + per : nblk;
  long sum = 0;
  for (long b = lo; b < hi; ++b) sum += ub[b];
  part[t] = sum;
  __syncthreads();
  for (int d = 1; d < kScanThreads; d <<= 1) {  // inclusive Hillis-Steele over the per-thread sums
    const long v = (t >= d) ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  long run = part[t] - sum;
  for (long b = lo; b < hi; ++b) {
    off[b] = run;
    run += ub[b];
  }
  if (t == kScanThreads - 1) off[nblk] = part[t];
}

// Per observation: cos/sin of latitude and longitude and s_lim = sin^2(|halfwidth| / R), the haversine argument at
// which the taper reaches exactly 0 (distance = 2 halfwidths).  With them the haversine argument of a (column, ob)
// pair costs ten multiply-adds and no trigonometry: k_gc_build uses that cheap value ONLY to reject pairs that are
// clearly beyond the cut-off (relative margin 1e-6); every pair it keeps is evaluated with the reference's formula.
constexpr int kObTrig = 6;  // doubles per ob: cos lat, sin lat, cos lon, sin lon, s_lim, (pad)
__global__ void k_gc_obtrig(long P, const double* __restrict__ ob_lat, const double* __restrict__ ob_lon,
                            const double* __restrict__ ob_hw, double* __restrict__ tab) {
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= P) return;
  const double plat = radians(ob_lat[k]), plon = radians(ob_lon[k]);
  const double ang = fabs(ob_hw[k]) / kEarthRadiusKm;  // half the cut-off angle
  double slim = sin(ang);
  slim = slim * slim;
  if (!(ang < 1.5)) slim = 4.0;  // cut-off beyond a quarter of the globe (or a NaN radius): nothing is rejected cheaply
  double* t = tab + k * kObTrig;
  t[0] = cos(plat);
  t[1] = sin(plat);
  t[2] = cos(plon);
  t[3] = sin(plon);
  t[4] = slim;
  t[5] = 0.0;
}

// order[i] = the block with the i-th longest list (counting sort on cnt >> shift, one workgroup): the sweep
// hands out blocks longest first, so the last workgroups to start are the cheapest ones.  On a regular
// lat/lon grid the lists near the poles are several times longer than near the equator, and in blockIdx
// order the polar rows are both the first and the LAST blocks: the tail of the launch then runs on a few CUs.
__global__ __launch_bounds__(kScanThreads) void k_gc_order(long nblk, const int* __restrict__ cnt, int shift,
                                                          int* __restrict__ order) {
  __shared__ int hist[kScanThreads];
  const int t = threadIdx.x;
  hist[t] = 0;
  __syncthreads();
  for (long b = t; b < nblk; b += kScanThreads) {
    int key = cnt[b] >> shift;
    key = kScanThreads - 1 - (key < kScanThreads ? key : kScanThreads - 1);
    atomicAdd(&hist[key], 1);
  }
  __syncthreads();
  const int mine = hist[t];
  for (int d = 1; d < kScanThreads; d <<= 1) {
    const int v = (t >= d) ? hist[t - d] : 0;
    __syncthreads();
    hist[t] += v;
    __syncthreads();
  }
  const int excl = hist[t] - mine;
  __syncthreads();
  hist[t] = excl;
  __syncthreads();
  for (long b = t; b < nblk; b += kScanThreads) {
    int key = cnt[b] >> shift;
    key = kScanThreads - 1 - (key < kScanThreads ? key : kScanThreads - 1);
    order[atomicAdd(&hist[key], 1)] = (int)b;
  }
}

// COUNT_ONLY: the same pass without the lists -- cnt[b] and the pair total only (efa_gc_block_counts: the cost of a
// column block for the cost-balanced column split of distributed.py)
template <bool COUNT_ONLY>
__global__ __launch_bounds__(64 * kBuildWaves) void k_gc_build(long ncol, long nblk, long P,
                                                               const double* __restrict__ glat,
                                                               const double* __restrict__ glon,
                                                               const double* __restrict__ ob_lat,
                                                               const double* __restrict__ ob_lon,
                                                               const double* __restrict__ ob_hw,
                                                               const double* __restrict__ coef,
                                                               const double* __restrict__ obtrig, int* __restrict__ cnt,
                                                               const long* __restrict__ off, int* __restrict__ idx,
                                                               double* __restrict__ wts,
                                                               unsigned long long* __restrict__ npairs) {
  // Round 3: two steps per 64 observations.  (1) lane <-> observation: the latitude candidates run the trig-free rejection
  // against each of the block's columns (their cos / sin from LDS) and the survivors -- observations within reach of at least one
  // column -- go into a small per-wave queue in ascending order; (2) lane <-> (column, one of four queued observations): the
  // reference's formula, now with nearly every lane busy.  Before, (2) ran on the latitude candidates directly and a step of
  // four candidates paid for the trigonometry whenever any of its 64 pairs survived (about half of the steps, a few lanes each).
  __shared__ double coltrig[kBuildWaves][kBlkCols][4];
  __shared__ int queue[kBuildWaves][128];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long b = (long)blockIdx.x * kBuildWaves + wv;
  if (b >= nblk) return;  // (no workgroup barrier below)
  const int c = lane & 15, o = lane >> 4;
  const long col = b * kBlkCols + c;
  const bool col_ok = col < ncol;
  const int ncols_live = (int)((ncol - b * kBlkCols < kBlkCols) ? ncol - b * kBlkCols : kBlkCols);
  const double la = col_ok ? glat[col] : 0.0, lo = col_ok ? glon[col] : 0.0;
  const double cg = cos(radians(la)), sg = sin(radians(la)), cl = cos(radians(lo)), sl = sin(radians(lo));
  if (lane < kBlkCols) {
    coltrig[wv][lane][0] = cg;
    coltrig[wv][lane][1] = sg;
    coltrig[wv][lane][2] = cl;
    coltrig[wv][lane][3] = sl;
  }
  __builtin_amdgcn_wave_barrier();  // (one wave's LDS operations complete in order)
  double la_lo, la_hi;
  block_lat_range(col_ok, la, la_lo, la_hi);
  const long first = COUNT_ONLY ? 0 : off[b];
  long running = first;
  long pairs = 0;  // (column, observation) pairs with a non-zero taper: SURVEY.md 8d's bytes_touched
  int qh = 0, qn = 0;  // the queue's head and length (wave-uniform)
  auto drain = [&](bool all) {
    while (qn >= 4 || (all && qn > 0)) {  // four queued observations per step, ascending
      const int take = qn < 4 ? qn : 4;
      const long k = (o < take) ? queue[wv][(qh + o) & 127] : -1;
      qh += take;
      qn -= take;
      double w = 0.0;
      if (k >= 0 && col_ok) {
        const double* t = obtrig + k * kObTrig;
        const double2 tp = *reinterpret_cast<const double2*>(t), tl = *reinterpret_cast<const double2*>(t + 2);
        const double cc = tp.x * cg;
        const double h = 0.5 * (1.0 - (cc + tp.y * sg)) + cc * (0.5 * (1.0 - (tl.x * cl + tl.y * sl)));  // sin^2(d / 2R), cheaply
        if (!(h > t[4] * (1.0 + 1e-6) + 1e-13)) {  // not clearly beyond 2 halfwidths: the reference's own arithmetic decides
          const double hw = ob_hw[k], olat = ob_lat[k];
          w = gaspari_cohn(distance_to_point_km(la, lo, olat, ob_lon[k]), hw);
        }
      }
      const unsigned long long bal = __ballot(w != 0.0);
      if (bal == 0ull) continue;  // wave-uniform
      pairs += __builtin_popcountll(bal);
      int before = 0, total = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = ((bal >> (16 * i)) & 0xFFFFull) != 0ull ? 1 : 0;
        before += (i < o) ? f : 0;
        total += f;
      }
      if (!COUNT_ONLY && ((bal >> (16 * o)) & 0xFFFFull) != 0ull) {
        const long e = running + before;
        if (c == 0) idx[e] = (int)k;
        wts[e * kBlkCols + c] = w;
      }
      running += total;
    }
  };
  for (long k0 = 0; k0 < P; k0 += 64) {
    const unsigned long long cand = lat_candidates(k0 + lane, P, la_lo, la_hi, ob_lat, ob_hw, coef);
    if (cand == 0ull) continue;
    bool surv = false;
    if ((cand >> lane) & 1ull) {
      const double* t = obtrig + (k0 + lane) * kObTrig;
      const double2 tp = *reinterpret_cast<const double2*>(t), tl = *reinterpret_cast<const double2*>(t + 2);
      const double lim = t[4] * (1.0 + 1e-6) + 1e-13;
      for (int c2 = 0; c2 < ncols_live; ++c2) {
        const double cc = tp.x * coltrig[wv][c2][0];
        const double h = 0.5 * (1.0 - (cc + tp.y * coltrig[wv][c2][1])) + cc * (0.5 * (1.0 - (tl.x * coltrig[wv][c2][2] + tl.y * coltrig[wv][c2][3])));
        surv = surv || !(h > lim);
      }
    }
    const unsigned long long sm = __ballot(surv);
    if (sm == 0ull) continue;
    if (surv) queue[wv][(qh + qn + __builtin_popcountll(sm & ((1ull << lane) - 1ull))) & 127] = (int)(k0 + lane);
    qn += __builtin_popcountll(sm);
    __builtin_amdgcn_wave_barrier();
    drain(false);
  }
  drain(true);
  if (lane == 0) {
    cnt[b] = (int)(running - first);
    if (COUNT_ONLY && idx) idx[b] = (int)pairs;  // counting pass: the block's (column, observation) pairs
    if (npairs) atomicAdd(npairs, (unsigned long long)pairs);
  }
}

constexpr int kChunk = 32;  // active observations staged in LDS at a time

// Workgroup = one column block.  Wave w owns columns 4w .. 4w+3 of the block; its 16 quads are 4 columns x 4
// variable x time slabs, RPL rows (slabs) per quad, so the four waves walk the same 4 RPL slabs of different
// columns in lock step.  The block's active observations are staged chunk by chunk into LDS ONCE per group
// of slabs (ye rows, tapers, coefficients), so the per-lane traffic of the inner loop is LDS only; a wave
// skips an observation whose taper is zero on all of ITS four columns (the list is per 16 columns, and a
// block spans up to 8 degrees of longitude: a sixth of the listed (column, ob) pairs have zero weight).
// Fetching ye per lane straight from L2 made the kernel vector-memory-issue bound (10 x 1 KB requests per
// wave and observation through one 64 B/clk path per CU).
// The kernel is bound by fp64 VALU issue (profiles/r02_cfg3_summary.txt: 57 % of its VALU instructions are the
// 4 M FMAs per (row, ob) the arithmetic needs, VALU busy 3/4 of the time); three waves per SIMD (<= 168 VGPRs)
// while two rows of up to 80 members fit that budget without spilling, two waves above (M <= 128 at one row per quad).
#ifndef EFA_GC_MINWAVES
#define EFA_GC_MINWAVES 3
#endif
#ifndef EFA_GC_RPL
#define EFA_GC_RPL 2
#endif
// Rows per quad (they share every ye row read from LDS and the staging of a chunk): 2 at three waves per SIMD up to 8 chunks;
// 4 at two waves for 9..10 chunks (65..80 members: configs[2], 64.0 -> 62.2 ms) and 3 at two waves for 11..13 chunks
// (81..104 members: configs[3], 151 -> 141 ms); the few registers that spill are row addresses and means, saved and
// reloaded once per group of slabs -- the loop over the observations stays spill-free (checked in the assembly).
#ifndef EFA_GC_RPL_MID
#define EFA_GC_RPL_MID 4
#endif
#ifndef EFA_GC_RPL_WIDE
#define EFA_GC_RPL_WIDE 3
#endif
#ifndef EFA_GC_RPL_XWIDE
#define EFA_GC_RPL_XWIDE 1  // at 14..16 chunks (105..128 members)
#endif
#ifndef EFA_GC_LANE
#define EFA_GC_LANE 1  // cycles of up to 104 members (an even number, aligned rows) run the row-per-lane kernel (k_sweep_gc_lane)
#endif
#ifndef EFA_GC_COLSPLIT
#define EFA_GC_COLSPLIT 1
#endif
// dot(x, ye) over the lane's slots with two FMA chains (the other row of the quad and the other waves
// of the SIMD fill the issue slots), then the quad total
template <int NC>
__device__ __forceinline__ double gc_dot(const double (&x)[2 * NC], const double (&y)[2 * NC]) {
  double s0 = x[0] * y[0], s1 = x[1] * y[1];
#pragma unroll
  for (int c = 1; c < NC; ++c) {
    s0 = __builtin_fma(x[2 * c], y[2 * c], s0);
    s1 = __builtin_fma(x[2 * c + 1], y[2 * c + 1], s1);
  }
  return group_sum<4>(s0 + s1);
}

// waves per SIMD the register budget allows: RPL rows and one ye row of 2 NC doubles per lane, ~36 registers of everything else
constexpr int gc_min_waves(int NC, int RPL) {
  return (4 * NC * (RPL + 1) + 36 <= 168) ? EFA_GC_MINWAVES : (4 * NC * (RPL + 1) + 36 <= 256 || NC * (RPL + 1) <= 52) ? 2 : 1;
}

template <int NC, bool VEC, bool FUSED, int RPL>
__global__ __launch_bounds__(256, gc_min_waves(NC, RPL)) void k_sweep_gc(const GcSweepArgs a) {
  constexpr int L = 4;
  constexpr int S = 2 * L * NC;  // padded ye row (doubles)
  __shared__ __align__(16) double ye_s[kChunk * S];
  __shared__ __align__(16) double2 ab_s[kChunk * kBlkCols];  // per (staged ob, column): what scales (x . ye) in the row / in its mean
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int j = lane & 3, r = lane >> 2;
  // blockIdx.x = (position in the longest-first order) * lead_split + (group of slabs): a shard with few, long
  // column blocks (the polar ranks of a cost-balanced split) still fills the device and has no tail of whole blocks
  const long b = a.order[blockIdx.x / a.lead_split];
  const int lead_lo = (int)(blockIdx.x % a.lead_split) * (int)a.lead_chunk;
  const int lead_hi = (lead_lo + (int)a.lead_chunk < (int)a.n_lead) ? lead_lo + (int)a.lead_chunk : (int)a.n_lead;
  // quad r of wave w: column cq of the block, slab slot sq of the group of slabs
  const int cq = EFA_GC_COLSPLIT ? 4 * wave + (r & 3) : r;
  const int sq = EFA_GC_COLSPLIT ? (r >> 2) : wave;
  const long col = b * kBlkCols + cq;
  const bool col_ok = col < a.ncol;
  const int M = a.M;
  const double rM1 = 1.0 / (double)(M - 1);
  const long e0 = a.off[b], e1 = e0 + a.cnt[b];

  // A quad holds RPL rows of the SAME column (slabs lead, lead + 4, ...): they share the taper and
  // every ye row read from LDS (the quad layout delivers each ye row once per quad), and a staged
  // chunk serves 4 RPL slabs instead of 4.
  for (int lead0 = lead_lo; lead0 < lead_hi; lead0 += 4 * RPL) {
    double x[RPL][2 * NC];
    double xm[RPL];
    bool live[RPL];
    long row[RPL];
    bool any_live = false;
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      const int lead = lead0 + sq + 4 * q;
      live[q] = col_ok && lead < lead_hi;
      any_live = any_live || live[q];
      row[q] = (long)lead * a.ncol + col;
      xm[q] = 0.0;
      if (live[q]) {
        load_row<L, NC, VEC>(a.Xin + (size_t)row[q] * M, M, j, x[q]);
        if (!FUSED) xm[q] = a.xin[row[q]];
      } else {
#pragma unroll
        for (int c = 0; c < 2 * NC; ++c) x[q][c] = 0.0;
      }
      if (FUSED) {  // prior members in: remove the ensemble mean (assimilation.py:146-147)
        xm[q] = group_rowsum<L, NC>(x[q]) / (double)M;
#pragma unroll
        for (int c = 0; c < 2 * NC; ++c) x[q][c] -= xm[q];  // padding slots never reach the output or the dot
      }
    }
    for (long c0 = e0; c0 < e1; c0 += kChunk) {
      const int ne = (int)((e1 - c0 < kChunk) ? (e1 - c0) : kChunk);
      __syncthreads();  // previous chunk fully consumed
      // ---- cooperative staging of ne entries
      if (VEC) {
        constexpr int S2 = S / 2;
        const int M2 = M / 2;
        for (int i = tid; i < ne * S2; i += 256) {
          const int ee = i / S2, m2 = i - ee * S2;
          const int k = a.idx[c0 + ee];
          reinterpret_cast<double2*>(ye_s)[i] =
              (m2 < M2) ? reinterpret_cast<const double2*>(a.Ye + (size_t)k * a.ye_stride)[m2] : make_double2(0.0, 0.0);
        }
      } else {
        for (int i = tid; i < ne * S; i += 256) {
          const int ee = i / S, m = i - ee * S;
          const int k = a.idx[c0 + ee];
          ye_s[i] = (m < M) ? a.Ye[(size_t)k * a.ye_stride + m] : 0.0;
        }
      }
      // The gain scalars of ensrf.py:95-136 -- kcov = (x . ye)/(M-1), times the taper, /kdenom, times innov for the mean
      // and times beta for the members -- do not depend on the state row: they are folded ONCE per (ob, column) here,
      // A = w (1/(M-1)) (1/kdenom) beta and B = w (1/(M-1)) (1/kdenom) innov, instead of six dependent multiplications
      // per row and observation in a loop that is bound by the number of fp64 instructions it issues.
      for (int i = tid; i < ne * kBlkCols; i += 256) {
        const double w = a.wts[(size_t)c0 * kBlkCols + i];
        const double* ck = a.coef + (size_t)a.idx[c0 + i / kBlkCols] * kCoefStride;  // innov, 1/kdenom, beta, active
        const double g = (w * rM1) * ck[1];
        ab_s[i] = make_double2(g * ck[2], g * ck[0]);  // w == 0 (or an ob that is not assimilated): both exactly 0
      }
      __syncthreads();
      // ---- apply the chunk to this wave's 16 RPL rows
      for (int ee = 0; ee < ne; ++ee) {
        const double2 ab = ab_s[ee * kBlkCols + cq];
        if (!any_live || __ballot(ab.x != 0.0) == 0ull) continue;  // none of this wave's rows (dead slabs / zero taper)
        double y[2 * NC];
        lds_read_row<L, NC>(ye_s + ee * S, j, y);
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
          if (RPL > 2 && lead0 + 4 * q >= lead_hi) continue;  // wave-uniform: this slot is beyond the last slab in every quad
          const double dot = gc_dot<NC>(x[q], y);        // :95 (a dead row holds zeros: its dot, and so its update, is exactly 0)
          xm[q] = __builtin_fma(ab.y, dot, xm[q]);       // :115, :119, :130
          const double kb = ab.x * dot;                  // :115, :119, :136
#pragma unroll
          for (int c = 0; c < 2 * NC; ++c) x[q][c] = __builtin_fma(-kb, y[c], x[q][c]);  // :141
        }
      }
    }
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      if (live[q]) {
        if (FUSED) {  // posterior members out (assimilation.py:168)
#pragma unroll
          for (int c = 0; c < 2 * NC; ++c) x[q][c] += xm[q];
        }
        store_row<L, NC, VEC>(a.Xout + (size_t)row[q] * M, M, j, x[q]);
        if (!FUSED && j == 0) a.xout[row[q]] = xm[q];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Row-per-lane form of the same sweep (round 3).  A lane holds one WHOLE state row (up to 104 members in registers), so the
// dot x . ye is lane-local -- no quad butterflies -- and every fp64 instruction of the loop over the observations is a
// multiply-add: v_fmac_f64 with a DPP row_newbcast source, which takes one multiplicand from lane l of the lane's own
// 16-lane row.  Lane l of each row holds ye members l, 16 + l, 32 + l, ...: the whole ye vector sits in 5..7 registers
// per lane and is read from LDS with that many 8-byte reads per wave and observation (the quad form: 10..13 16-byte reads
// delivering the same row to each of 16 quads).  tools/dpp_fmac_probe.hip: the DPP form issues at the rate of the plain FMA.
// A wave is 4 columns x 16 slabs (the same four columns per wave as the quad form, so the same zero-taper skipping); the last,
// partial group of slabs of a column block is folded onto fewer waves (8 slabs: 8 columns per wave, two waves; 4 slabs: all
// 16 columns on one wave) instead of running with dead lanes.
// The DPP instructions are inline assembly (the compiler has no 64-bit DPP intrinsic).  Their DPP operand, ye, is written by
// LDS reads only, never by a VALU instruction, so the "VALU write -> DPP read" hazard (which the compiler does not track
// through inline assembly) cannot arise; tests/test_cpu_host.py checks the generated code for exactly that.
template <int L>
__device__ __forceinline__ void fmac_bcast(double& acc, double y, double x) {  // acc += (y of lane L of this 16-lane row) * x
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(y), "v"(x), "n"(L));
}
template <int MP, int... I>
__device__ __forceinline__ double lane_dot(const double (&x)[MP], const double (&y)[(MP + 15) / 16], std::integer_sequence<int, I...>) {
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  (fmac_bcast<I % 16>(acc[I & 3], y[I / 16], x[I]), ...);
  return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
// x[16 C + i] += (ye member 16 C + i) * nkb for the members of ye register C
template <int MP, int C, int... I>
__device__ __forceinline__ void lane_update_group(double (&x)[MP], double yc, double nkb, std::integer_sequence<int, I...>) {
  (fmac_bcast<I>(x[16 * C + I], yc, nkb), ...);
}
// The update walks the ye registers in order and re-loads each one with the NEXT active observation's members as soon as its
// sixteen multiply-adds are issued: the LDS latency of the next ye row hides behind the rest of this update and the head of
// the next dot, without a second set of ye registers (there are none to spare at 100 members).
template <int MP, int C = 0>
__device__ __forceinline__ void lane_update_prefetch(double (&x)[MP], double (&y)[(MP + 15) / 16], double nkb, const double* __restrict__ ynext) {
  constexpr int NG = (MP + 15) / 16;
  if constexpr (C < NG) {
    constexpr int n = (MP - 16 * C < 16) ? MP - 16 * C : 16;
    lane_update_group<MP, C>(x, y[C], nkb, std::make_integer_sequence<int, n>{});
    y[C] = ynext[16 * C];
    lane_update_prefetch<MP, C + 1>(x, y, nkb, ynext);
  }
}

constexpr int kLaneMaxMembers = 104;
#ifndef EFA_GC_LANE_CHUNK
#define EFA_GC_LANE_CHUNK 32
#endif
constexpr int kChunkL = EFA_GC_LANE_CHUNK;  // observations staged at a time by the row-per-lane kernel (a 64-bit mask of them per wave)
template <int MP, bool FUSED>  // members padded to a multiple of 4; FUSED: prior members in, posterior members out
__global__ __launch_bounds__(256, 2) void k_sweep_gc_lane(const GcSweepArgs a) {
  constexpr int NG = (MP + 15) / 16;  // ye registers per lane
  constexpr int YS = 16 * NG;         // padded ye row in LDS (doubles)
  __shared__ __align__(16) double ye_s[kChunkL * YS];
  __shared__ __align__(16) double2 ab_s[kChunkL * kBlkCols];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  // One workgroup per (column block, group of 16 slabs), blocks longest list first, a block's groups next to each other (its
  // list stays in L2) but starting at a different group from block to block.  (The list is staged once per group of slabs
  // whichever workgroup takes it, so the fine split costs nothing.  The hardware deals consecutive workgroups to the eight
  // XCDs in turn: with several groups per workgroup and unequal parts, the larger parts of every block landed on the same
  // XCDs -- measured as up to 25 % imbalance on a polar shard; all blocks' first groups, then all second groups, ... is even
  // but 2.5 % slower on the whole grid, the lists leaving L2 between a block's groups.)
  const long pos = blockIdx.x / a.lead_split;
  const long b = a.order[pos];
  const int lead_lo = 16 * (int)((blockIdx.x % a.lead_split + pos) % a.lead_split);
  const int lead_hi = (lead_lo + 16 < (int)a.n_lead) ? lead_lo + 16 : (int)a.n_lead;
  const int M = a.M, M2 = M / 2;
  const double rM1 = 1.0 / (double)(M - 1);
  const long e0 = a.off[b], e1 = e0 + a.cnt[b];
  const auto seq = std::make_integer_sequence<int, MP>{};

  for (int lead0 = lead_lo; lead0 < lead_hi; lead0 += 16) {
    // 16 slabs x 4 columns per wave; the last group of a block: the next power of two of what is left, more columns per wave
    const int rem = lead_hi - lead0;
    const int lg_cols = (rem > 8) ? 2 : (rem > 4) ? 3 : (rem > 2) ? 4 : (rem > 1) ? 5 : 6;
    const int ncw = 1 << lg_cols;                 // columns per wave
    const int cq = (lane & (ncw - 1)) + ncw * wave;
    const int lead = lead0 + (lane >> lg_cols);
    const long col = b * kBlkCols + cq;
    const bool live = cq < kBlkCols && col < a.ncol && lead < lead_hi;
    const bool any_live = __ballot(live) != 0ull;
    const long row = (long)lead * a.ncol + col;
    double x[MP];
    double xm = 0.0;
    if (live) {
      const double2* p = reinterpret_cast<const double2*>(a.Xin + (size_t)row * M);
      if (M == MP) {
#pragma unroll
        for (int i = 0; i < MP / 2; ++i) {
          const double2 v = p[i];
          x[2 * i] = v.x;
          x[2 * i + 1] = v.y;
        }
      } else {
#pragma unroll
        for (int i = 0; i < MP / 2; ++i) {
          const double2 v = (i < M2) ? p[i] : make_double2(0.0, 0.0);
          x[2 * i] = v.x;
          x[2 * i + 1] = v.y;
        }
      }
      if (FUSED) {  // prior members in: remove the ensemble mean (assimilation.py:146-147)
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int i = 0; i < MP; i += 4) {
          s0 += x[i];
          s1 += x[i + 1];
          s2 += x[i + 2];
          s3 += x[i + 3];
        }
        xm = ((s0 + s1) + (s2 + s3)) / (double)M;
#pragma unroll
        for (int i = 0; i < MP; ++i) x[i] = (i < M) ? x[i] - xm : 0.0;
      } else {
        xm = a.xin[row];
      }
    } else {
#pragma unroll
      for (int i = 0; i < MP; ++i) x[i] = 0.0;
    }
    for (long c0 = e0; c0 < e1; c0 += kChunkL) {
      const int ne = (int)((e1 - c0 < kChunkL) ? (e1 - c0) : kChunkL);
      __syncthreads();  // previous chunk fully consumed
      {
        constexpr int S2 = YS / 2;
        for (int i = tid; i < ne * S2; i += 256) {
          const int ee = i / S2, m2 = i - ee * S2;
          const int k = a.idx[c0 + ee];
          reinterpret_cast<double2*>(ye_s)[i] =
              (m2 < M2) ? reinterpret_cast<const double2*>(a.Ye + (size_t)k * a.ye_stride)[m2] : make_double2(0.0, 0.0);
        }
      }
      for (int i = tid; i < ne * kBlkCols; i += 256) {  // the folded gain scalars, as in k_sweep_gc
        const double w = a.wts[(size_t)c0 * kBlkCols + i];
        const double* ck = a.coef + (size_t)a.idx[c0 + i / kBlkCols] * kCoefStride;
        const double g = (w * rM1) * ck[1];
        ab_s[i] = make_double2(g * ck[2], g * ck[0]);
      }
      __syncthreads();
      // the staged observations with a non-zero taper on any of this wave's columns, as a bit mask (wave-uniform)
      bool mine = false;
      if (lane < ne && any_live) {
        const int c_lo = ncw * wave, c_hi = (c_lo + ncw < kBlkCols) ? c_lo + ncw : kBlkCols;
        for (int c = c_lo; c < c_hi; ++c) mine = mine || (ab_s[lane * kBlkCols + c].x != 0.0);
      }
      unsigned long long todo = __ballot(mine);
      if (todo == 0ull) continue;
      const double2* abq = ab_s + (cq & (kBlkCols - 1));  // (a lane beyond the block's 16 columns holds a zero row: whatever it reads is multiplied by 0)
      const double* yq = ye_s + (lane & 15);
      int ee = __builtin_ctzll(todo);
      todo &= todo - 1;
      double2 ab = abq[ee * kBlkCols];
      double y[NG];
#pragma unroll
      for (int c = 0; c < NG; ++c) y[c] = yq[ee * YS + 16 * c];
      while (true) {
        const int en = (todo != 0ull) ? __builtin_ctzll(todo) : ee;  // the next one (after the last: itself again, harmlessly)
        const double dot = lane_dot<MP>(x, y, seq);      // :95
        const double2 abn = abq[en * kBlkCols];
        xm = __builtin_fma(ab.y, dot, xm);               // :115, :119, :130
        const double nkb = -(ab.x * dot);                // :115, :119, :136
        lane_update_prefetch<MP>(x, y, nkb, yq + en * YS);  // :141
        if (todo == 0ull) break;
        todo &= todo - 1;
        ee = en;
        ab = abn;
      }
    }
    if (live) {  // posterior members out (assimilation.py:168), or perturbations and mean
      double2* p = reinterpret_cast<double2*>(a.Xout + (size_t)row * M);
      const double add = FUSED ? xm : 0.0;
      if (M == MP) {
#pragma unroll
        for (int i = 0; i < MP / 2; ++i) p[i] = make_double2(x[2 * i] + add, x[2 * i + 1] + add);
      } else {
#pragma unroll
        for (int i = 0; i < MP / 2; ++i)
          if (i < M2) p[i] = make_double2(x[2 * i] + add, x[2 * i + 1] + add);
      }
      if (!FUSED) a.xout[row] = xm;
    }
  }
}

template <int MP>
hipError_t gc_lane_launch_one(const GcSweepArgs& a, hipStream_t s) {
  if (a.fused_members) hipLaunchKernelGGL((k_sweep_gc_lane<MP, true>), dim3((unsigned)(a.nblk * a.lead_split)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((k_sweep_gc_lane<MP, false>), dim3((unsigned)(a.nblk * a.lead_split)), dim3(256), 0, s, a);
  return hipGetLastError();
}
template <int... Q>
hipError_t gc_lane_dispatch(int q, const GcSweepArgs& a, hipStream_t s, std::integer_sequence<int, Q...>) {
  hipError_t r = hipErrorInvalidValue;
  (void)((q == Q + 1 ? (r = gc_lane_launch_one<4 * (Q + 1)>(a, s), true) : false) || ...);
  return r;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <int NC>
hipError_t gc_launch(const GcSweepArgs& a0, hipStream_t s) {
  GcSweepArgs a = a0;
  const bool vec = (a.M % 2 == 0) && (a.ye_stride % 2 == 0) && aligned16(a.Xin) && aligned16(a.Xout) && aligned16(a.Ye);
  constexpr int RPL = (NC <= 8) ? EFA_GC_RPL : (NC <= 10) ? EFA_GC_RPL_MID : (NC <= 13) ? EFA_GC_RPL_WIDE : (NC <= 16) ? EFA_GC_RPL_XWIDE : 1;  // rows per quad while they fit the register file
  // groups of slabs per column block: whole iterations of the slab loop (4 RPL slabs), as many as it takes to give
  // every CU a dozen workgroups, at most one group per iteration
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus = n;
  }
  const long iters = (a.n_lead + 4 * RPL - 1) / (4 * RPL);
  long split = (12L * cus + a.nblk - 1) / a.nblk;
  if (split > iters) split = iters;
  if (split < 1) split = 1;
  a.lead_chunk = ((iters + split - 1) / split) * (4 * RPL);
  a.lead_split = (int)((a.n_lead + a.lead_chunk - 1) / a.lead_chunk);
  const dim3 grid((unsigned)(a.nblk * a.lead_split)), block(256);
  if (a.fused_members) {
    if (vec) hipLaunchKernelGGL((k_sweep_gc<NC, true, true, RPL>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_sweep_gc<NC, false, true, RPL>), grid, block, 0, s, a);
  } else {
    if (vec) hipLaunchKernelGGL((k_sweep_gc<NC, true, false, RPL>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_sweep_gc<NC, false, false, RPL>), grid, block, 0, s, a);
  }
  return hipGetLastError();
}

}  // namespace

long gc_num_blocks(long ncol) { return (ncol + kBlkCols - 1) / kBlkCols; }

hipError_t launch_gc_bound(long ncol, long P, const double* glat, const double* ob_lat, const double* ob_hw,
                           const double* coef, int* ub, long* off, hipStream_t s) {
  const long nblk = gc_num_blocks(ncol);
  if (nblk <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_gc_bound, dim3((unsigned)((nblk + kBuildWaves - 1) / kBuildWaves)), dim3(64 * kBuildWaves), 0, s, ncol,
                     nblk, P, glat, ob_lat, ob_hw, coef, ub);
  hipLaunchKernelGGL(k_gc_scan, dim3(1), dim3(kScanThreads), 0, s, nblk, ub, off);
  return hipGetLastError();
}

hipError_t launch_gc_fill(long ncol, long P, const double* glat, const double* glon, const double* ob_lat,
                          const double* ob_lon, const double* ob_hw, const double* coef, double* obtrig, const long* off,
                          int* cnt, int* idx, double* wts, int* order, unsigned long long* npairs, hipStream_t s) {
  const long nblk = gc_num_blocks(ncol);
  if (nblk <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_gc_obtrig, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, P, ob_lat, ob_lon, ob_hw, obtrig);
  hipLaunchKernelGGL(k_gc_build<false>, dim3((unsigned)((nblk + kBuildWaves - 1) / kBuildWaves)), dim3(64 * kBuildWaves), 0, s,
                     ncol, nblk, P, glat, glon, ob_lat, ob_lon, ob_hw, coef, obtrig, cnt, off, idx, wts, npairs);
  int shift = 0;
  while ((P >> shift) >= kScanThreads) ++shift;
  hipLaunchKernelGGL(k_gc_order, dim3(1), dim3(kScanThreads), 0, s, nblk, cnt, shift, order);
  return hipGetLastError();
}

hipError_t launch_gc_count(long ncol, long P, const double* glat, const double* glon, const double* ob_lat,
                           const double* ob_lon, const double* ob_hw, const double* coef, double* obtrig, int* cnt,
                           int* blk_pairs, unsigned long long* npairs, hipStream_t s) {
  const long nblk = gc_num_blocks(ncol);
  if (nblk <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_gc_obtrig, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, P, ob_lat, ob_lon, ob_hw, obtrig);
  hipLaunchKernelGGL(k_gc_build<true>, dim3((unsigned)((nblk + kBuildWaves - 1) / kBuildWaves)), dim3(64 * kBuildWaves), 0, s,
                     ncol, nblk, P, glat, glon, ob_lat, ob_lon, ob_hw, coef, obtrig, cnt, nullptr, blk_pairs, nullptr, npairs);
  return hipGetLastError();
}

static int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus = n;
  }
  return cus;
}

hipError_t launch_sweep_gc(const GcSweepArgs& a0, hipStream_t s) {
  const GcSweepArgs& a = a0;
  if (a.M < 2 || a.M > kMaxMembers) return hipErrorInvalidValue;
  if (a.nblk <= 0 || a.n_lead <= 0) return hipSuccess;
  // 16-byte aligned rows of up to 104 members (an even number of them): the row-per-lane kernel
  if (EFA_GC_LANE && a.M <= kLaneMaxMembers && (a.M % 2 == 0) && (a.ye_stride % 2 == 0) && aligned16(a.Xin) &&
      aligned16(a.Xout) && aligned16(a.Ye)) {
    GcSweepArgs l = a;
    l.lead_split = (int)((l.n_lead + 15) / 16);  // groups of 16 slabs: one workgroup each
    l.lead_chunk = 16;
    return gc_lane_dispatch((l.M + 3) / 4, l, s, std::make_integer_sequence<int, kLaneMaxMembers / 4>{});
  }
  int nch = (a.M + 7) / 8;
  if (nch > 16) nch = (nch <= 20) ? 20 : (nch <= 24) ? 24 : 32;
  switch (nch) {
    case 1: return gc_launch<1>(a, s);
    case 2: return gc_launch<2>(a, s);
    case 3: return gc_launch<3>(a, s);
    case 4: return gc_launch<4>(a, s);
    case 5: return gc_launch<5>(a, s);
    case 6: return gc_launch<6>(a, s);
    case 7: return gc_launch<7>(a, s);
    case 8: return gc_launch<8>(a, s);
    case 9: return gc_launch<9>(a, s);
    case 10: return gc_launch<10>(a, s);
    case 11: return gc_launch<11>(a, s);
    case 12: return gc_launch<12>(a, s);
    case 13: return gc_launch<13>(a, s);
    case 14: return gc_launch<14>(a, s);
    case 15: return gc_launch<15>(a, s);
    case 16: return gc_launch<16>(a, s);
    case 20: return gc_launch<20>(a, s);
    case 24: return gc_launch<24>(a, s);
    case 32: return gc_launch<32>(a, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace efa
