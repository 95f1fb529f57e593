// Batched-obs dense path (BASELINE.json configs[4], SURVEY.md 8b `efa_cov_contract_f32`):
// all state-obs covariances of a batch at once,
//     C[i][k] = sum_m Xbp[i][m] * Ye[k][m]           (N x P) = (N x M) . (M x P), float32
// i.e. the numerator `np.dot(Xbp, ye.T)` of ensrf.py:95 for every recorded ye row in one
// (state x member) . (member x obs) contraction on the matrix cores.
//
// LDS-tiled GEMM on v_mfma_f32_32x32x2_f32 (exact f32 FMA chain, 64 FLOP/clk/SIMD):
//   - workgroup tile 128 rows x 128 obs, K stepped in chunks of 32 members;
//   - 4 waves, each a 64 x 64 sub-tile = 2 x 2 MFMA tiles of 32 x 32 (4 x 16 accumulator regs);
//   - both operands are row-major with the member fastest, so a chunk is fetched with 16-byte
//     loads (8 lanes cover one row's 128 B) and stored k-major in LDS with a 129-float row
//     stride: the transposing ds_write_b32 and the fragment ds_read_b32 are both conflict-free.
// The N x P output (16 GB at config 5) makes the kernel write-heavy as well as MFMA-bound.
#include "efa_device.h"
#include "efa_internal.h"

namespace efa {
namespace {

typedef float v16f32 __attribute__((ext_vector_type(16)));

constexpr int kBM = 128, kBN = 128, kBK = 32;
constexpr int kLd = kBM + 1;  // LDS row stride (floats) of a k-major chunk

__global__ __launch_bounds__(256) void k_contract_f32(long N, int M, long P, const float* __restrict__ X,
                                                      const float* __restrict__ Ye, float* __restrict__ C) {
  __shared__ float As[kBK * kLd];
  __shared__ float Bs[kBK * kLd];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const long col0 = (long)blockIdx.x * kBN;
  const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;  // this wave's 64 x 64 sub-tile
  const long nrb = (N + kBM - 1) / kBM;
  // blockIdx.x runs over obs blocks so that concurrently resident workgroups share one block
  // of state rows (read from HBM once) and walk the L2-resident Ye
  for (long rb = blockIdx.y; rb < nrb; rb += gridDim.y) {
  const long row0 = rb * kBM;

  v16f32 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

  // staging map: thread t moves 16 B = 4 members (quad q = t & 7) of row (t >> 3) + 32 * pass
  const int q = tid & 7, rr = tid >> 3;
  for (int k0 = 0; k0 < M; k0 += kBK) {
#pragma unroll
    for (int pass = 0; pass < kBM / 32; ++pass) {
      const int r = rr + 32 * pass;
      const int k = k0 + 4 * q;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
      if (k < M) {  // M % 4 == 0 is checked on the host
        if (row0 + r < N) va = *reinterpret_cast<const float4*>(X + (size_t)(row0 + r) * M + k);
        if (col0 + r < P) vb = *reinterpret_cast<const float4*>(Ye + (size_t)(col0 + r) * M + k);
      }
      float* pa = As + (4 * q) * kLd + r;
      float* pb = Bs + (4 * q) * kLd + r;
      pa[0] = va.x; pa[kLd] = va.y; pa[2 * kLd] = va.z; pa[3 * kLd] = va.w;
      pb[0] = vb.x; pb[kLd] = vb.y; pb[2 * kLd] = vb.z; pb[3 * kLd] = vb.w;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kBK / 2; ++s) {
      const int kk = 2 * s + (lane >> 5);  // A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31]
      const float a0 = As[kk * kLd + wr + (lane & 31)];
      const float a1 = As[kk * kLd + wr + 32 + (lane & 31)];
      const float b0 = Bs[kk * kLd + wc + (lane & 31)];
      const float b1 = Bs[kk * kLd + wc + 32 + (lane & 31)];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }

  // C/D map of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
      const long col = col0 + wc + 32 * jn + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const long row = row0 + wr + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < N && col < P) C[(size_t)row * P + col] = acc[i][jn][e];
      }
    }
  }
}


// ---- members <= 128: the state operand lives in registers ------------------------------------------
// With so few members the contraction is a stream of output tiles: per 32 x 32 tile only M / 2 matrix-core
// instructions, then 4 KB of output.  k_contract_f32_ra keeps a wave's 32 state rows as MFMA A fragments in
// registers for its whole life and streams the observations past them 128 at a time through a double-buffered
// LDS tile: no A traffic in the loop, the next tile's global loads are in flight during the current tile's
// 256 MFMAs (two halves of 64 observations, 2 independent accumulator tiles each), one barrier per 128 observations.  8 waves per
// workgroup (256 rows), one workgroup per CU (2 x 66 KB of LDS).
// The member order of the sum is permuted to make every fragment a contiguous run: lanes 0..31 of
// v_mfma_f32_32x32x2_f32 (k = 0) take members 0 .. KH-1 in turn, lanes 32..63 (k = 1) members KH .. 2 KH-1,
// so each step adds members s and KH + s, in that order (KH = M / 2 rounded up to a multiple of 8, zero padded).
// LDS rows are padded to 2 KH + 4 floats: the 16-byte fragment reads of 16 consecutive observations hit 64
// distinct banks.  Output goes out with non-temporal stores (16 GB at config 5; Ye stays in L2).
constexpr int kRaObs = 128;      // observations per LDS tile
constexpr int kRaThreads = 512;  // 8 waves x 32 state rows
constexpr int kRaRows = 32 * (kRaThreads / 64);
template <int KH>
__global__ __launch_bounds__(kRaThreads, 2) void k_contract_f32_ra(long N, int M, long P, int col_split,
                                                                   const float* __restrict__ X,
                                                                   const float* __restrict__ Ye, float* __restrict__ C) {
  constexpr int LD = 2 * KH + 4;
  constexpr int NL = kRaObs / (kRaThreads / 32);       // staging passes per tile: 16 observations each
  extern __shared__ __align__(16) float Bs[];         // [2][kRaObs][LD]
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;
  const long row0 = (long)blockIdx.x * kRaRows + __builtin_amdgcn_readfirstlane(wave) * 32;
  // this workgroup's share of the observations (col_split workgroups per row block)
  const long tiles = (P + kRaObs - 1) / kRaObs;
  const long per = (tiles + col_split - 1) / col_split;
  const long t_lo = (long)blockIdx.y * per, t_hi = (t_lo + per < tiles) ? t_lo + per : tiles;

  float a[KH];
  {
    const long row = row0 + i;
#pragma unroll
    for (int t = 0; t < KH / 4; ++t) {
      const int k = h * KH + 4 * t;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < N && k < M) v = *reinterpret_cast<const float4*>(X + (size_t)row * M + k);  // M % 4 == 0 (host)
      a[4 * t] = v.x;
      a[4 * t + 1] = v.y;
      a[4 * t + 2] = v.z;
      a[4 * t + 3] = v.w;
    }
  }

  // Staging map: pass u moves observations 16 u + (tid >> 5), lane (tid & 31) the float4 of members 4 (tid & 31) ..
  // (lanes beyond the padded row idle).  Every address is a wave-uniform base plus ONE per-lane 32-bit offset
  // that never changes, so the loop carries no 64-bit per-lane address arithmetic (it cost 66 spilled registers).
  float4 stage[NL];
  const int sj = tid >> 5, sk = 4 * (tid & 31);
  const bool s_in = sk < 2 * KH;   // inside the padded LDS row
  const bool s_ld = sk < M;        // inside the real row
  const unsigned g_off = (unsigned)sj * (unsigned)M + (unsigned)sk;  // floats, relative to the tile's first row
  const unsigned l_off = (unsigned)sj * LD + (unsigned)sk;
  auto fetch_tile = [&](long tl) {  // 128 observations x 2 KH members, zero padded; all loads in flight at once
    const long col0 = tl * kRaObs;
    const float* tile = Ye + (size_t)col0 * M;  // uniform
    const bool interior = col0 + kRaObs <= P;
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      stage[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* rowp = tile + (size_t)(16 * u) * M;  // uniform
      if (s_ld && (interior || col0 + 16 * u + sj < P)) stage[u] = *reinterpret_cast<const float4*>(rowp + g_off);
    }
  };
  auto put_tile = [&](float* buf) {
    if (s_in) {
#pragma unroll
      for (int u = 0; u < NL; ++u) *reinterpret_cast<float4*>(buf + 16 * u * LD + l_off) = stage[u];
    }
  };
  if (t_lo < t_hi) {
    fetch_tile(t_lo);
    put_tile(Bs);
  }
  __syncthreads();

  for (long tl = t_lo; tl < t_hi; ++tl) {
    const long col0 = tl * kRaObs;
    const int cur = (int)((tl - t_lo) & 1);
    const bool more = tl + 1 < t_hi;
    if (more) fetch_tile(tl + 1);
    // two halves of 64 observations, two independent accumulator tiles each (register budget: 256 per wave)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      v16f32 acc[2];
#pragma unroll
      for (int jn = 0; jn < 2; ++jn)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[jn][e] = 0.f;
      const float* bp = Bs + cur * (kRaObs * LD) + (64 * hf + i) * LD + h * KH;
#pragma unroll
      for (int t = 0; t < KH / 4; ++t) {
        float4 b[2];
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) b[jn] = *reinterpret_cast<const float4*>(bp + 32 * jn * LD + 4 * t);
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * t], b[jn].x, acc[jn], 0, 0, 0);
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * t + 1], b[jn].y, acc[jn], 0, 0, 0);
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * t + 2], b[jn].z, acc[jn], 0, 0, 0);
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * t + 3], b[jn].w, acc[jn], 0, 0, 0);
      }
      // C/D map of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
      const long colh = col0 + 64 * hf;
      if (row0 + 32 <= N && colh + 64 <= P) {  // interior (wave-uniform): no masks, uniform base + one lane offset
        float* cw = C + (size_t)row0 * P + colh;      // uniform
        const unsigned c_off = 4u * h * (unsigned)P + (unsigned)i;
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float* cu = cw + (size_t)((e & 3) + 8 * (e >> 2)) * P + 32 * jn;  // uniform
            __builtin_nontemporal_store(acc[jn][e], cu + c_off);
          }
      } else {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
          const long col = colh + 32 * jn + i;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const long row = row0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (row < N && col < P) __builtin_nontemporal_store(acc[jn][e], C + (size_t)row * P + col);
          }
        }
      }
    }
    if (more) put_tile(Bs + (cur ^ 1) * (kRaObs * LD));  // that buffer was last read before the previous barrier
    __syncthreads();
  }
}

template <int KH>
hipError_t contract_ra_launch(long N, int M, long P, const float* X, const float* Ye, float* C, hipStream_t s) {
  const long nrb = (N + kRaRows - 1) / kRaRows;
  // enough workgroups for ~16 rounds of one per CU: split the observations when there are few row blocks
  int split = 1;
  const long tiles = (P + kRaObs - 1) / kRaObs;
  while (nrb * split < 4096 && split * 2 <= tiles && split < 16) split *= 2;
  const size_t lds = (size_t)2 * kRaObs * (2 * KH + 4) * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_contract_f32_ra<KH>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_contract_f32_ra<KH>), dim3((unsigned)nrb, (unsigned)split), dim3(kRaThreads), lds, s, N, M, P, split,
                     X, Ye, C);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_contract_f32(long N, int M, long P, const float* X, const float* Ye, float* C, hipStream_t s) {
  if (N <= 0 || P <= 0) return hipSuccess;
  if (M < 4 || (M & 3) != 0) return hipErrorInvalidValue;
  if (M <= 128 && N <= (long)kRaRows * 2147483647L && P < (1L << 28)) {
    switch ((M + 15) / 16) {  // KH = 8 * ceil(M / 16)
      case 1: return contract_ra_launch<8>(N, M, P, X, Ye, C, s);
      case 2: return contract_ra_launch<16>(N, M, P, X, Ye, C, s);
      case 3: return contract_ra_launch<24>(N, M, P, X, Ye, C, s);
      case 4: return contract_ra_launch<32>(N, M, P, X, Ye, C, s);
      case 5: return contract_ra_launch<40>(N, M, P, X, Ye, C, s);
      case 6: return contract_ra_launch<48>(N, M, P, X, Ye, C, s);
      case 7: return contract_ra_launch<56>(N, M, P, X, Ye, C, s);
      default: return contract_ra_launch<64>(N, M, P, X, Ye, C, s);
    }
  }
  const long nrb = (N + kBM - 1) / kBM;
  const dim3 grid((unsigned)((P + kBN - 1) / kBN), (unsigned)(nrb < 65535 ? nrb : 65535));
  hipLaunchKernelGGL(k_contract_f32, grid, dim3(256), 0, s, N, M, P, X, Ye, C);
  return hipGetLastError();
}

}  // namespace efa
