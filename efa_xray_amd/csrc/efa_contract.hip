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

}  // namespace

hipError_t launch_contract_f32(long N, int M, long P, const float* X, const float* Ye, float* C, hipStream_t s) {
  if (N <= 0 || P <= 0) return hipSuccess;
  if (M < 4 || (M & 3) != 0) return hipErrorInvalidValue;
  const long nrb = (N + kBM - 1) / kBM;
  const dim3 grid((unsigned)((P + kBN - 1) / kBN), (unsigned)(nrb < 65535 ? nrb : 65535));
  hipLaunchKernelGGL(k_contract_f32, grid, dim3(256), 0, s, N, M, P, X, Ye, C);
  return hipGetLastError();
}

}  // namespace efa
