// Diagnostic: fp64 FMA issue rate per wave vs waves per SIMD (one workgroup of 1..16 waves on one CU).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int DEP>
__global__ void k(int iters, double* out, unsigned long long* cyc) {
  double a[16];
  for (int q = 0; q < 16; ++q) a[q] = threadIdx.x + q;
  const double m = 1.0 + 1e-9 * threadIdx.x, c = 1e-7;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (DEP) {
#pragma unroll
      for (int q = 0; q < 16; ++q) a[0] = __builtin_fma(a[0], m, c);
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) a[q] = __builtin_fma(a[q], m, c);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int q = 0; q < 16; ++q) s += a[q];
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int DEP>
__global__ void k32(int iters, float* out, unsigned long long* cyc) {
  float a[16];
  for (int q = 0; q < 16; ++q) a[q] = threadIdx.x + q;
  const float m = 1.0f + 1e-6f * threadIdx.x, c = 1e-7f;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) a[DEP ? 0 : q] = __builtin_fmaf(a[DEP ? 0 : q], m, c);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int q = 0; q < 16; ++q) s += a[q];
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 8192 * 2); hipMalloc(&cyc, 8 * 16);
  const int iters = 20000;
  for (int waves : {1, 2, 4, 8, 12, 16}) {
    unsigned long long c[16];
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), 0, 0, iters, out, cyc);
    hipDeviceSynchronize();
    hipMemcpy(c, cyc, 8 * waves, hipMemcpyDeviceToHost);
    double ind = c[0] / (16.0 * iters);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 0, 0, iters, out, cyc);
    hipDeviceSynchronize();
    hipMemcpy(c, cyc, 8 * waves, hipMemcpyDeviceToHost);
    double dep = c[0] / (16.0 * iters);
    hipLaunchKernelGGL(k32<0>, dim3(1), dim3(64 * waves), 0, 0, iters, (float*)out, cyc);
    hipDeviceSynchronize();
    hipMemcpy(c, cyc, 8 * waves, hipMemcpyDeviceToHost);
    double ind32 = c[0] / (16.0 * iters);
    printf("%2d waves/WG: fp64 fma independent %.1f cyc/instr/wave, dependent %.1f; fp32 independent %.1f\n", waves, ind, dep, ind32);
  }
  return 0;
}
