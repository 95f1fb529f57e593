// Diagnostic: issue rate of v_mfma_f64_4x4x4_4b_f64 vs v_mfma_f64_16x16x4_f64 (one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(int iters, double* out, unsigned long long* cyc) {
  const int lane = threadIdx.x;
  double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0) {
    double acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0;
    for (int t = 0; t < 8; ++t) s += acc[t];
    out[lane] = s;
  } else {
    v4f64 acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = (v4f64){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0;
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][3];
    out[lane] = s;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 8192); (void)hipMalloc(&cyc, 8);
  unsigned long long c;
  const int iters = 20000;
  hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, iters, out, cyc); (void)hipDeviceSynchronize();
  (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("v_mfma_f64_4x4x4_4b:  %.1f cycles each (512 flop)\n", c / (double)iters / 8);
  hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, iters, out, cyc); (void)hipDeviceSynchronize();
  (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("v_mfma_f64_16x16x4:   %.1f cycles each (2048 flop)\n", c / (double)iters / 8);
  return 0;
}
