// Diagnostic: do fp64 VALU FMAs and fp64 MFMAs of two waves on one SIMD overlap?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
// waves 0..3 (one per SIMD): MFMA stream if (mode & 1); waves 4..7 (second wave per SIMD): VALU fp64 stream if (mode & 2)
__global__ void k(int mode, int iters, double* out, unsigned long long* cyc) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if (mode & 1) {
      v4f64 acc[7];
      for (int t = 0; t < 7; ++t) acc[t] = (v4f64){0, 0, 0, 0};
      double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 7; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
      }
      double s = 0;
      for (int t = 0; t < 7; ++t) s += acc[t][0] + acc[t][3];
      out[threadIdx.x] = s;
    }
  } else {
    if (mode & 2) {
      double x[14];
      for (int q = 0; q < 14; ++q) x[q] = lane + q;
      const double m = 1.0 + 1e-9 * lane, c = 1e-7;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 14; ++q) x[q] = __builtin_fma(x[q], m, c);
      }
      double s = 0;
      for (int q = 0; q < 14; ++q) s += x[q];
      out[threadIdx.x] = s;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[wave] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 8192); hipMalloc(&cyc, 64);
  const int iters = 20000;
  unsigned long long c[8];
  for (int mode : {1, 2, 3}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, mode, iters, out, cyc);
    hipDeviceSynchronize();
    hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("mode %d: MFMA wave %.1f cycles per 7 MFMAs (%.1f each); VALU wave %.1f cycles per 14 FMAs (%.1f each)\n", mode,
           c[0] / (double)iters, c[0] / (double)iters / 7, c[4] / (double)iters, c[4] / (double)iters / 14);
  }
  return 0;
}
