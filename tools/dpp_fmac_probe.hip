// Diagnostic: issue rate of v_fmac_f64_dpp (row_newbcast: one lane of each 16-lane row broadcast as a multiplicand) against
// the plain v_fma_f64, per wave and with 1, 2, 3 waves per SIMD; and a check of what row_newbcast delivers.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int mode, int iters, double* out, unsigned long long* cyc) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double x[16];
  for (int q = 0; q < 16; ++q) x[q] = lane + q;
  double y0 = 1e-9 * (lane + 1), y1 = 2e-9 * (lane + 1);
  const double m = 1e-7 * lane;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (mode == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 16; ++q) x[q] = __builtin_fma(y0, m, x[q]);
    }
  } else if (mode == 1) {
    for (int it = 0; it < iters; ++it) {
#define F(q, l) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #l " row_mask:0xf bank_mask:0xf" : "+v"(x[q]) : "v"(y0), "v"(m));
      F(0, 0) F(1, 1) F(2, 2) F(3, 3) F(4, 4) F(5, 5) F(6, 6) F(7, 7) F(8, 8) F(9, 9) F(10, 10) F(11, 11) F(12, 12) F(13, 13) F(14, 14) F(15, 15)
    }
  } else {  // dependent accumulation into 2 chains (the dot)
    double a0 = 0, a1 = 0;
    for (int it = 0; it < iters; ++it) {
#define G(acc, q, l) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #l " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(y1), "v"(x[q]));
      G(a0, 0, 0) G(a1, 1, 1) G(a0, 2, 2) G(a1, 3, 3) G(a0, 4, 4) G(a1, 5, 5) G(a0, 6, 6) G(a1, 7, 7)
      G(a0, 8, 8) G(a1, 9, 9) G(a0, 10, 10) G(a1, 11, 11) G(a0, 12, 12) G(a1, 13, 13) G(a0, 14, 14) G(a1, 15, 15)
    }
    x[0] = a0 + a1;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int q = 0; q < 16; ++q) s += x[q];
  out[threadIdx.x] = s;
  if (lane == 0) cyc[wave] = t1 - t0;
}
__global__ void layout(double* out) {
  const int lane = threadIdx.x;
  double acc = 0.0, y = 100.0 + lane, one = 1.0;
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(y), "v"(one));
  out[lane] = acc;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 16 * 8);
  const int iters = 20000;
  unsigned long long c[16];
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves : {4, 8, 12}) {
    for (int mode : {0, 1, 2}) {
      hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, mode, iters, out, cyc);
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, mode, iters, out, cyc);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
      const double per_simd = (double)iters * 16 * (waves / 4);   // instructions per SIMD
      printf("%d waves per CU, %-28s: %.3f ms, %.2f ns per instruction per SIMD (= %.2f cycles at 2.4 GHz), %.1f TFLOP/s; s_memtime %.2f ticks per instruction per wave\n",
             waves, mode == 0 ? "v_fma_f64 independent" : mode == 1 ? "v_fmac_f64_dpp independent" : "v_fmac_f64_dpp two chains", ms,
             ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, per_simd * 1024 * 128 / (ms * 1e-3) * 1e-12, c[0] / (double)iters / 16);
    }
  }
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, out);
  double h[64];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("row_newbcast:5 delivers:");
  for (int l = 0; l < 64; l += 7) printf(" lane %d <- %.0f", l, h[l]);
  printf("\n");
  return 0;
}
