// Diagnostic: fp64 FMA cost by operand pattern (single wave).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(int iters, double* out, unsigned long long* cyc, const double* in) {
  double a[16], g[16], h[16];
  for (int q = 0; q < 16; ++q) { a[q] = threadIdx.x + q; g[q] = in[threadIdx.x + 64 * q]; h[q] = in[threadIdx.x + 64 * q + 1024]; }
  double kb = in[threadIdx.x + 2048], t = in[threadIdx.x + 2112];
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) a[q] = __builtin_fma(a[q], kb, t);
    } else if (MODE == 1) {
#pragma unroll
      for (int q = 0; q < 16; ++q) a[q] = __builtin_fma(-kb, g[q], a[q]);
    } else if (MODE == 2) {
#pragma unroll
      for (int q = 0; q < 16; ++q) a[q] = __builtin_fma(-kb, g[q], a[q]);
#pragma unroll
      for (int q = 0; q < 16; ++q) a[q] = __builtin_fma(-h[q], t, a[q]);
    } else if (MODE == 3) {  // as the helper: groups of four
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
        for (int q = 4 * q4; q < 4 * q4 + 4; ++q) a[q] = __builtin_fma(-kb, g[q], a[q]);
#pragma unroll
        for (int q = 4 * q4; q < 4 * q4 + 4; ++q) a[q] = __builtin_fma(-h[q], t, a[q]);
      }
    } else if (MODE == 4) {  // v_mul / v_add mix
#pragma unroll
      for (int q = 0; q < 16; ++q) a[q] = a[q] * kb + g[q] * t;
    }
    asm volatile("" : "+v"(kb), "+v"(t));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int q = 0; q < 16; ++q) s += a[q];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double *out, *in; unsigned long long* cyc;
  hipMalloc(&out, 8192); hipMalloc(&in, 8 * 4096); hipMalloc(&cyc, 8);
  hipMemset(in, 0, 8 * 4096);
  const int iters = 20000;
  unsigned long long c;
#define RUN(MODE, NOPS, NAME) \
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, iters, out, cyc, in); hipDeviceSynchronize(); \
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-44s %.1f cyc/instr\n", NAME, c / (double)(NOPS) / iters);
  RUN(0, 16, "fma(a, kb, t) 16 chains");
  RUN(1, 16, "fma(-kb, g[q], a[q])");
  RUN(2, 32, "fma(-kb,g,a) x16 then fma(-h,t,a) x16");
  RUN(3, 32, "helper pattern (groups of 4)");
  RUN(4, 48, "mul+mul+add (3 instr)");
  return 0;
}
