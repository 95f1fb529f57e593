// Diagnostic: what the instruction kinds of the Phase-A pivot step cost ONE wave (one workgroup, one wave on the CU):
// v_readlane (uniform runtime lane) alone and feeding an fp64 FMA, fp64 FMA dependent / independent, v_rsq/v_rcp
// chains, ds_write_b128.  Cycles = s_memtime ticks per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rl(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ void k(int iters, int lane0, double* out, unsigned long long* cyc) {
  __shared__ double buf[4096];
  double a[8], acc = 0.0;
  for (int q = 0; q < 8; ++q) a[q] = 1.0 + 1e-3 * (threadIdx.x + q);
  double x = 1.0 + 1e-3 * threadIdx.x;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int l = (lane0 + it) & 63;
    if (MODE == 0) {  // 8 readlane pairs, results summed by SALU-free VALU adds (needed to keep them)
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += rl(a[q], l);
    } else if (MODE == 1) {  // 8 fp64 adds only (baseline for mode 0)
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += a[q];
    } else if (MODE == 2) {  // readlane -> fma dependent chain: x = fma(x, rl(x), c)
#pragma unroll
      for (int q = 0; q < 8; ++q) x = __builtin_fma(x, 1e-9, rl(x, l));
    } else if (MODE == 3) {  // fp64 fma dependent chain
#pragma unroll
      for (int q = 0; q < 8; ++q) x = __builtin_fma(x, 1.0000001, 1e-9);
    } else if (MODE == 4) {  // fp64 fma independent
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] = __builtin_fma(a[q], 1.0000001, 1e-9);
    } else if (MODE == 5) {  // rsq chain
#pragma unroll
      for (int q = 0; q < 8; ++q) x = __builtin_amdgcn_rsq(x) + 1.0;
    } else if (MODE == 6) {  // rcp chain
#pragma unroll
      for (int q = 0; q < 8; ++q) x = __builtin_amdgcn_rcp(x) + 1.0;
    } else if (MODE == 7) {  // ds_write_b128, 8 per iteration, no waits
#pragma unroll
      for (int q = 0; q < 8; ++q) reinterpret_cast<double2*>(buf)[threadIdx.x + 64 * q] = make_double2(a[q], x);
    } else if (MODE == 8) {  // lane-0 exec-masked ds_write_b128 x 4 + b32 (the pivot's scalar record)
      if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) reinterpret_cast<double2*>(buf)[1024 + q + 4 * (it & 7)] = make_double2(a[q], x);
        reinterpret_cast<int*>(buf)[4000 + (it & 7)] = it;
      }
      x = __builtin_fma(x, 1.0000001, 1e-9);
    } else if (MODE == 9) {  // v_cndmask pairs
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] = (l & (1 << q)) ? a[q] : x;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = acc + x + buf[threadIdx.x];
  for (int q = 0; q < 8; ++q) s += a[q];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 8192); hipMalloc(&cyc, 64);
  const int iters = 20000;
  const char* names[] = {"8 x (readlane pair + add)", "8 x add (baseline)", "8 x readlane pair -> dependent fma", "8 x dependent fma",
                         "8 x independent fma", "8 x (rsq + add) chain", "8 x (rcp + add) chain", "8 x ds_write_b128", "lane-0 record (4 b128 + b32) + fma", "8 x cndmask pair"};
#define RUN(M) { unsigned long long c; hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, iters, 3, out, cyc); hipDeviceSynchronize(); \
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-40s %7.1f cycles per iteration\n", names[M], (double)c / iters); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9)
  return 0;
}
