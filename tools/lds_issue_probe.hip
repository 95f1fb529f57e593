// Diagnostic: issue cost of LDS instructions for one wave (no waits on the results inside the loop).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(int iters, double* out, unsigned long long* cyc) {
  extern __shared__ double lds_base[];
  double* lds = lds_base;
  const int lane = threadIdx.x & 63;
  lds += (threadIdx.x >> 6) * 1024;  // every wave its own 8 KB region
  double2* p2 = reinterpret_cast<double2*>(lds);
  double acc = 0;
  const double2 v = make_double2(1.0 + lane, 2.0 + lane);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int base = 0;
    if (MODE == 0) {  // 8 x ds_write_b128, all lanes
#pragma unroll
      for (int q = 0; q < 8; ++q) p2[base + q * 64 + lane] = v;
    } else if (MODE == 1) {  // 8 x ds_write_b128, lane 0 only
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) p2[base + q * 64] = v;
      }
    } else if (MODE == 2) {  // 8 x ds_write_b64 all lanes
#pragma unroll
      for (int q = 0; q < 8; ++q) lds[base + q * 64 + lane] = v.x;
    } else if (MODE == 3) {  // 8 x ds_write_b32 lane 0 (flags)
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) reinterpret_cast<int*>(lds)[q * 16] = it;
      }
    } else if (MODE == 4) {  // 8 x ds_read_b64 all lanes, consumed once at the end of the iteration
      double s = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += lds[base + q * 64 + lane];
      acc += s;
    } else if (MODE == 5) {  // 8 x ds_read_b128
      double s = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) { const double2 r = p2[base + q * 64 + lane]; s += r.x + r.y; }
      acc += s;
    }
    asm volatile("" ::: "memory");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 8192); hipMalloc(&cyc, 8);
  unsigned long long c;
  const int iters = 20000;
  const size_t lds = 8 * 8192 + 1024;
#define RUN(MODE, NAME) \
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), lds, 0, iters, out, cyc); hipDeviceSynchronize(); \
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-44s %.1f cycles per instruction\n", NAME, c / (double)iters / 8);
#define RUNW(MODE, W, NAME) \
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * W), lds, 0, iters, out, cyc); hipDeviceSynchronize(); \
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-44s %d waves: %.1f cycles per instruction per wave\n", NAME, W, c / (double)iters / 8);
  for (int w : {1, 2, 4, 8}) {
    if (w == 1) { RUNW(2, 1, "ds_write_b64") RUNW(4, 1, "ds_read_b64") RUNW(0, 1, "ds_write_b128") RUNW(5, 1, "ds_read_b128") }
    if (w == 2) { RUNW(2, 2, "ds_write_b64") RUNW(4, 2, "ds_read_b64") RUNW(0, 2, "ds_write_b128") RUNW(5, 2, "ds_read_b128") }
    if (w == 4) { RUNW(2, 4, "ds_write_b64") RUNW(4, 4, "ds_read_b64") RUNW(0, 4, "ds_write_b128") RUNW(5, 4, "ds_read_b128") }
    if (w == 8) { RUNW(2, 8, "ds_write_b64") RUNW(4, 8, "ds_read_b64") RUNW(0, 8, "ds_write_b128") RUNW(5, 8, "ds_read_b128") }
  }
  RUN(0, "ds_write_b128, 64 lanes");
  RUN(1, "ds_write_b128, lane 0 only");
  RUN(2, "ds_write_b64, 64 lanes");
  RUN(3, "ds_write_b32, lane 0 only");
  RUN(4, "ds_read_b64, 64 lanes (8 in flight)");
  RUN(5, "ds_read_b128, 64 lanes (8 in flight)");
  return 0;
}
