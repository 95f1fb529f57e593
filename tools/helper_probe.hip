// Diagnostic: cost of the Gram helper's inner step (16 uniform-address ds_read_b128 + 32 fp64 FMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(int iters, double* out, unsigned long long* cyc) {
  __shared__ double2 rec[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) rec[i] = make_double2(1e-9 * i, 1e-9 * (i + 1));
  __syncthreads();
  double gr[16];
  for (int q = 0; q < 16; ++q) gr[q] = threadIdx.x + q;
  double kb = 1e-3 * threadIdx.x, t = 2e-3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const double2* r = rec + (it & 63) * 64;
    double2 ga[16];
    if (MODE == 0) {  // uniform-address b128
#pragma unroll
      for (int q = 0; q < 16; ++q) ga[q] = r[2 * q + (it & 1)];
    } else if (MODE == 1) {  // per-lane b128 (consecutive)
#pragma unroll
      for (int q = 0; q < 16; ++q) ga[q] = rec[((it + q) & 63) * 64 + threadIdx.x];
    } else if (MODE == 2) {  // uniform-address, two b64
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const double* p = reinterpret_cast<const double*>(r);
        ga[q].x = p[2 * q + (it & 1)];
        ga[q].y = p[64 + 2 * q + (it & 1)];
      }
    } else {  // readlane from a lane-distributed pair
      const double2 own = r[threadIdx.x];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = 2 * q + (it & 1);
        ga[q].x = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(own.x), i), __builtin_amdgcn_readlane(__double2loint(own.x), i));
        ga[q].y = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(own.y), i), __builtin_amdgcn_readlane(__double2loint(own.y), i));
      }
    }
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
      for (int q = 4 * q4; q < 4 * q4 + 4; ++q) gr[q] = __builtin_fma(-kb, ga[q].x, gr[q]);
#pragma unroll
      for (int q = 4 * q4; q < 4 * q4 + 4; ++q) gr[q] = __builtin_fma(-ga[q].y, t, gr[q]);
    }
    asm volatile("" : "+v"(kb), "+v"(t));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int q = 0; q < 16; ++q) s += gr[q];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 8192); hipMalloc(&cyc, 8);
  unsigned long long c;
  const int iters = 20000;
#define RUN(MODE, NAME) \
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, iters, out, cyc); hipDeviceSynchronize(); \
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-48s %.1f cyc/iter\n", NAME, c / (double)iters);
  RUN(0, "16 uniform-address ds_read_b128 + 32 fma");
  RUN(1, "16 per-lane ds_read_b128 + 32 fma");
  RUN(2, "32 uniform-address ds_read_b64 + 32 fma");
  RUN(3, "1 per-lane b128 + 64 v_readlane + 32 fma");
  return 0;
}
