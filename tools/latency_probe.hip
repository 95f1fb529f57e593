// Dependent-chain latencies on gfx950 (one wave, one SIMD): fp64 FMA/add/mul, v_rsq_f64, v_rcp_f64,
// IEEE div/sqrt, DPP butterfly, LDS read round trip, LDS poll loop.  Also the raw accuracy of rsq/rcp.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define N 256
__device__ __forceinline__ double dppstep(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return x + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pollstep(int* flag, double x) {
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) {}
  return x + 1.0;
}
__device__ __forceinline__ void indep4(double& x, double& a, double& b, double& y) {
  x = __builtin_fma(x, y, 1e-300);
  a = __builtin_fma(a, 0.9999999, 1e-300);
  b = __builtin_fma(b, 1.0000001, 1e-300);
  y = __builtin_fma(y, 1.0, 1e-300);
}
__global__ void k(double* out, unsigned long long* t, double a, double b) {
  __shared__ double lds[512];
  __shared__ int flag;
  lds[threadIdx.x] = a + threadIdx.x;
  flag = 0;
  __syncthreads();
  unsigned long long t0, t1;
  double x = a + threadIdx.x * 1e-9, y = b;
  int ti = 0;
#define TIME(name, body)                                                   \
  t0 = __builtin_amdgcn_s_memtime();                                       \
  _Pragma("unroll 1") for (int i = 0; i < N; ++i) { body; }               \
  t1 = __builtin_amdgcn_s_memtime();                                       \
  if (threadIdx.x == 0) t[ti] = t1 - t0;                                   \
  ti++;
  TIME("fma", x = __builtin_fma(x, y, 1e-300))
  TIME("add", x = x + y)
  TIME("mul", x = x * 1.0000001)
  TIME("rsq", x = __builtin_amdgcn_rsq(x + 2.0))
  TIME("rcp", x = __builtin_amdgcn_rcp(x + 2.0))
  TIME("div", x = 1.0 / (x + 2.0))
  TIME("sqrt", x = sqrt(x + 2.0))
  TIME("dpp3", x = dppstep(x))
  TIME("lds_rt", x = lds[((int)x) & 255] + 1.0)
  TIME("lds_poll", x = pollstep(&flag, x))
  TIME("fma_indep4", indep4(x, a, b, y))
  out[threadIdx.x] = x + a + b + y;
  // accuracy
  if (threadIdx.x < 64) {
    double v = 0.37 + 1.913 * threadIdx.x;
    out[256 + threadIdx.x] = __builtin_amdgcn_rsq(v) * sqrt(v) - 1.0;
    out[320 + threadIdx.x] = __builtin_amdgcn_rcp(v) * v - 1.0;
  }
}
int main() {
  double* out; unsigned long long* t;
  (void)hipMalloc(&out, 512 * 8); (void)hipMalloc(&t, 64 * 8);
  k<<<1, 64>>>(out, t, 1.0000001, 0.9999999);
  k<<<1, 64>>>(out, t, 1.0000001, 0.9999999);
  unsigned long long ht[64]; double ho[512];
  (void)hipMemcpy(ht, t, sizeof ht, hipMemcpyDeviceToHost); (void)hipMemcpy(ho, out, sizeof ho, hipMemcpyDeviceToHost);
  const char* names[] = {"fma f64 (dependent)", "add f64", "mul f64", "v_rsq_f64 (+add)", "v_rcp_f64 (+add)", "IEEE 1/x (+add)", "IEEE sqrt (+add)", "dpp step (2 mov_dpp + add)", "LDS read round trip (+cvt,add)", "LDS poll (flag already set)", "4 independent fma"};
  for (int i = 0; i < 11; ++i) printf("%-34s %7.1f ticks/iter\n", names[i], ht[i] / (double)N);
  double mr = 0, mc = 0;
  for (int i = 0; i < 64; ++i) { mr = fmax(mr, fabs(ho[256 + i])); mc = fmax(mc, fabs(ho[320 + i])); }
  printf("max rel err: v_rsq_f64 %.3e   v_rcp_f64 %.3e\n", mr, mc);
  return 0;
}
