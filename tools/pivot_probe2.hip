// Diagnostic: the Gram pivot step with its LDS traffic, alone and next to polling waves.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rl(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int ctl_load(const int* p) {
  const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
  return v;
}
__device__ __forceinline__ void ctl_set(int* p, int v) {
  asm volatile("" ::: "memory");
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// MODE bit 0: LDS publication + reads as in the kernel; bit 1: other waves poll (s_sleep 1); bit 2: pollers do not sleep;
// bit 3: s_setprio(3) for the pivot wave
template <int MODE, int HI = 0>
__global__ void k(int iters, double* out, unsigned long long* cyc) {
  extern __shared__ double lds[];
  double* s_km = lds + (HI ? 10240 : 0);                 // [64][64]
  double2* s_gk = reinterpret_cast<double2*>(lds + 4096);  // [64][64]
  double* s_sc = lds + 4096 + 8192;   // [64][8]
  double* G_s = s_sc + 512;           // [64][64]
  double* pv = G_s + 4096;            // [3][64]
  int* ctl = reinterpret_cast<int*>(pv + 192);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) G_s[i] = 100.0 + (i & 63);
  if (threadIdx.x < 192) pv[threadIdx.x] = 1.0;
  if (threadIdx.x < 16) ctl[threadIdx.x] = 0;
  __syncthreads();
  if (wave != 0) {
    // pollers: follow the pivot's step counter
    int seen = 0;
    long guard = 0;
    while (seen < iters && ++guard < 100000000L) {
      const int v = __builtin_amdgcn_readfirstlane(ctl_load(&ctl[2]));
      if (v > seen) seen = v;
      if (!(MODE & 4)) __builtin_amdgcn_s_sleep(1);
    }
    return;
  }
  if (MODE & 8) __builtin_amdgcn_s_setprio(3);
  double g = 100.0 + lane, g1 = 90.0 + lane, mu = 1e-3 * lane, xmv = 0.5 * lane;
  const double invM = 0.01, rM1 = 1.0 / 99.0;
  double valn = pv[0], errn = pv[64], sqn = pv[128];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int kk = it & 63;
    const double valk = valn, errk = errn, sqk = sqn;
    if (MODE & 1) {
      const int kn = (kk + 1) & 63;
      valn = pv[kn];
      errn = pv[64 + kn];
      sqn = pv[128 + kn];
    }
    const double Gkk = rl(g, kk), muk = rl(mu, kk), xmk = rl(xmv, kk);
    const double mu2 = muk * muk;
    const double kdenom = __builtin_fma(Gkk, invM, errk - mu2);
    const double q0 = __builtin_amdgcn_rsq(kdenom);
    const double e = __builtin_fma(-kdenom * q0, q0, 1.0);
    const double d = e * __builtin_fma(0.375, e, 0.5);
    const double q = __builtin_fma(q0, d, q0);
    const double rden = q * q;
    const double sq0 = sqk * q0;
    const double b0 = 1.0 + sq0;
    const double r0 = __builtin_amdgcn_rcp(b0);
    const double eb = __builtin_fma(-b0, r0, 1.0);
    const double beta0 = __builtin_fma(r0, __builtin_fma(eb, eb, eb), r0);
    const double beta = __builtin_fma(-((beta0 * beta0) * sq0), d, beta0);
    const double kc = g * rM1;
    const double km = kc * rden;
    const double kb = beta * km;
    const double innov = valk - xmk;
    double r2 = 95.0 + lane;
    if (MODE & 1) {
      r2 = G_s[((kk + 2) & 63) * 64 + lane];
      s_km[kk * 64 + lane] = km;
      s_gk[kk * 64 + lane] = make_double2(g, kb);
      if (lane == 0) {
        double2* sc = reinterpret_cast<double2*>(s_sc + kk * 8);
        sc[0] = make_double2(xmk, muk);
        sc[1] = make_double2(innov, rden);
        sc[2] = make_double2(beta, 1.0);
        sc[3] = make_double2(kdenom, Gkk);
        ctl_set(&ctl[2], it + 1);
      }
    }
    xmv = xmv + km * innov;
    mu = __builtin_fma(-kb, muk, mu);
    const double t = __builtin_fma(-kb, Gkk, g);
    const int k1 = (kk + 1) & 63, k2 = (kk + 2) & 63;
    const double gi = rl(g, k1), ai = rl(kb, k1);
    const double gnew = __builtin_fma(-ai, t, __builtin_fma(-kb, gi, g1));
    const double gi2 = rl(g, k2), ai2 = rl(kb, k2);
    g1 = __builtin_fma(-ai2, t, __builtin_fma(-kb, gi2, r2)) * 0.0 + 90.0 + lane;
    g = gnew * 0.0 + 100.0 + lane + 1e-9 * gnew;  // keep values sane
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[lane] = g + g1 + mu + xmv;
  if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 8192); hipMalloc(&cyc, 8);
  unsigned long long c;
  const int iters = 20000;
  size_t lds = (4096 + 8192 + 512 + 4096 + 192) * 8 + 64;
#define RUN(MODE, THREADS, NAME) \
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(THREADS), lds, 0, iters, out, cyc); hipDeviceSynchronize(); \
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-60s %.1f cyc/step\n", NAME, c / (double)iters);
  RUN(0, 64, "chain only, one wave");
  RUN(1, 64, "chain + LDS publication/reads, one wave");
  RUN(3, 512, "chain + LDS, 7 polling waves (s_sleep 1)");
  RUN(7, 512, "chain + LDS, 7 polling waves (no sleep)");
  RUN(11, 512, "chain + LDS, 7 polling waves (s_sleep 1), setprio 3");
  RUN(15, 512, "chain + LDS, 7 polling waves (no sleep), setprio 3");
  RUN(3, 128, "chain + LDS, 1 polling wave (s_sleep 1)");
  lds = 150 * 1024;
  RUN(1, 64, "chain + LDS, one wave, 150 KB LDS allocated");
  RUN(3, 512, "chain + LDS, 7 polling waves, 150 KB LDS allocated");
#define RUNH(MODE, THREADS, NAME) \
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
  hipLaunchKernelGGL((k<MODE, 1>), dim3(1), dim3(THREADS), lds, 0, iters, out, cyc); hipDeviceSynchronize(); \
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-60s %.1f cyc/step\n", NAME, c / (double)iters);
  RUNH(1, 64, "chain + LDS, one wave, arrays above 80 KB");
  return 0;
}
