// Diagnostic: intrinsic cost of the Gram pivot step's dependent chain (single wave, nothing else running).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rl(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ void k(int iters, double* out, unsigned long long* cyc) {
  const int lane = threadIdx.x;
  double g = 100.0 + lane, g1 = 90.0 + lane, mu = 1e-3 * lane, xmv = 0.5 * lane;
  const double invM = 0.01, rM1 = 1.0 / 99.0;
  double acc = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int kk = it & 63;
    const double errk = 1.0 + 1e-3 * kk, sqk = 1.0, valk = 0.25;
    if (MODE == 0) {  // full gain chain + tail
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
      xmv = xmv + km * innov;
      mu = __builtin_fma(-kb, muk, mu);
      const double t = __builtin_fma(-kb, Gkk, g);
      const int k1 = (kk + 1) & 63;
      const double gi = rl(g, k1), ai = rl(kb, k1);
      const double gnew = __builtin_fma(-ai, t, __builtin_fma(-kb, gi, g1));
      g1 = g * 0.999 + 1.0;
      g = gnew * 0.5 + 50.0 + lane;  // keep values sane
    } else if (MODE == 1) {  // only the rsq/rcp part, dependent on g
      const double Gkk = rl(g, kk);
      const double kdenom = __builtin_fma(Gkk, invM, errk);
      const double q0 = __builtin_amdgcn_rsq(kdenom);
      const double b0 = 1.0 + sqk * q0;
      const double r0 = __builtin_amdgcn_rcp(b0);
      g = g + r0;
    } else if (MODE == 2) {  // 10 dependent v_mul_f64 / v_add_f64
      double x = g;
#pragma unroll
      for (int q = 0; q < 5; ++q) { x = x * 1.0000001; x = x + 1e-9; }
      g = x;
    } else {  // readlane -> fma chain x4
      double x = g;
#pragma unroll
      for (int q = 0; q < 4; ++q) { const double s = rl(x, (kk + q) & 63); x = __builtin_fma(s, 1e-9, x); }
      g = x;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = g + g1 + mu + xmv + acc;
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
  RUN(0, "full pivot step (gain chain + tail)");
  RUN(1, "readlane, fma, rsq, mul, add, rcp, add (dependent)");
  RUN(2, "10 dependent mul/add");
  RUN(3, "4 x (readlane pair -> fma) dependent");
  return 0;
}
