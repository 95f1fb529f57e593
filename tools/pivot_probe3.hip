// Diagnostic: the Gram pivot step exactly as in efa_pipeline_gram.hip (same statements), in a 2-wave
// workgroup with a stand-in helper wave, to bisect what the step costs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
__device__ __forceinline__ double rl(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int g_ctl_lane(const int* p) {
  const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
  return v;
}
__device__ __forceinline__ int g_ctl(const int* p) { return __builtin_amdgcn_readfirstlane(g_ctl_lane(p)); }
__device__ __forceinline__ void g_ctl_set(int* p, int v) {
  asm volatile("" ::: "memory");
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
enum { cReady = 0, cBail = 1, cSReady = 2, cFwd = 3, cProg = 4, cHProg = 8 };
constexpr int kRowsWG = 64;
// MODE bits: 1 guard, 2 diag selects, 4 helper row hand-over (flag + row), 8 publication
typedef double v4f64_t __attribute__((ext_vector_type(4)));
template <int MODE, int TPB, int BIG = 0, int WARM_MFMA = 0>
__global__ __launch_bounds__(TPB) void k(int nblocks, double* out, unsigned long long* cyc, int* status) {
  extern __shared__ __align__(16) double lds[];
  double* G_s = lds;                                   // [64][64]
  double2* s_gk = reinterpret_cast<double2*>(G_s + 4096);  // [64][64]
  double* s_sc = G_s + 4096 + 8192;                    // [64][4]
  double* pv = s_sc + 256;                             // [3][64]
  int* ctl = reinterpret_cast<int*>(pv + 192);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nb = 64;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) G_s[i] = ((i >> 6) == (i & 63)) ? 900.0 : 1.0 + 1e-3 * (i & 63);
  if (threadIdx.x < 192) pv[threadIdx.x] = 1.0;
  const double invM = 0.01, rM1 = 1.0 / 99.0;
  long budget = 4000000;
  int polls = 0;
  u64 total = 0;
  double big[BIG > 0 ? BIG : 1];
  for (int q = 0; q < BIG; ++q) big[q] = out[(threadIdx.x + q) & 1023];
  for (int blk = 0; blk < nblocks; ++blk) {
    __syncthreads();
    if (threadIdx.x < 16) ctl[threadIdx.x] = (threadIdx.x >= cProg) ? -1 : 0;
    __syncthreads();
    if (wave >= 2) return;
    if (wave == 1) {  // stand-in helper: hands row kk+3 over as soon as record kk is there
      for (int kk = 0; kk + 3 < nb; ++kk) {
        while (g_ctl(&ctl[cSReady]) <= kk) {
          if (--budget < 0) return;
        }
        const int h = (kk + 3) & 1;
        G_s[(kk + 3) * kRowsWG + lane] = 1.0 + 1e-3 * lane + ((kk + 3) == lane ? 899.0 : 0.0);
        if (lane == 0) g_ctl_set(&ctl[cHProg + h], kk + 3);
      }
      continue;
    }
    if (WARM_MFMA) {  // as the kernel does before its loop: a Gram tile on the matrix core, two barriers
      v4f64_t acc = {0.0, 0.0, 0.0, 0.0};
      for (int q = 0; q < 26; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(G_s[q * 64 + lane], G_s[(q + 1) * 64 + lane], acc, 0, 0, 0);
      if (acc[0] == 12345.678) out[0] = acc[1];
    }
    const bool my_asm = true;
    const u64 asm_mask = __ballot(my_asm);
    const double thr = -1.0;
    double mu = 1e-12 * lane, xmv = 0.5;
    double o_pm = 0.0, o_pv = 0.0, o_in = 0.0, o_rd = 0.0, o_be = 0.0, o_km = 0.0;
    double g = G_s[lane], g1 = G_s[kRowsWG + lane];
    struct Pre { double val, err, sq, tw; };
    const bool gc = false;
    const double* twp = pv + lane;
    Pre pa{pv[0], pv[kRowsWG], pv[2 * kRowsWG], twp[0]}, pb{0.0, 1.0, 1.0, 1.0};
    auto give_up = [&]() { __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); g_ctl_set(&ctl[cBail], 1); };
    auto wait_row = [&](int kk, double& r2) {
      const int* flag = &ctl[cHProg + (kk & 1)];
      for (;;) {
        const int f = g_ctl_lane(flag);
        r2 = G_s[(kk + 2) * kRowsWG + lane];
        if (__builtin_amdgcn_readfirstlane(f) >= kk + 2) return true;
        if ((++polls & 15) == 0) {
          budget -= 16;
          if (g_ctl(&ctl[cBail]) != 0) return false;
          if (budget <= 0) { if (lane == 0) give_up(); return false; }
        }
      }
    };
    auto pivot_step = [&](const int kk, const bool has1, const bool has2, const bool poll, const Pre& cur, Pre& nxt) {
      const double valk = cur.val, errk = cur.err, sqk = cur.sq, twk = cur.tw;
      {
        const int kn = (kk + 1 < kRowsWG) ? kk + 1 : kk;
        nxt.val = pv[kn];
        nxt.err = pv[kRowsWG + kn];
        nxt.sq = pv[2 * kRowsWG + kn];
        nxt.tw = twp[gc ? kn * kRowsWG : 0];
      }
      const bool act = ((asm_mask >> kk) & 1) != 0;
      if (MODE & 1) {
        if (__builtin_expect(((__ballot(!(g > thr)) >> kk) & 1) != 0, 0)) { if (lane == 0) give_up(); return false; }
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
      double kc = g * rM1;
      kc = (gc ? twk : 1.0) * kc;
      const double km = act ? kc * rden : 0.0;
      const double kb = beta * km;
      const double innov = valk - xmk;
      int f_early = 0;
      double r2 = 1.0;
      if (has2 && (MODE & 4)) {
        f_early = poll ? g_ctl_lane(&ctl[cHProg + (kk & 1)]) : 0;
        r2 = G_s[(kk + 2) * kRowsWG + lane];
      }
      if (MODE & 8) {
        s_gk[kk * kRowsWG + lane] = make_double2(g, kb);
        if (lane == 0) {
          double2* sc = reinterpret_cast<double2*>(s_sc + (size_t)kk * 4);
          sc[0] = make_double2(innov, rden);
          sc[1] = make_double2(beta, act ? 1.0 : 0.0);
          g_ctl_set(&ctl[cSReady], kk + 1);
        }
      } else if (lane == 0) g_ctl_set(&ctl[cSReady], kk + 1);
      if ((MODE & 2) && lane == kk) {
        o_pm = xmk; o_pv = __builtin_fma(Gkk, invM, -mu2); o_in = innov; o_rd = rden; o_be = beta; o_km = km;
      }
      xmv = xmv + km * innov;
      mu = __builtin_fma(-kb, muk, mu);
      if (has1) {
        const double t = __builtin_fma(-kb, Gkk, g);
        const double gi = rl(g, kk + 1), ai = rl(kb, kk + 1);
        const double gnew = __builtin_fma(-ai, t, __builtin_fma(-kb, gi, g1));
        if (has2) {
          if ((MODE & 4) && __builtin_expect(poll && __builtin_amdgcn_readfirstlane(f_early) < kk + 2, 0)) {
            if (!wait_row(kk, r2)) return false;
          }
          const double gi2 = rl(g, kk + 2), ai2 = rl(kb, kk + 2);
          g1 = __builtin_fma(-ai2, t, __builtin_fma(-kb, gi2, r2));
        }
        g = gnew;
      }
      return true;
    };
#pragma unroll
    for (int q = 0; q < BIG; ++q) asm volatile("" : "+v"(big[q]));
    const u64 t0 = __builtin_amdgcn_s_memtime();
    {
      int kk = 0;
      bool ok = true;
      ok = pivot_step(0, true, true, false, pa, pb);
      for (kk = 1; ok && kk + 1 < nb - 2; kk += 2) {
        ok = pivot_step(kk, true, true, true, pb, pa);
        if (ok) ok = pivot_step(kk + 1, true, true, true, pa, pb);
      }
      for (; ok && kk < nb; ++kk) {
        if (kk & 1) ok = pivot_step(kk, kk + 1 < nb, kk + 2 < nb, kk >= 1, pb, pa);
        else ok = pivot_step(kk, kk + 1 < nb, kk + 2 < nb, kk >= 1, pa, pb);
      }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    total += t1 - t0;
    out[lane] = g + g1 + mu + xmv + o_pm + o_pv + o_in + o_rd + o_be + o_km;
  }
  double sb = 0;
  for (int q = 0; q < BIG; ++q) sb += big[q];
  if (BIG > 0) out[threadIdx.x] = sb;
  if (threadIdx.x == 0) cyc[0] = total;
}
int main() {
  double* out; unsigned long long* cyc; int* status;
  hipMalloc(&out, 8192); hipMalloc(&cyc, 8); hipMalloc(&status, 8); hipMemset(status, 0, 8);
  unsigned long long c;
  const int nblocks = 200;
  const size_t lds = (4096 + 8192 + 256 + 192) * 8 + 64;
#define RUNT(MODE, TPB, LDSB, NAME) \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, TPB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDSB)); \
  hipLaunchKernelGGL((k<MODE, TPB>), dim3(1), dim3(TPB), LDSB, 0, nblocks, out, cyc, status); (void)hipDeviceSynchronize(); \
  (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-56s %.0f cycles per step\n", NAME, c / (double)nblocks / 64);
#define RUN(MODE, NAME) \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
  hipLaunchKernelGGL((k<MODE, 128>), dim3(1), dim3(128), lds, 0, nblocks, out, cyc, status); (void)hipDeviceSynchronize(); \
  (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-56s %.0f cycles per step\n", NAME, c / (double)nblocks / 64);
  RUN(0, "arithmetic only");
  RUN(1, "+ guard");
  RUN(2, "+ diag selects");
  RUN(4, "+ helper row (flag + row read, poll)");
  RUN(8, "+ publication");
  RUN(15, "all (the kernel's step)");
  RUNT(15, 512, lds, "all, 512-thread workgroup (6 waves exit at once)");
  RUNT(15, 512, 150 * 1024, "all, 512 threads, 150 KB of LDS allocated");
  RUNT(15, 128, 150 * 1024, "all, 128 threads, 150 KB of LDS allocated");
#define RUNB(BIG, NAME) \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<15, 512, BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(150 * 1024)); \
  hipLaunchKernelGGL((k<15, 512, BIG>), dim3(1), dim3(512), 150 * 1024, 0, nblocks, out, cyc, status); (void)hipDeviceSynchronize(); \
  (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-56s %.0f cycles per step\n", NAME, c / (double)nblocks / 64);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<15, 512, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(150 * 1024));
  hipLaunchKernelGGL((k<15, 512, 0, 1>), dim3(1), dim3(512), 150 * 1024, 0, nblocks, out, cyc, status); (void)hipDeviceSynchronize();
  (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-56s %.0f cycles per step\n", "all, 512 threads, MFMAs before every block", c / (double)nblocks / 64);
  RUNB(40, "all, 512 threads, +80 live VGPRs");
  RUNB(90, "all, 512 threads, +180 live VGPRs");
  {  // wall-clock check of the tick unit
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int nb2 = 4000;
    hipLaunchKernelGGL((k<15, 128>), dim3(1), dim3(128), lds, 0, nb2, out, cyc, status); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<15, 128>), dim3(1), dim3(128), lds, 0, nb2, out, cyc, status);
    (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("wall check: %d blocks: %.3f ms wall = %.1f ns per step; %.0f ticks per step -> %.2f GHz tick rate\n", nb2, ms,
           ms * 1e6 / nb2 / 64, c / (double)nb2 / 64, (c / (double)nb2 / 64) / (ms * 1e6 / nb2 / 64));
  }
  return 0;
}
