// Development microbenchmark for the Phase-B transform kernel (not product code).
// Times ablations of efa_transform.hip's structure on random data so that the
// product kernel can be tuned against measurements:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/transform_lab.hip -o tools/transform_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v4f64 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int NU = 13, NT = 7, M = 100;

template <int THREADS, bool DO_LOAD, bool DO_MFMA, bool DO_STORE, int STORE_MODE, int BARRIER_EVERY>
__global__ __launch_bounds__(THREADS) void k_t(const double* __restrict__ Xin, double* __restrict__ Xout,
                                               const double* __restrict__ T, long nrows) {
  extern __shared__ __align__(16) double Bs[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 2 * NU * NT * 64; i += THREADS) {
    const int l = i & 63, st = i >> 6, t = st % NT, s = st / NT, u = s >> 1, h = s & 1, g = l >> 4, n = l & 15;
    const int m = 8 * u + 2 * g + h, j = 16 * t + n;
    Bs[i] = (m < M && j < M) ? T[m * M + j] : 0.0;
  }
  __syncthreads();
  const int lane = tid & 63, g = lane >> 4, n = lane & 15;
  const long ntiles = (nrows + 15) / 16;
  const long wave = (long)blockIdx.x * (THREADS / 64) + (tid >> 6);
  const long nwaves = (long)gridDim.x * (THREADS / 64);
  double a[2 * NU], an[2 * NU];
  auto load = [&](long tile, double (&d)[2 * NU]) {
    const double* p = Xin + (size_t)(tile * 16 + n) * M;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int m0 = 8 * u + 2 * g;
      if (DO_LOAD && m0 < M) { const double2 v = *reinterpret_cast<const double2*>(p + m0); d[2*u] = v.x; d[2*u+1] = v.y; }
      else { d[2*u] = 1.0 + lane; d[2*u+1] = 0.5; }
    }
  };
  long tile = wave;
  if (tile < ntiles) load(tile, a);
  while (tile < ntiles) {
    const long next = tile + nwaves;
    if (next < ntiles) load(next, an);
    v4f64 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (v4f64){0, 0, 0, 0};
    if (DO_MFMA) {
#pragma unroll
      for (int s = 0; s < 2 * NU; ++s) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const double b = Bs[((size_t)s * NT + t) * 64 + lane];
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b, acc[t], 0, 0, 0);
        }
        if (BARRIER_EVERY && (s % BARRIER_EVERY) == BARRIER_EVERY - 1) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = (v4f64){a[t], a[t + 1], a[t + 2], a[t + 3]};
    }
    const long r0 = tile * 16;
    if (DO_STORE) {
      if (STORE_MODE == 0) {          // 8-byte stores straight from the accumulator layout
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int col = 16 * t + n;
          if (col < M) {
#pragma unroll
            for (int v = 0; v < 4; ++v) Xout[(size_t)(r0 + 4 * v + g) * M + col] = acc[t][v];
          }
        }
      } else {                        // pair lanes (n even/odd) -> 16-byte stores
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int colb = 16 * t + (n & ~1);
#pragma unroll
          for (int vp = 0; vp < 2; ++vp) {
            // even lane keeps row v=2vp and receives partner's v=2vp; odd lane keeps v=2vp+1
            const double mine_e = acc[t][2 * vp], mine_o = acc[t][2 * vp + 1];
            const double send = (n & 1) ? mine_e : mine_o;
            int lo = __double2loint(send), hi = __double2hiint(send);
            lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
            hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
            const double recv = __hiloint2double(hi, lo);
            const int v = 2 * vp + (n & 1);
            const double2 out = (n & 1) ? make_double2(recv, mine_o) : make_double2(mine_e, recv);
            if (colb < M) *reinterpret_cast<double2*>(Xout + (size_t)(r0 + 4 * v + g) * M + colb) = out;
          }
        }
      }
    } else {
      double s = 0;
#pragma unroll
      for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
      if (s == 123.456) Xout[lane] = s;
    }
#pragma unroll
    for (int c = 0; c < 2 * NU; ++c) a[c] = an[c];
    tile = next;
  }
}


// ---- variant 2: column-tile outer loop, ACCS accumulators in flight, each tile stored as soon as done
template <int THREADS, int ACCS, int STAGGER>
__global__ __launch_bounds__(THREADS) void k_t2(const double* __restrict__ Xin, double* __restrict__ Xout,
                                                const double* __restrict__ T, long nrows) {
  extern __shared__ __align__(16) double Bs[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 2 * NU * NT * 64; i += THREADS) {
    const int l = i & 63, st = i >> 6, s = st % (2 * NU), t = st / (2 * NU), u = s >> 1, h = s & 1, g = l >> 4, n = l & 15;
    const int m = 8 * u + 2 * g + h, j = 16 * t + n;
    Bs[i] = (m < M && j < M) ? T[m * M + j] : 0.0;   // layout [t][s][lane]
  }
  __syncthreads();
  const int lane = tid & 63, g = lane >> 4, n = lane & 15;
  const long ntiles = (nrows + 15) / 16;
  const long wave = (long)blockIdx.x * (THREADS / 64) + (tid >> 6);
  const long nwaves = (long)gridDim.x * (THREADS / 64);
  double a[2 * NU], an[2 * NU];
  auto load = [&](long tile, double (&d)[2 * NU]) {
    const double* p = Xin + (size_t)(tile * 16 + n) * M;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      int m0 = 8 * u + 2 * g;
      if (u == NU - 1) m0 = (m0 < M) ? m0 : M - 2;
      const double2 v = *reinterpret_cast<const double2*>(p + m0);
      d[2 * u] = v.x; d[2 * u + 1] = v.y;
    }
  };
  if (STAGGER && (tid >> 6) >= (THREADS / 128)) __builtin_amdgcn_s_sleep(127);
  long tile = wave;
  if (tile < ntiles) load(tile, a);
  const bool last_ok = (8 * (NU - 1) + 2 * g) < M;
  while (tile < ntiles) {
    const long next = tile + nwaves;
    load(next < ntiles ? next : tile, an);
    if (!last_ok) { a[2 * NU - 2] = 0; a[2 * NU - 1] = 0; }
    const long r0 = tile * 16;
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += ACCS) {
      v4f64 acc[ACCS];
#pragma unroll
      for (int q = 0; q < ACCS; ++q) acc[q] = (v4f64){0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 2 * NU; ++s) {
#pragma unroll
        for (int q = 0; q < ACCS; ++q) {
          if (t0 + q < NT) {
            const double b = Bs[((size_t)(t0 + q) * (2 * NU) + s) * 64 + lane];
            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b, acc[q], 0, 0, 0);
          }
        }
        if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < ACCS; ++q) {
        const int col = 16 * (t0 + q) + n;
        if (t0 + q < NT && col < M) {
#pragma unroll
          for (int v = 0; v < 4; ++v) Xout[(size_t)(r0 + 4 * v + g) * M + col] = acc[q][v];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 2 * NU; ++c) a[c] = an[c];
    tile = next;
  }
}

template <typename K>
float run(K kern, int threads, int grid, const double* X, double* Y, const double* T, long rows, const char* name) {
  const size_t lds = (size_t)2 * NU * NT * 64 * 8;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, 0, X, Y, T, rows);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, 0, X, Y, T, rows);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("%-44s threads %3d grid %4d : %7.3f ms  (%6.0f GB/s r+w, %5.1f TF useful)\n", name, threads, grid, ms,
         16.0 * rows * M / ms / 1e6, 2.0 * rows * M * M / ms / 1e9);
  return ms;
}

__global__ void k_copy(const double2* __restrict__ a, double2* __restrict__ b, size_t n2) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main(int argc, char** argv) {
  const long rows = argc > 1 ? atol(argv[1]) : 10000000L;
  double *X, *Y, *T;
  CK(hipMalloc(&X, (size_t)rows * M * 8)); CK(hipMalloc(&Y, (size_t)rows * M * 8)); CK(hipMalloc(&T, M * M * 8));
  std::vector<double> h((size_t)M * M);
  for (auto& v : h) v = (rand() / (double)RAND_MAX - 0.5) * 0.1;
  CK(hipMemcpy(T, h.data(), M * M * 8, hipMemcpyHostToDevice));
  std::vector<double> hx(1 << 20);
  for (auto& v : hx) v = rand() / (double)RAND_MAX - 0.5;
  for (size_t off = 0; off < (size_t)rows * M; off += hx.size()) {
    size_t n = std::min(hx.size(), (size_t)rows * M - off);
    CK(hipMemcpy(X + off, hx.data(), n * 8, hipMemcpyHostToDevice));
  }
  {  // plain copy ceiling
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_copy<<<2048, 256>>>((const double2*)X, (double2*)Y, (size_t)rows * M / 2);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) k_copy<<<2048, 256>>>((const double2*)X, (double2*)Y, (size_t)rows * M / 2);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-44s                       : %7.3f ms  (%6.0f GB/s r+w)\n", "float4-style copy (16B/lane)", ms, 16.0 * rows * M / ms / 1e6);
  }
  run(k_t<512, true, true, true, 0, 2>, 512, 256, X, Y, T, rows, "full (product structure)");
  run(k_t2<512, 1, 0>, 512, 256, X, Y, T, rows, "t-outer, 1 acc");
  run(k_t2<512, 2, 0>, 512, 256, X, Y, T, rows, "t-outer, 2 acc");
  run(k_t2<512, 4, 0>, 512, 256, X, Y, T, rows, "t-outer, 4 acc");
  run(k_t2<512, 2, 1>, 512, 256, X, Y, T, rows, "t-outer, 2 acc, staggered halves");
  run(k_t2<768, 1, 0>, 768, 256, X, Y, T, rows, "t-outer, 1 acc, 768 thr");
  run(k_t2<768, 2, 0>, 768, 256, X, Y, T, rows, "t-outer, 2 acc, 768 thr");
  run(k_t2<1024, 1, 0>, 1024, 256, X, Y, T, rows, "t-outer, 1 acc, 1024 thr");
  run(k_t2<1024, 2, 0>, 1024, 256, X, Y, T, rows, "t-outer, 2 acc, 1024 thr");
  run(k_t<512, true, true, false, 0, 2>, 512, 256, X, Y, T, rows, "no store");
  run(k_t<512, true, false, true, 0, 2>, 512, 256, X, Y, T, rows, "no mfma (load+store 8B)");
  run(k_t<512, true, false, true, 1, 2>, 512, 256, X, Y, T, rows, "no mfma (load+store 16B paired)");
  run(k_t<512, false, true, false, 0, 2>, 512, 256, X, Y, T, rows, "mfma only (LDS B reads)");
  run(k_t<512, true, true, true, 1, 2>, 512, 256, X, Y, T, rows, "full, 16B paired stores");
  run(k_t<512, true, true, true, 1, 1>, 512, 256, X, Y, T, rows, "full, 16B stores, sched barrier every step");
  run(k_t<512, true, true, true, 1, 0>, 512, 256, X, Y, T, rows, "full, 16B stores, no sched barrier");
  run(k_t<256, true, true, true, 1, 2>, 256, 256, X, Y, T, rows, "full, 16B stores, 256 thr (1 wave/SIMD)");
  run(k_t<256, true, true, true, 1, 0>, 256, 256, X, Y, T, rows, "full, 16B, 256 thr, no sched barrier");
  run(k_t<256, false, true, false, 0, 0>, 256, 256, X, Y, T, rows, "mfma only, 256 thr, no sched barrier");
  return 0;
}
