// Streaming helper kernels around the hot loop: perturbation formation (a3),
// posterior rebuild (a15), per-batch Gaspari-Cohn taper table (a10), the linear
// forward-operator stencil (f1) and the bench's synthetic-ensemble generator.
#include "efa_device.h"
#include "efa_internal.h"

namespace efa {
namespace {

constexpr int kThreads = 256;

// One wave per row: lane l handles members l, l+64, ...  Rows are contiguous
// so a wave reads/writes M*8 contiguous bytes per row (coalesced).
// xm = mean over members (assimilation.py:146), Xp = scale*(X - xm) (:147, :67).
__global__ __launch_bounds__(kThreads) void k_form_perts(long rows, int M, const double* __restrict__ X,
                                                         double scale, double* __restrict__ xm,
                                                         double* __restrict__ Xp) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (kThreads / 64);
  for (long row = wave; row < rows; row += nwaves) {
    const double* p = X + (size_t)row * M;
    double v[4];
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = lane + 64 * j;
      v[j] = (m < M) ? p[m] : 0.0;
      s += v[j];
    }
    const double mean = wave_sum(s) / (double)M;
    double* o = Xp + (size_t)row * M;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = lane + 64 * j;
      if (m < M) o[m] = (scale == 1.0) ? (v[j] - mean) : (v[j] - mean) * scale;
    }
    if (lane == 0) xm[row] = mean;
  }
}

// post = xam[:,None] + Xap (assimilation.py:168); flat, 2 doubles per lane.
__global__ __launch_bounds__(kThreads) void k_posterior(long rows, int M, const double* __restrict__ xm,
                                                        const double* __restrict__ Xp,
                                                        double* __restrict__ post) {
  const size_t total = (size_t)rows * M;
  const size_t stride = (size_t)gridDim.x * kThreads;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += stride) {
    const size_t row = i / M;
    post[i] = xm[row] + Xp[i];
  }
}

// W[k][col] = GC(distance_to_point(grid[col], ob_k) / halfwidth_k)
// (observation.py:59-87 on an EnsembleState; the caller broadcasts it over
// var x time, ensrf.py:108-111 -- here by indexing col = row % ncol).
__global__ __launch_bounds__(kThreads) void k_taper_table(long ncol, int nb, const double* __restrict__ glat,
                                                          const double* __restrict__ glon,
                                                          const double* __restrict__ ob_lat,
                                                          const double* __restrict__ ob_lon,
                                                          const double* __restrict__ ob_hw,
                                                          double* __restrict__ W) {
  const size_t total = (size_t)ncol * nb;
  const size_t stride = (size_t)gridDim.x * kThreads;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += stride) {
    const int k = (int)(i / ncol);
    const long c = (long)(i - (size_t)k * ncol);
    const double d = distance_to_point_km(glat[c], glon[c], ob_lat[k], ob_lon[k]);
    W[i] = gaspari_cohn(d, ob_hw[k]);
  }
}

// HX[k,:] = sum_j w[k,j] * X[idx[k,j]-row_offset,:] over locally owned points.
__global__ __launch_bounds__(kThreads) void k_forward_stencil(long rows, long row_offset, int M,
                                                              const double* __restrict__ X, long P,
                                                              int npt, const int64_t* __restrict__ idx,
                                                              const double* __restrict__ wts,
                                                              double* __restrict__ HX) {
  const size_t total = (size_t)P * M;
  const size_t stride = (size_t)gridDim.x * kThreads;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += stride) {
    const long k = (long)(i / M);
    const int m = (int)(i - (size_t)k * M);
    double acc = 0.0;
    bool first = true;
    for (int j = 0; j < npt; ++j) {
      const double w = wts[(size_t)k * npt + j];
      const long g = idx[(size_t)k * npt + j] - row_offset;
      if (w == 0.0 || g < 0 || g >= rows) continue;
      const double t = w * X[(size_t)g * M + m];
      acc = first ? t : acc + t;
      first = false;
    }
    HX[i] = acc;
  }
}

__global__ __launch_bounds__(kThreads) void k_fill_synthetic(long rows, long row_offset, int M,
                                                             uint64_t seed, double sigma,
                                                             double* __restrict__ X) {
  const size_t total = (size_t)rows * M;
  const size_t stride = (size_t)gridDim.x * kThreads;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += stride) {
    const uint64_t row = (uint64_t)(i / M) + (uint64_t)row_offset;
    const uint64_t m = (uint64_t)(i % M);
    const double mu = normal_from(mix64(seed ^ 0xA5A5A5A5ull) + row * 0x100000001B3ull);
    const double z = normal_from(mix64(seed) + row * 1099511628211ull + m * 0x9E3779B97F4A7C15ull);
    X[i] = mu + sigma * z;
  }
}

__global__ void k_set_identity(int M, double* __restrict__ T, double* __restrict__ w) {
  const int total = M * M;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x)
    T[i] = (i / M == i % M) ? 1.0 : 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) w[i] = 0.0;
}

// Diagnostic occupier (option "debug_occupy_ms"): holds LDS on as many CUs as it has blocks, for a bounded time
__global__ void k_occupy(long ticks) {
  extern __shared__ double hold[];
  if (threadIdx.x == 0) hold[0] = 1.0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(32);
  if (hold[0] == 12345.0) hold[1] = 0.0;
}

inline unsigned grid_for(size_t work_items, int per_block) {
  size_t g = (work_items + per_block - 1) / per_block;
  const size_t cap = 256u * 8u;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

hipError_t launch_form_perts(long rows, int M, const double* X, double scale, double* xm, double* Xp,
                             hipStream_t s) {
  if (M < 1 || M > kMaxMembers) return hipErrorInvalidValue;
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_form_perts, dim3(grid_for((size_t)rows, kThreads / 64)), dim3(kThreads), 0, s, rows, M,
                     X, scale, xm, Xp);
  return hipGetLastError();
}

hipError_t launch_posterior(long rows, int M, const double* xm, const double* Xp, double* post,
                            hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_posterior, dim3(grid_for((size_t)rows * M, kThreads)), dim3(kThreads), 0, s, rows, M,
                     xm, Xp, post);
  return hipGetLastError();
}

hipError_t launch_taper_table(long ncol, int nb, const double* grid_lat, const double* grid_lon,
                              const double* ob_lat, const double* ob_lon, const double* ob_hw, double* W,
                              hipStream_t s) {
  if (ncol <= 0 || nb <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_taper_table, dim3(grid_for((size_t)ncol * nb, kThreads)), dim3(kThreads), 0, s, ncol,
                     nb, grid_lat, grid_lon, ob_lat, ob_lon, ob_hw, W);
  return hipGetLastError();
}

hipError_t launch_forward_stencil(long rows, long row_offset, int M, const double* X, long P, int npt,
                                  const int64_t* idx, const double* wts, double* HX, hipStream_t s) {
  if (P <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_forward_stencil, dim3(grid_for((size_t)P * M, kThreads)), dim3(kThreads), 0, s, rows,
                     row_offset, M, X, P, npt, idx, wts, HX);
  return hipGetLastError();
}

hipError_t launch_fill_synthetic(long rows, long row_offset, int M, uint64_t seed, double sigma, double* X,
                                 hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_fill_synthetic, dim3(grid_for((size_t)rows * M, kThreads)), dim3(kThreads), 0, s, rows,
                     row_offset, M, seed, sigma, X);
  return hipGetLastError();
}

hipError_t launch_occupy(int blocks, size_t lds_bytes, double ms, hipStream_t s) {
  if (blocks <= 0 || ms <= 0) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_occupy), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_occupy, dim3((unsigned)blocks), dim3(64), lds_bytes, s, (long)(ms * 1e5));
  return hipGetLastError();
}

// Everything one Phase-A call needs before its first launch, in ONE launch (five small operations cost ~10 us of launch
// latency each -- a fifth of configs[1]'s whole cycle): the caller's obs block into the working rows, the identity rows
// that come out as [T|w], the trajectory records filled with their sentinel, the status words cleared.
// (Round 3, late: the per-ob input pack also comes in HERE, read by the kernel straight from the mapped pinned buffer the host
//  filled -- 16 bytes per thread, each read once.  As a copy of its own it sat between two kernels on the copy engine, and every
//  kernel <-> copy switch leaves the stream idle for ~10 us: 15 us of copy + 20 us of gaps per cycle.)
__global__ void k_phase_a_prep(long P, int M, const double* __restrict__ Yp, const double* __restrict__ ym, double* __restrict__ Yw,
                               double* __restrict__ ymw, int carry_T, unsigned long long* __restrict__ traj, size_t traj_words,
                               unsigned long long sentinel, int* __restrict__ status, const uint4* __restrict__ pack_host,
                               uint4* __restrict__ pack_dev, size_t pack_n16) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  for (size_t i = tid; i < pack_n16; i += nth) pack_dev[i] = pack_host[i];
  const size_t n = (size_t)P * M;
  if ((n & 1) == 0 && ((reinterpret_cast<uintptr_t>(Yp) | reinterpret_cast<uintptr_t>(Yw)) & 15u) == 0) {
    const double2* src = reinterpret_cast<const double2*>(Yp);
    double2* dst = reinterpret_cast<double2*>(Yw);
    for (size_t i = tid; i < n / 2; i += nth) dst[i] = src[i];
  } else {
    for (size_t i = tid; i < n; i += nth) Yw[i] = Yp[i];
  }
  for (size_t i = tid; i < (size_t)P; i += nth) ymw[i] = ym[i];
  if (carry_T) {
    double* T = Yw + n;
    for (size_t i = tid; i < (size_t)M * M; i += nth) T[i] = (i / M == i % M) ? 1.0 : 0.0;
    for (size_t i = tid; i < (size_t)M; i += nth) ymw[P + i] = 0.0;
  }
  if (traj != nullptr)
    for (size_t i = tid; i < traj_words; i += nth) traj[i] = sentinel;
  if (status != nullptr && tid < 3) status[tid] = 0;
}

hipError_t launch_phase_a_prep(long P, int M, const double* Yp, const double* ym, double* Yw, double* ymw, int carry_T,
                               unsigned long long* traj, size_t traj_words, unsigned long long sentinel, int* status,
                               const void* pack_host, void* pack_dev, size_t pack_bytes, hipStream_t s) {
  size_t work = (size_t)P * M / 2 + 1;
  if (traj && traj_words > work) work = traj_words;
  hipLaunchKernelGGL(k_phase_a_prep, dim3(grid_for(work, 256 * 4)), dim3(256), 0, s, P, M, Yp, ym, Yw, ymw, carry_T, traj, traj_words,
                     sentinel, status, static_cast<const uint4*>(pack_host), static_cast<uint4*>(pack_dev), pack_bytes / 16);
  return hipGetLastError();
}

// A persistent Phase-A launch's status words and the diagnostics it wrote, stored by a kernel straight into mapped pinned host
// memory (coalesced 16-byte stores, each byte once): behind it the next kernel starts at once, where two device-to-host copies
// cost 15 us plus a ~10 us engine switch on either side.  The host reads them after waiting for an event recorded behind this kernel.
__global__ void k_results_to_host(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, const int* __restrict__ st_src,
                                  int* __restrict__ st_dst) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  for (size_t i = tid; i < n16; i += nth) dst[i] = src[i];
  if (tid < 3) st_dst[tid] = st_src[tid];
}
hipError_t launch_results_to_host(const void* src_dev, void* dst_host, size_t bytes, const int* st_dev, int* st_host, hipStream_t s) {
  const size_t n16 = (bytes + 15) / 16;
  hipLaunchKernelGGL(k_results_to_host, dim3(grid_for(n16 ? n16 : 1, 256)), dim3(256), 0, s, static_cast<const uint4*>(src_dev),
                     static_cast<uint4*>(dst_host), n16, st_dev, st_host);
  return hipGetLastError();
}

hipError_t launch_set_identity(int M, double* T, double* w, hipStream_t s) {
  hipLaunchKernelGGL(k_set_identity, dim3(16), dim3(256), 0, s, M, T, w);
  return hipGetLastError();
}

}  // namespace efa
