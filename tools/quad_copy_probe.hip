// FETCH_SIZE calibration for the QUAD layout of k_sweep_gc: a pure copy that reads rows exactly as that
// kernel does -- a wave instruction covers 16 rows (a column block's 16 columns of one slab, or 16 slabs), each row contributing one
// contiguous 64-byte segment (4 lanes x 16 B) -- and writes them back the same way.  Run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... ; rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
// and compare the counters with the bytes moved: MI355X_MICROARCH.md says FETCH_SIZE reports 1/2 of the bytes of wide coalesced
// streaming reads; this tells whether that also holds for 16 x 64-B segments per instruction.
//   hipcc --offload-arch=gfx950 -O3 -o tools/quad_copy_probe tools/quad_copy_probe.hip ; tools/quad_copy_probe [rows] [M] [ncol]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// quad copy: block = 16 consecutive columns x n_lead slabs; wave w of 4 takes columns 4w..4w+3, its 16 quads = 4 columns x 4 slabs
template <int NC>
__global__ __launch_bounds__(256) void k_quad_copy(const double* __restrict__ in, double* __restrict__ out, long ncol, long n_lead, int M) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 3, r = lane >> 2;
  const long col = (long)blockIdx.x * 16 + 4 * wave + (r & 3);
  if (col >= ncol) return;
  for (long lead0 = 0; lead0 < n_lead; lead0 += 4) {
    const long lead = lead0 + (r >> 2);
    if (lead >= n_lead) continue;
    const long row = lead * ncol + col;
    double2 x[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int m0 = 8 * c + 2 * j;
      x[c] = (m0 < M) ? *reinterpret_cast<const double2*>(in + row * M + m0) : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int m0 = 8 * c + 2 * j;
      if (m0 < M) *reinterpret_cast<double2*>(out + row * M + m0) = x[c];
    }
  }
}
// wide streaming copy for comparison: consecutive lanes, consecutive 16 B
__global__ __launch_bounds__(256) void k_stream_copy(const double2* __restrict__ in, double2* __restrict__ out, long n2) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}

int main(int argc, char** argv) {
  const long ncol = argc > 3 ? atol(argv[3]) : 361L * 720;
  const int M = argc > 2 ? atoi(argv[2]) : 80;
  const long n_lead = argc > 1 ? atol(argv[1]) : 148;
  const long rows = ncol * n_lead;
  const size_t bytes = (size_t)rows * M * 8;
  double *a, *b;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    if (M <= 80) hipLaunchKernelGGL((k_quad_copy<10>), dim3((unsigned)((ncol + 15) / 16)), dim3(256), 0, 0, a, b, ncol, n_lead, M);
    else hipLaunchKernelGGL((k_quad_copy<13>), dim3((unsigned)((ncol + 15) / 16)), dim3(256), 0, 0, a, b, ncol, n_lead, M);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_quad_copy   : %.3f ms, %.1f GB read + %.1f GB written -> %.2f TB/s\n", ms, bytes / 1e9, bytes / 1e9, 2.0 * bytes / ms / 1e9);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_stream_copy, dim3(256 * 16), dim3(256), 0, 0, reinterpret_cast<const double2*>(a), reinterpret_cast<double2*>(b), (long)(bytes / 16));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_stream_copy : %.3f ms -> %.2f TB/s\n", ms, 2.0 * bytes / ms / 1e9);
  }
  return 0;
}
