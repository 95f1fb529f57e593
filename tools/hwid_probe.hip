// Diagnostic: which SIMD does each wave of a 512-thread workgroup run on?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  extern __shared__ double lds[];
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
  if (threadIdx.x == 9999) lds[0] = 1;
}
int main() {
  unsigned* out; hipMalloc(&out, 4 * 16 * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipLaunchKernelGGL(k, dim3(4), dim3(512), 150 * 1024, 0, out);
  hipDeviceSynchronize();
  unsigned h[64]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 4; ++b) {
    printf("block %d:", b);
    for (int w = 0; w < 8; ++w) {
      unsigned id = h[b * 16 + w];
      printf("  w%d: simd %u cu %u wave_slot %u |", w, (id >> 4) & 3, (id >> 8) & 15, id & 15);
    }
    printf("\n");
  }
  return 0;
}
