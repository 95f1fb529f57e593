// Probe the operand/result lane layout of v_mfma_f64_16x16x4_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A, const double* B, double* D) {  // A[16][4], B[4][16] row-major; D out [64][4]
  int l = threadIdx.x;
  // hypothesis: A lane l holds A[l%16][l/16]; B lane l holds B[l/16][l%16]
  double a = A[(l % 16) * 4 + (l / 16)];
  double b = B[(l / 16) * 16 + (l % 16)];
  v4f64 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) D[l * 4 + v] = c[v];
}
int main() {
  double hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = 1 + i + 100.0 * k;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = 1 + 0.01 * j + 7.0 * k * k;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[i*4+k]*hB[k*16+j]; ref[i*16+j] = s; }
  double *dA, *dB, *dD;
  (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, sizeof hD);
  (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dD);
  (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  // find for each (lane, v) which (i, j) of ref matches
  int ok = 1;
  for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
    int fi = -1, fj = -1, n = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (fabs(ref[i*16+j] - hD[l*4+v]) < 1e-9) { fi = i; fj = j; ++n; }
    if (l < 20 || l % 16 == 0) printf("lane %2d v %d -> D[%d][%d] (matches %d)\n", l, v, fi, fj, n);
    if (!(n == 1 && fi == 4 * (l / 16) + v && fj == l % 16)) ok = 0;
  }
  printf("hypothesis i=4*(lane/16)+v, j=lane%%16: %s\n", ok ? "CONFIRMED" : "WRONG");
  return 0;
}
