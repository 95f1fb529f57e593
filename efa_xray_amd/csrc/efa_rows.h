// Register-resident row helpers shared by the sweep, diag and pipeline kernels.
//
// "L lanes per row" layout (L = 4, 8 or 16): L consecutive lanes own one row; lane j of
// the group holds members {2L*c+2j, 2L*c+2j+1 : c = 0..NC-1}.  Padding slots (member
// index >= M) always hold exactly 0.
#pragma once
#include <hip/hip_runtime.h>

namespace efa {

// ---- group reductions ------------------------------------------------------
#define EFA_DPP_STEP(v, ctrl)                                                          \
  do {                                                                                 \
    int _lo = __double2loint(v), _hi = __double2hiint(v);                              \
    _lo = __builtin_amdgcn_mov_dpp(_lo, (ctrl), 0xF, 0xF, true);                       \
    _hi = __builtin_amdgcn_mov_dpp(_hi, (ctrl), 0xF, 0xF, true);                       \
    v = v + __hiloint2double(_hi, _lo);                                                \
  } while (0)

template <int L>
__device__ __forceinline__ double group_sum(double v) {
  EFA_DPP_STEP(v, 0xB1);  // quad_perm [1,0,3,2]
  EFA_DPP_STEP(v, 0x4E);  // quad_perm [2,3,0,1]
  if (L >= 8) EFA_DPP_STEP(v, 0x141);   // row_half_mirror: quads 0<->1, 2<->3
  if (L >= 16) EFA_DPP_STEP(v, 0x140);  // row_mirror: the two halves of the 16-lane row
  return v;
}

// ---- row <-> registers -------------------------------------------------------
template <int L, int NC, bool VEC>
__device__ __forceinline__ void load_row(const double* __restrict__ p, int M, int j, double (&x)[2 * NC]) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int m0 = 2 * L * c + 2 * j;
    if (VEC) {
      if (m0 < M) {
        const double2 v = *reinterpret_cast<const double2*>(p + m0);
        x[2 * c] = v.x;
        x[2 * c + 1] = v.y;
      } else {
        x[2 * c] = 0.0;
        x[2 * c + 1] = 0.0;
      }
    } else {
      x[2 * c] = (m0 < M) ? p[m0] : 0.0;
      x[2 * c + 1] = (m0 + 1 < M) ? p[m0 + 1] : 0.0;
    }
  }
}

template <int L, int NC, bool VEC>
__device__ __forceinline__ void store_row(double* __restrict__ p, int M, int j, const double (&x)[2 * NC]) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int m0 = 2 * L * c + 2 * j;
    if (VEC) {
      if (m0 < M) *reinterpret_cast<double2*>(p + m0) = make_double2(x[2 * c], x[2 * c + 1]);
    } else {
      if (m0 < M) p[m0] = x[2 * c];
      if (m0 + 1 < M) p[m0 + 1] = x[2 * c + 1];
    }
  }
}

template <int L, int NC>
__device__ __forceinline__ void lds_read_row(const double* __restrict__ ys, int j, double (&y)[2 * NC]) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const double2 v = *reinterpret_cast<const double2*>(ys + 2 * L * c + 2 * j);
    y[2 * c] = v.x;
    y[2 * c + 1] = v.y;
  }
}

// dot(x, ye) over the lane's slots with up to four independent FMA chains, then the group total
template <int L, int NC>
__device__ __forceinline__ double group_dot(const double (&x)[2 * NC], const double (&y)[2 * NC]) {
  double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int c = 0; c < 2 * NC; ++c) s[c & 3] = __builtin_fma(x[c], y[c], s[c & 3]);
  return group_sum<L>((s[0] + s[1]) + (s[2] + s[3]));
}

// sum over the lane's valid slots of (x - mean)^2, then the group total
template <int L, int NC>
__device__ __forceinline__ double group_centered_sumsq(const double (&x)[2 * NC], double mean, int M, int j) {
  double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int m0 = 2 * L * c + 2 * j;
    const double d0 = (m0 < M) ? (x[2 * c] - mean) : 0.0;
    const double d1 = (m0 + 1 < M) ? (x[2 * c + 1] - mean) : 0.0;
    s[(2 * c) & 3] = __builtin_fma(d0, d0, s[(2 * c) & 3]);
    s[(2 * c + 1) & 3] = __builtin_fma(d1, d1, s[(2 * c + 1) & 3]);
  }
  return group_sum<L>((s[0] + s[1]) + (s[2] + s[3]));
}

// sum over ALL of the lane's slots of (x - mean)^2 (no masking), then the group total
template <int L, int NC>
__device__ __forceinline__ double group_sumsq_about(const double (&x)[2 * NC], double mean) {
  double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int c = 0; c < 2 * NC; ++c) {
    const double d = x[c] - mean;
    s[c & 3] = __builtin_fma(d, d, s[c & 3]);
  }
  return group_sum<L>((s[0] + s[1]) + (s[2] + s[3]));
}

template <int L, int NC>
__device__ __forceinline__ double group_rowsum(const double (&x)[2 * NC]) {
  double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int c = 0; c < 2 * NC; ++c) s[c & 3] += x[c];
  return group_sum<L>((s[0] + s[1]) + (s[2] + s[3]));
}


}  // namespace efa
