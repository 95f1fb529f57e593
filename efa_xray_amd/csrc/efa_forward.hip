// f1: the reference's default forward operator (Observation.estimate -> EnsembleState.interpolate,
// observation.py:40-50, ensemble.py:152-239) for all P point observations at once, on the device:
//   k_grid_trig     sin(radians(lat)), cos(radians(lon)) of every grid point, once per call
//   k_nearest4      one workgroup per ob: the 4 grid points nearest in the reference's sin/cos
//                   pseudo-distance (ensemble.py:160-165) -- replaces a full argsort of ny*nx values per ob
//   k_interp_weights one lane per ob: inverse great-circle-distance weights with the < 1 km rule
//                   (ensemble.py:178-200) and the linear time weights as coded (ensemble.py:201-224),
//                   expanded to a stencil of up to 8 (global state row, weight) entries
//   k_forward_cols  applies a stencil to a column shard of the state (the all-reduce payload)
// Pinned to the reference since round 3: its nearest_points / interpolate / estimate run verbatim in the
// build container on a duck-typed state and their outputs are fixtures tests/golden/G9, G10, G11
// (tests/test_gpu_api.py).  Ties in the pseudo-distance (the reference's unstable argsort leaves their
// order open) go to the lower index.
#include "efa_device.h"
#include "efa_internal.h"

namespace efa {
namespace {

constexpr int kNT = 256;

__global__ __launch_bounds__(kNT) void k_grid_trig(long n, const double* __restrict__ glat, const double* __restrict__ glon,
                                                   double* __restrict__ sl, double* __restrict__ cl) {
  for (long i = (long)blockIdx.x * kNT + threadIdx.x; i < n; i += (long)gridDim.x * kNT) {
    sl[i] = sin(radians(glat[i]));
    cl[i] = cos(radians(glon[i]));
  }
}

struct Cand {
  double d;
  long i;
};
__device__ __forceinline__ bool closer(double d1, long i1, double d2, long i2) {  // (d, i) lexicographic; NaN last
  return (d1 < d2) || (d1 == d2 && i1 < i2) || (d2 != d2 && d1 == d1);
}

// nearest[k*4 + r] = flat grid index of the r-th nearest point of ob k (r = 0 nearest)
__global__ __launch_bounds__(kNT) void k_nearest4(long n, const double* __restrict__ sl, const double* __restrict__ cl,
                                                  const double* __restrict__ ob_lat, const double* __restrict__ ob_lon,
                                                  long* __restrict__ nearest) {
  __shared__ double red_d[kNT / 64];
  __shared__ long red_i[kNT / 64];
  __shared__ int red_t[kNT / 64];
  __shared__ int win_thread;
  const long k = blockIdx.x;
  const int tid = threadIdx.x;
  const double s0 = sin(radians(ob_lat[k])), c0 = cos(radians(ob_lon[k]));
  const double inf = __builtin_huge_val();
  Cand best[4] = {{inf, -1}, {inf, -1}, {inf, -1}, {inf, -1}};
  for (long i = tid; i < n; i += kNT) {
    const double d = hypot(sl[i] - s0, cl[i] - c0);  // ensemble.py:160-163
    if (closer(d, i, best[3].d, best[3].i)) {        // insertion into the sorted four
      best[3] = {d, i};
#pragma unroll
      for (int r = 3; r > 0; --r) {
        if (closer(best[r].d, best[r].i, best[r - 1].d, best[r - 1].i)) {
          const Cand t = best[r];
          best[r] = best[r - 1];
          best[r - 1] = t;
        }
      }
    }
  }
  int head = 0;  // this thread's candidates best[head..3] are still in play
  for (int r = 0; r < 4; ++r) {
    double d = inf;
    long i = -1;
    // head is thread-dependent: select without dynamic register indexing
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q == head) {
        d = best[q].d;
        i = best[q].i;
      }
    if (i < 0) d = __builtin_nan("");  // exhausted: sorts last
    int t = tid;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double d2 = __shfl_xor(d, off, 64);
      const long i2 = __shfl_xor(i, off, 64);
      const int t2 = __shfl_xor(t, off, 64);
      if (closer(d2, i2, d, i) || (i < 0 && i2 >= 0)) {
        d = d2;
        i = i2;
        t = t2;
      }
    }
    if ((tid & 63) == 0) {
      red_d[tid >> 6] = d;
      red_i[tid >> 6] = i;
      red_t[tid >> 6] = t;
    }
    __syncthreads();
    if (tid == 0) {
      double bd = red_d[0];
      long bi = red_i[0];
      int bt = red_t[0];
      for (int w = 1; w < kNT / 64; ++w)
        if (closer(red_d[w], red_i[w], bd, bi) || (bi < 0 && red_i[w] >= 0)) {
          bd = red_d[w];
          bi = red_i[w];
          bt = red_t[w];
        }
      nearest[k * 4 + r] = bi;
      win_thread = bt;
    }
    __syncthreads();
    if (tid == win_thread) ++head;
    __syncthreads();
  }
}

// One lane per ob.  Stencil entry e = 4*t + j: time slot t (0: earlier valid time, 1: later / exact), point j.
__global__ __launch_bounds__(kNT) void k_interp_weights(long P, int nvar, int nt, int ny, int nx, int latlon_1d, long n_grid,
                                                        const double* __restrict__ glat, const double* __restrict__ glon,
                                                        const double* __restrict__ valids, const int* __restrict__ ob_var,
                                                        const double* __restrict__ ob_time, const double* __restrict__ ob_lat,
                                                        const double* __restrict__ ob_lon, const long* __restrict__ nearest,
                                                        long* __restrict__ sten_idx, double* __restrict__ sten_wts,
                                                        unsigned char* __restrict__ status) {
  const long k = (long)blockIdx.x * kNT + threadIdx.x;
  if (k >= P) return;
  for (int e = 0; e < 8; ++e) {
    sten_idx[k * 8 + e] = -1;
    sten_wts[k * 8 + e] = 0.0;
  }
  const double lat = ob_lat[k], lon = ob_lon[k], t = ob_time[k];
  const int iv = ob_var[k];
  if (iv < 0 || iv >= nvar) {
    status[k] = 3;
    return;
  }
  // ---- space (ensemble.py:178-200) ----
  long col[4];
  double dist[4], sw[4];
  int npt = 0;
  for (int j = 0; j < 4; ++j) {
    const long g = nearest[k * 4 + j];
    if (g < 0) break;  // fewer than 4 grid points
    dist[npt] = haversine_km(glat[g], glon[g], lat, lon);  // haversine((grid lat, lon), (ob lat, lon)): ensemble.py:181-184
    if (latlon_1d) {
      if (g >= ny || g >= nx) {  // closey = closex = closen (ensemble.py:186-188): the reference would raise IndexError
        status[k] = 2;
        return;
      }
      col[npt] = g * nx + g;
    } else {
      col[npt] = g;
    }
    ++npt;
  }
  if (npt == 0) {
    status[k] = 2;
    return;
  }
  int nclose = 0, amin = 0;
  for (int j = 0; j < npt; ++j) {
    if (dist[j] < 1.0) ++nclose;
    if (dist[j] < dist[amin]) amin = j;  // first minimum, as np.argmin
  }
  if (nclose > 0) {
    for (int j = 0; j < npt; ++j) sw[j] = (j == amin) ? 1.0 : 0.0;  // exact match within 1 km (ensemble.py:193-196)
  } else {
    double tot = 0.0;
    for (int j = 0; j < npt; ++j) {
      sw[j] = 1.0 / dist[j];
      tot = (j == 0) ? sw[j] : tot + sw[j];  // np.sum of <= 4 values: left to right
    }
    for (int j = 0; j < npt; ++j) sw[j] = sw[j] / tot;
  }
  // ---- time (ensemble.py:201-224) ----
  if (t < valids[0] || t > valids[nt - 1] || !(t == t)) {
    status[k] = 1;  // "Interpolation is outside of time range in state!"
    return;
  }
  int last = 0;
  while (last < nt - 1 && !(valids[last] >= t)) ++last;  // (valids >= t).argmax()
  int tslot[2] = {-1, -1};
  double tw[2] = {0.0, 0.0};
  if (valids[last] == t) {
    tslot[1] = last;
    tw[1] = 1.0;
  } else {
    const double tot = fabs(valids[last] - valids[last - 1]);
    const double ths = fabs(t - valids[last]);
    tslot[0] = last - 1;
    tslot[1] = last;
    tw[1] = ths / tot;        // timeweights[lastdex]   = thissec / totsec   (as coded)
    tw[0] = 1.0 - ths / tot;  // timeweights[lastdex-1] = 1 - thissec / totsec
  }
  const long ncol = (long)ny * nx;
  for (int s = 0; s < 2; ++s) {
    if (tslot[s] < 0 || tw[s] == 0.0) continue;
    for (int j = 0; j < npt; ++j) {
      sten_idx[k * 8 + 4 * s + j] = ((long)iv * nt + tslot[s]) * ncol + col[j];
      sten_wts[k * 8 + 4 * s + j] = tw[s] * sw[j];
    }
  }
  status[k] = 0;
}

// HX[k,:] = sum_e w[k,e] * X[local(row[k,e]),:] over the entries whose column this shard owns
__global__ __launch_bounds__(kNT) void k_forward_cols(long ncol, long col_lo, long col_hi, long n_lead, int M,
                                                      const double* __restrict__ X, long P, int npt,
                                                      const long* __restrict__ idx, const double* __restrict__ wts,
                                                      double* __restrict__ HX) {
  const size_t total = (size_t)P * M;
  const long ncl = col_hi - col_lo;
  for (size_t i = (size_t)blockIdx.x * kNT + threadIdx.x; i < total; i += (size_t)gridDim.x * kNT) {
    const long k = (long)(i / M);
    const int m = (int)(i - (size_t)k * M);
    double acc = 0.0;
    bool first = true;
    for (int j = 0; j < npt; ++j) {
      const double w = wts[(size_t)k * npt + j];
      const long g = idx[(size_t)k * npt + j];
      if (w == 0.0 || g < 0) continue;
      const long lead = g / ncol, col = g - lead * ncol;
      if (col < col_lo || col >= col_hi || lead >= n_lead) continue;
      const double t = w * X[(size_t)(lead * ncl + (col - col_lo)) * M + m];
      acc = first ? t : acc + t;
      first = false;
    }
    HX[i] = acc;
  }
}

inline unsigned grid_for(size_t items, int per_block) {
  size_t g = (items + per_block - 1) / per_block;
  if (g > 256u * 8u) g = 256u * 8u;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

hipError_t launch_interp_stencils(const InterpArgs& a, hipStream_t s) {
  if (a.P <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_grid_trig, dim3(grid_for((size_t)a.n_grid, kNT)), dim3(kNT), 0, s, a.n_grid, a.glat, a.glon, a.sl, a.cl);
  hipLaunchKernelGGL(k_nearest4, dim3((unsigned)a.P), dim3(kNT), 0, s, a.n_grid, a.sl, a.cl, a.ob_lat, a.ob_lon, a.nearest);
  hipLaunchKernelGGL(k_interp_weights, dim3((unsigned)((a.P + kNT - 1) / kNT)), dim3(kNT), 0, s, a.P, a.nvar, a.nt, a.ny, a.nx,
                     a.latlon_1d, a.n_grid, a.glat, a.glon, a.valids, a.ob_var, a.ob_time, a.ob_lat, a.ob_lon, a.nearest,
                     a.sten_idx, a.sten_wts, a.status);
  return hipGetLastError();
}

hipError_t launch_forward_cols(long ncol, long col_lo, long col_hi, long n_lead, int M, const double* X, long P, int npt,
                               const long* idx, const double* wts, double* HX, hipStream_t s) {
  if (P <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_forward_cols, dim3(grid_for((size_t)P * M, kNT)), dim3(kNT), 0, s, ncol, col_lo, col_hi, n_lead, M, X, P,
                     npt, idx, wts, HX);
  return hipGetLastError();
}

}  // namespace efa
