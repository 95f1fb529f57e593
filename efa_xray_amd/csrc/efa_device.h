// Device-side helpers shared by the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace efa {

constexpr double kEarthRadiusKm = 6371.0;  // ensemble.py:259, observation.py:138
constexpr double kPiOver180 = 3.14159265358979323846 / 180.0;

__device__ __forceinline__ double radians(double deg) { return deg * kPiOver180; }

// Great-circle km, operand order of Observation-side `haversine(loc1, loc2)`
// (observation.py:135-146): loc1 = the observation being assimilated.
__device__ __forceinline__ double haversine_km(double lat1, double lon1, double lat2, double lon2) {
  const double p1 = radians(lat1);
  const double p2 = radians(lat2);
  const double dp = p2 - p1;
  const double dl = radians(lon2 - lon1);
  const double s1 = sin(dp / 2);
  const double s2 = sin(dl / 2);
  const double a = s1 * s1 + cos(p1) * cos(p2) * (s2 * s2);
  const double c = 2 * atan2(sqrt(a), sqrt(1 - a));
  return kEarthRadiusKm * c;
}

// Great-circle km from a grid point to the observation, operand order of
// EnsembleState.distance_to_point (ensemble.py:254-267).
__device__ __forceinline__ double distance_to_point_km(double grid_lat, double grid_lon,
                                                       double ob_lat, double ob_lon) {
  const double plat = radians(ob_lat);
  const double plon = radians(ob_lon);
  const double glat = radians(grid_lat);
  const double dlat = plat - glat;
  const double dlon = plon - radians(grid_lon);
  const double s1 = sin(dlat / 2);
  const double s2 = sin(dlon / 2);
  const double a = s1 * s1 + cos(plat) * cos(glat) * (s2 * s2);
  const double c = 2 * atan2(sqrt(a), sqrt(1.0 - a));
  return kEarthRadiusKm * c;
}

// Gaspari-Cohn taper (observation.py:117-130), same Horner order.
__device__ __forceinline__ double gaspari_cohn(double dist_km, double halfwidth_km) {
  const double r = dist_km / fabs(halfwidth_km);
  if (r <= 1.0) {
    return ((((-0.25 * r + 0.5) * r + 0.625) * r - 5.0 / 3.0) * (r * r) + 1.0);
  } else if (r < 2.0) {
    return (((((r / 12.0 - 0.5) * r + 0.625) * r + 5.0 / 3.0) * r - 5.0) * r + 4.0 -
            2.0 / (3.0 * r));
  }
  return 0.0;  // r >= 2 or NaN
}

// Sum over the 4 lanes of a quad (lanes 4q..4q+3); every lane gets the total.
// DPP quad_perm butterflies on the two 32-bit halves of the double.
__device__ __forceinline__ double quad_sum(double v) {
  {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
    v = v + __hiloint2double(hi, lo);
  }
  {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true);
    v = v + __hiloint2double(hi, lo);
  }
  return v;
}

// Sum over all 64 lanes of the wave; every lane gets the total.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Counter-based uniform/normal generator (splitmix64 finaliser) used for the
// bench's synthetic ensemble: value depends only on (seed, row, member).
__device__ __host__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ double u01(uint64_t h) {  // (0,1)
  return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double normal_from(uint64_t key) {
  const double u1 = u01(mix64(key));
  const double u2 = u01(mix64(key ^ 0xD1B54A32D192ED03ull));
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

}  // namespace efa
