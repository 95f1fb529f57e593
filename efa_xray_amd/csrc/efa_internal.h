// Internal interfaces between the C-ABI layer (efa_capi.hip) and the kernel
// translation units.  Nothing here is exported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace efa {

// Largest ensemble the register-resident kernels are instantiated for.
constexpr int kMaxMembers = 256;
// Rows (observations) handled by one obs-space diag workgroup == max obs_batch.
constexpr int kMaxBatch = 64;

// Per-observation coefficients recorded by Phase A (4 doubles per ob):
//   [0] innov   = ob.value - mye                     (ensrf.py:85)
//   [1] rden    = 1 / (varye + ob.error)             (ensrf.py:91,119)
//   [2] beta    = 1/(1+sqrt(err/(varye+err)))        (ensrf.py:135)
//   [3] active  = 1.0 if ob.assimilate_this else 0.0 (ensrf.py:74)
constexpr int kCoefStride = 4;

enum TaperMode : int {
  kTaperNone = 0,   // loc in (None, False)
  kTaperTable = 1,  // state rows: W[k][col] precomputed per batch (2-D taper, ensrf.py:108-111)
  kTaperObs = 2     // obs rows: haversine between observations, in-kernel (ensrf.py:113)
};

struct SweepArgs {
  const double* Xin;   // [nrows][M] perturbation rows (may alias Xout)
  const double* xin;   // [nrows]   means
  double* Xout;
  double* xout;
  long nrows;
  int M;
  const double* Ye;    // [nb][ye_stride] recorded obs-space perturbations of the batch
  long ye_stride;      // doubles between consecutive ye rows (>= M)
  const double* coef;  // [nb][kCoefStride]
  int nb;
  int taper_mode;
  const double* W;     // kTaperTable: [nb][ncol]
  long ncol;
  const double* row_lat;  // kTaperObs: lat/lon of the ob each swept row belongs to [nrows]
  const double* row_lon;
  const double* ob_lat;   // kTaperObs: lat/lon/halfwidth of the batch obs [nb]
  const double* ob_lon;
  const double* ob_hw;
  long skip_lo, skip_hi;  // rows in [skip_lo, skip_hi) are not touched
  long taper_rows;        // kTaperObs applies to rows < taper_rows (extra rows: weight 1)
};

struct DiagArgs {
  double* Yp;  // [P(+extra)][M] obs block perturbations (in/out)
  double* ym;  // [P(+extra)]    obs block means (in/out)
  int M;
  long b0;  // first ob of the batch
  int nb;   // obs in the batch (<= kMaxBatch)
  const double* ob_value;  // device [P]
  const double* ob_error;
  const uint8_t* ob_assim;
  int loc_mode;
  const double* ob_lat;
  const double* ob_lon;
  const double* ob_hw;
  double* Ye_rec;  // [P][M] out: ye of ob k at the moment it is assimilated
  double* coef;    // [P][kCoefStride] out
  double* prior_mean;  // device [P] out
  double* prior_var;
  double* post_mean;
  double* post_var;
  uint8_t* assimilated;
};

// ---- persistent Phase-A pipeline (efa_pipeline.hip) --------------------------------
// Trajectory record of ob k in global memory: kTrajPad doubles of ye (zero padded) followed
// by 8 scalars.  Every 8-byte element is written once with an agent-scope store and is
// pre-filled with kTrajSentinel, so a reader validates each element on its own: no flag,
// no ordering between the elements (MI355X_MICROARCH.md, "R2 granule").
constexpr unsigned long long kTrajSentinel = 0x7FF8DEADBEEF0001ull;  // a NaN payload no arithmetic produces
constexpr int kTrajScalars = 8;  // mye, mean(ye), innov, rden, beta, active, prior_var, (unused)
constexpr int kPipeLanes = 4;    // lanes per row in the pipeline kernel (== its compute waves per workgroup)
inline int traj_pad(int M) { return 2 * kPipeLanes * ((M + 2 * kPipeLanes - 1) / (2 * kPipeLanes)); }
inline long traj_stride(int M) { return traj_pad(M) + kTrajScalars; }
constexpr int kPipeRowsPerWG = 64;
constexpr int kPipeMaxWGs = 256;  // one workgroup per CU (320 threads k_pipe, 512 threads k_pipe_gram): all co-resident

struct PipeArgs {
  double* Yp;   // [R][M] obs block (+ extra identity rows), in/out
  double* ym;   // [R]
  long R;       // rows swept (P + extra)
  long P;       // observations
  int M;
  const double* ob_value;  // device [P]
  const double* ob_error;
  const uint8_t* ob_assim;
  const double* ob_errsq;  // device [P][4]: {error, sqrt(error), assimilate (1.0 / 0.0), 0} (the band leader reads them with wave-uniform loads)
  int loc_mode;
  const double* tw;  // GC: dense obs-obs taper [P][R] (row k = ob k against every row); else null
  unsigned long long* traj;  // [P][traj_stride(M)] sentinel-filled
  double* coef;         // [P][kCoefStride] out
  double* prior_mean;   // [P] out
  double* prior_var;
  double* post_mean;
  double* post_var;
  uint8_t* assimilated;
  int* status;          // [3]: [0] abort flag (in-kernel), [1] 1 = given up, [2] 1 = because of the Gram cancellation guard
  long spin_limit;      // bound of the in-kernel polls (count)
  long spin_ticks;      // ... and in wall time: s_memrealtime ticks (100 MHz) since the kernel started; 0 = none
  int cu_count;         // compute units of the device: the launch is refused unless the grid is co-resident
  unsigned long long* dbg;  // diagnostic (debug & 4): [P][8] cycle stamps of the leader chain, else null
  int debug;  // diagnostic bits (single-workgroup timing runs only): 1 no global publication, 2 no prefetch
};

hipError_t launch_pipeline(const PipeArgs& a, hipStream_t s);
bool pipeline_supported(int M, long R);
hipError_t launch_pipeline_gram(const PipeArgs& a, hipStream_t s);  // efa_pipeline_gram.hip
bool pipeline_gram_supported(int M, long R, int loc_mode);
hipError_t launch_pipeline_band(const PipeArgs& a, hipStream_t s);  // efa_pipeline_band.hip (unlocalised cycles)
bool pipeline_band_supported(int M, long R, int loc_mode);
long band_traj_stride(int M);  // doubles per trajectory record as k_pipe_band lays them out (its rows are padded to its own lane layout)
hipError_t launch_fill_u64(unsigned long long* p, size_t n, unsigned long long v, hipStream_t s);
hipError_t launch_obs_taper_matrix(long P, long R, const double* ob_lat, const double* ob_lon, const double* ob_hw,
                                   double* trig_scratch /* [P][6] */, double* tw, hipStream_t s);

// ---- one-pass localised sweep (efa_gcsweep.hip) ---------------------------------------
struct GcSweepArgs {
  long ncol, n_lead;
  int M;
  long nblk;            // column blocks of 16
  const long* off;      // [nblk+1] offsets into idx / wts (a block's entries start at off[b])
  const int* cnt;       // [nblk] entries of block b
  const int* order;     // [nblk] blocks by descending list length: workgroup i takes block order[i]
  const int* idx;       // [nnz] observation index, ascending within a block
  const double* wts;    // [nnz][16] taper of the block's 16 columns
  const double* coef;   // [P][kCoefStride]
  const double* Ye;     // recorded ye rows
  long ye_stride;
  const double* Xin;    // [n_lead*ncol][M] perturbations (or prior members when fused_members)
  const double* xin;    // [n_lead*ncol] means (unused when fused_members)
  double* Xout;
  double* xout;
  int fused_members;
  int lead_split;       // set by the launcher: groups of slabs a column block is cut into (one workgroup each)
  long lead_chunk;      // slabs per group
};
long gc_num_blocks(long ncol);
// list build in one pass: upper bounds + device prefix sum (off[nblk] = capacity needed), then the entries
hipError_t launch_gc_bound(long ncol, long P, const double* glat, const double* ob_lat, const double* ob_hw,
                           const double* coef, int* ub, long* off, hipStream_t s);
hipError_t launch_gc_fill(long ncol, long P, const double* glat, const double* glon, const double* ob_lat,
                          const double* ob_lon, const double* ob_hw, const double* coef, double* obtrig /* [P][6] scratch */,
                          const long* off, int* cnt, int* idx, double* wts, int* order, unsigned long long* npairs,
                          hipStream_t s);
// the counting half of the list build alone: cnt[b] = observations with a non-zero taper on any of block b's 16 columns
hipError_t launch_gc_count(long ncol, long P, const double* glat, const double* glon, const double* ob_lat,
                           const double* ob_lon, const double* ob_hw, const double* coef, double* obtrig /* [P][6] scratch */,
                           int* cnt, int* blk_pairs /* [nblk] (column, ob) pairs of each block, or null */,
                           unsigned long long* npairs, hipStream_t s);
hipError_t launch_sweep_gc(const GcSweepArgs& a, hipStream_t s);

struct TransformArgs {
  const double* Xin;  // [rows][M] perturbations, or full members when fused_members
  const double* xin;  // [rows] means (unused when fused_members)
  double* Xout;       // [rows][M]
  double* xout;       // [rows] (unused when fused_members)
  long nrows;
  int M;
  const double* T;  // [M][M] row-major: Xap = Xbp * T
  const double* w;  // [M]:             xam = xbm + Xbp * w
  int fused_members;  // 1: Xin holds prior members, Xout receives posterior members
};

int sweep_slots(int M);                                  // padded row length of the quad layout
size_t diag_lds_bytes(int slots, int nb, int loc_mode);  // dynamic LDS of the diag kernel

hipError_t launch_sweep(const SweepArgs& a, hipStream_t s);
hipError_t launch_diag(const DiagArgs& a, hipStream_t s);
hipError_t launch_transform(const TransformArgs& a, hipStream_t s);
bool transform_supported(int M);

hipError_t launch_form_perts(long rows, int M, const double* X, double scale, double* xm,
                             double* Xp, hipStream_t s);
hipError_t launch_posterior(long rows, int M, const double* xm, const double* Xp, double* post,
                            hipStream_t s);
hipError_t launch_taper_table(long ncol, int nb, const double* grid_lat, const double* grid_lon,
                              const double* ob_lat, const double* ob_lon, const double* ob_hw,
                              double* W, hipStream_t s);
hipError_t launch_forward_stencil(long rows, long row_offset, int M, const double* X, long P,
                                  int npt, const int64_t* idx, const double* wts, double* HX,
                                  hipStream_t s);
// ---- f1: interpolation stencils on the device (efa_forward.hip) -----------------------------
struct InterpArgs {
  long P;
  int nvar, nt, ny, nx, latlon_1d;
  long n_grid;
  const double *glat, *glon;   // device [n_grid]
  double *sl, *cl;             // device workspace [n_grid]
  const double* valids;        // device [nt]
  const int* ob_var;           // device [P]
  const double *ob_time, *ob_lat, *ob_lon;  // device [P]
  long* nearest;               // device workspace [P][4]
  long* sten_idx;              // device out [P][8]
  double* sten_wts;            // device out [P][8]
  unsigned char* status;       // device out [P]
};
hipError_t launch_interp_stencils(const InterpArgs& a, hipStream_t s);
hipError_t launch_forward_cols(long ncol, long col_lo, long col_hi, long n_lead, int M, const double* X, long P, int npt,
                               const long* idx, const double* wts, double* HX, hipStream_t s);
hipError_t launch_fill_synthetic(long rows, long row_offset, int M, uint64_t seed, double sigma,
                                 double* X, hipStream_t s);
hipError_t launch_set_identity(int M, double* T, double* w, hipStream_t s);
hipError_t launch_phase_a_prep(long P, int M, const double* Yp, const double* ym, double* Yw, double* ymw, int carry_T,
                               unsigned long long* traj, size_t traj_words, unsigned long long sentinel, int* status,
                               const void* pack_host, void* pack_dev, size_t pack_bytes, hipStream_t s);
hipError_t launch_results_to_host(const void* src_dev, void* dst_host, size_t bytes, const int* st_dev, int* st_host, hipStream_t s);
// diagnostic: `blocks` workgroups that hold `lds_bytes` of LDS each and spin for `ms` milliseconds
hipError_t launch_occupy(int blocks, size_t lds_bytes, double ms, hipStream_t s);
hipError_t launch_contract_f32(long N, int M, long P, const float* X, const float* Ye, float* C, hipStream_t s);

}  // namespace efa
