/*
 * efa_hip.h -- C ABI of libefa_hip.so: the MI355X (gfx950) implementation of
 * efa_xray's serial EnSRF assimilation update.
 *
 * Boundary.  The reference (lmadaus/efa_xray) has no FFI; its seam is the
 * `Assimilation` subclass contract
 *     EnSRF(state, obs, nproc=1, inflation=None, verbose=True, loc=False).update()
 * (efa_xray/assimilation/ensrf.py:28-33,151).  This library sits exactly
 * between `format_prior_state` (assimilation.py:120-154) and
 * `format_posterior_state` (assimilation.py:157-171): it replaces the
 * per-observation loop ensrf.py:50-149.  The Python host
 * (efa_xray_amd.assimilation.ensrf.EnSRF) binds these symbols with ctypes;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions.
 *  - extern "C", plain pointers and sizes.  `double` is IEEE binary64.
 *  - Matrices are row-major with the ensemble member as the fastest axis:
 *    `Xp[row * M + mem]` -- the layout of `EnsembleState.to_vect()`
 *    (efa_xray/state/ensemble.py:110-114).
 *  - Pointers named *_dev are device (HBM) addresses on the context's GPU;
 *    every other pointer is host memory.  Per-observation arrays (length P)
 *    are always host memory.
 *  - The caller owns every buffer.  The library never frees or retains a
 *    caller pointer after the call returns.
 *  - Every function returns 0 on success and a negative efa_status on
 *    failure; efa_last_error() returns a thread-local message.  No C++
 *    exception crosses the ABI.
 *  - One context per GPU.  Calls on one context must be serialised by the
 *    caller.  All work is issued on the context's stream; functions that
 *    return results to host memory synchronise that stream before returning.
 *  - There is NO CPU fallback: without a usable gfx950 device
 *    efa_ctx_create() fails with EFA_ERR_NO_DEVICE.
 */
#ifndef EFA_HIP_H
#define EFA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EFA_ABI_VERSION 1

typedef enum efa_status {
  EFA_OK = 0,
  EFA_ERR_INVALID = -1,    /* bad argument (shape, NULL pointer, option) */
  EFA_ERR_NO_DEVICE = -2,  /* no HIP device / wrong architecture */
  EFA_ERR_HIP = -3,        /* a HIP runtime call failed */
  EFA_ERR_UNSUPPORTED = -4 /* valid request this build cannot serve */
} efa_status;

/* localisation modes: `loc` of EnSRF (ensrf.py:28,99) */
#define EFA_LOC_NONE 0 /* loc in (None, False) */
#define EFA_LOC_GC 1   /* loc == 'GC': Gaspari-Cohn, observation.py:117-130 */

/* how the state sweep (Phase B) is executed; results agree to ~1e-14 */
#define EFA_PATH_AUTO 0      /* transform when unlocalised and worth it */
#define EFA_PATH_SWEEP 1     /* per-batch fused covariance/gain/update sweep */
#define EFA_PATH_TRANSFORM 2 /* one pass: Xap = Xbp*T, xam = xbm + Xbp*w */

typedef struct efa_ctx efa_ctx;

/* ---- library / context --------------------------------------------------*/
int efa_abi_version(void);
const char *efa_last_error(void);
/* number of visible HIP devices (0 without a GPU; never fails the process) */
int efa_device_count(int *count);
/* create a context on HIP device `device_id` (must be gfx950) */
int efa_ctx_create(int device_id, efa_ctx **out);
int efa_ctx_destroy(efa_ctx *ctx);
/* issue all work on the caller's hipStream_t (e.g. torch's current stream).
 * NULL is the device's legacy default stream.  A new context uses a private
 * non-blocking stream; efa_ctx_set_option(ctx, "own_stream", 1) returns to it. */
int efa_ctx_set_stream(efa_ctx *ctx, void *hip_stream);
/* options: "obs_batch" (obs fused per sweep launch, 1..64, default 64),
 *          "path" (EFA_PATH_*), "timing" (0/1/2, see efa_last_timing), "pipeline" (1: run Phase A as
 *          one persistent launch when it applies, 0: per-batch kernels),
 *          "gram" (how the persistent launch leads a 64-ob block: 2 (default) in Gram space in
 *          bands of 4 obs, with or without localisation; 1 in Gram space step by step;
 *          0 on the vectors; the Gram-space leaders fall back to 0 if their cancellation guard trips),
 *          "spin_limit" (bound of the pipeline's in-kernel polls), "spin_ms" (its wall-time
 *          bound: the persistent launch gives up, and the per-batch kernels take over, when a
 *          wave has waited that long -- e.g. because another kernel keeps part of the grid from
 *          becoming resident; -1 = 100 ms + P/100 ms),
 *          "gc_onepass" (1: localised state sweep in one pass with per-column-block
 *          active lists, 0: per-batch taper tables),
 *          "own_stream" (see above);
 *          read-only: "phase_a_kind" (1 pipeline / 2 per-batch / 3 Gram pipeline / 4 band pipeline, last call),
 *          "gc_active_pairs"
 *          ((column, ob) pairs with a non-zero taper in the last one-pass sweep) */
int efa_ctx_set_option(efa_ctx *ctx, const char *key, long value);
int efa_ctx_get_option(efa_ctx *ctx, const char *key, long *value);
int efa_ctx_synchronize(efa_ctx *ctx);

/* ---- device memory for callers without their own allocator -------------*/
int efa_malloc(efa_ctx *ctx, size_t bytes, void **dev_out);
int efa_free(efa_ctx *ctx, void *dev);
int efa_memcpy_h2d(efa_ctx *ctx, void *dst_dev, const void *src, size_t bytes);
int efa_memcpy_d2h(efa_ctx *ctx, void *dst, const void *src_dev, size_t bytes);
int efa_memcpy_d2d(efa_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);

/* ---- a3: mean-removed perturbation matrix --------------------------------
 * Replaces `xbm = prior.mean(axis=1); Xbp = prior - xbm[:,None]`
 * (assimilation.py:146-147) for `rows` rows of M members.  Also used for the
 * obs-space priors of compute_ob_priors (assimilation.py:46-48).
 * `scale` multiplies the perturbations (constant inflation,
 * assimilation.py:62-68); pass 1.0 for none.  X_dev may equal Xp_dev. */
int efa_form_perts_dev(efa_ctx *ctx, long rows, int M, const double *X_dev,
                       double scale, double *xm_dev, double *Xp_dev);

/* ---- a15: posterior member values ----------------------------------------
 * Replaces `post = (xam[:,None] + Xap)[:Nstate]` (assimilation.py:168).
 * post_dev may equal Xp_dev. */
int efa_posterior_dev(efa_ctx *ctx, long rows, int M, const double *xm_dev,
                      const double *Xp_dev, double *post_dev);

/* ---- f1 (linear part): forward operator as a sparse row stencil ---------
 * HX[k,:] = sum_j wts[k*npt+j] * X[idx[k*npt+j] - row_offset, :] over the
 * stencil points owned by this shard (row_offset <= idx < row_offset+rows);
 * points owned elsewhere contribute 0 so that a sum all-reduce across shards
 * gives the full estimate (the payload of the multi-GPU exchange).
 * Stands in for Observation.estimate -> EnsembleState.interpolate
 * (observation.py:40-50, ensemble.py:170-239) once the stencil is known.
 * idx/wts are host arrays of length P*npt. */
int efa_forward_stencil_dev(efa_ctx *ctx, long rows, long row_offset, int M,
                            const double *X_dev, long P, int npt,
                            const int64_t *idx, const double *wts,
                            double *HX_dev);

/* ---- f1 (search + weights): the reference's default forward operator on the device ------
 * Observation.estimate -> EnsembleState.interpolate (observation.py:40-50,
 * ensemble.py:170-239) for P point observations at once, as a linear stencil of up
 * to 8 (state row, weight) entries per ob:
 *   space: the 4 grid points nearest in the reference's sin/cos pseudo-distance
 *          (ensemble.py:152-168; one workgroup scans the grid per ob instead of a
 *          full argsort; ties go to the lower flat index), weights = inverse
 *          great-circle distance, normalised; if a point lies within 1 km the
 *          nearest takes weight 1 (ensemble.py:178-200 -- the reference's own
 *          exact-match branch raises IndexError as written);
 *   time:  linear between the two valid times that bracket the ob, with the
 *          weights AS CODED in ensemble.py:201-224.
 * Inputs (host arrays): grid_lat/grid_lon [n_grid] degrees, n_grid = ny*nx for
 *   2-D lat/lon (latlon_1d = 0) or the length of a 1-D coordinate (latlon_1d = 1:
 *   the reference then uses one index for y and x, ensemble.py:186-190);
 *   valid_times [nt] ascending on any numeric axis; per ob: ob_var (index of
 *   Observation.obtype in EnsembleState.vars()), ob_time (same axis), ob_lat, ob_lon.
 * Outputs: the stencil stays in the context for efa_forward_interp_dev; optional
 *   host copies sten_idx [P*8] (global state rows in to_vect() order, -1 unused),
 *   sten_wts [P*8], ob_status [P]: 0 ok, 1 time outside the state's range (the
 *   reference prints a message and returns None), 2 grid index out of range (1-D
 *   lat/lon), 3 bad variable index.  Entries 0-3: the earlier valid time, 4-7: the
 *   later one (or the exact match).  Pinned to the reference by fixtures
 *   tests/golden/G9 (2-D lat/lon), G10 (1-D lat/lon) and G11 (EnSRF.update() end to
 *   end): its nearest_points / interpolate / estimate run verbatim in the build
 *   container on a duck-typed state (tests/golden/make_goldens.py). */
int efa_interp_stencils(efa_ctx *ctx, int nvar, int nt, int ny, int nx,
                        int latlon_1d, long n_grid, const double *grid_lat,
                        const double *grid_lon, const double *valid_times, long P,
                        const int32_t *ob_var, const double *ob_time,
                        const double *ob_lat, const double *ob_lon,
                        int64_t *sten_idx, double *sten_wts, uint8_t *ob_status);
/* HX[k,:] = sum of the context's stencil over the entries whose (y,x) column lies in
 * this shard's [col_lo, col_hi) of the ncol = ny*nx global columns; the shard holds
 * row lead*(col_hi-col_lo) + (col-col_lo) for lead in [0,n_lead).  A sum all-reduce
 * over the shards gives the full estimate (compute_ob_priors, assimilation.py:45-46). */
int efa_forward_interp_dev(efa_ctx *ctx, long ncol, long col_lo, long col_hi,
                           long n_lead, int M, const double *X_dev, double *HX_dev);

/* ---- a5-a14: the serial EnSRF loop, data resident in HBM -----------------
 * Replaces ensrf.py:50-149 for all P observations.
 *
 * State block (this GPU's shard):  xm_dev[rows], Xp_dev[rows*M], in/out.
 *   Row i of the shard is state element (lead, col) with
 *   i = lead*ncol + col, lead in [0,n_lead), col in [0,ncol): n_lead =
 *   nvar*ntimes, ncol = the shard's (y,x) columns -- the order of to_vect().
 *   With EFA_LOC_NONE pass n_lead=1, ncol=rows.
 * Obs block (replicated on every shard): ym_dev[P], Yp_dev[P*M], in/out:
 *   the obs-space prior means/perturbations that the reference appends to
 *   the state (assimilation.py:149-150); on return they hold the reference's
 *   final values of those augmented rows.
 * Per observation k (host arrays, length P):
 *   ob_value, ob_error (error VARIANCE, ensrf.py:79,91), ob_assim (0/1:
 *   Observation.assimilate_this, ensrf.py:74), and for EFA_LOC_GC ob_lat,
 *   ob_lon (degrees) and ob_halfwidth_km (Observation.localize_radius).
 * Grid (host arrays, length ncol, degrees; EFA_LOC_GC only): lat/lon of each
 *   local column (ensemble.py:254-267 distance_to_point).
 * Diagnostics (host arrays, length P; written for every ob as the reference
 *   does, ensrf.py:66,70,75,146-149): prior_mean, prior_var for all obs;
 *   post_mean, post_var only where assimilated[k]==1 (left untouched
 *   otherwise).
 */
int efa_ensrf_update_dev(efa_ctx *ctx, long rows, int M, long P,
                         double *xm_dev, double *Xp_dev,
                         double *ym_dev, double *Yp_dev,
                         const double *ob_value, const double *ob_error,
                         const uint8_t *ob_assim, int loc_mode,
                         const double *ob_lat, const double *ob_lon,
                         const double *ob_halfwidth_km,
                         const double *grid_lat, const double *grid_lon,
                         long ncol, long n_lead,
                         double *prior_mean, double *prior_var,
                         double *post_mean, double *post_var,
                         uint8_t *assimilated);

/* The two phases of efa_ensrf_update_dev, separately callable.
 * Phase A (obs space, serial in k, identical on every shard): consumes the
 * obs block, records the trajectory the state sweep needs and returns the
 * diagnostics.  Phase B (state space): applies the recorded trajectory to
 * `rows` state rows; independent per row, so shards never communicate.
 * efa_obs_phase_dev must precede efa_state_phase_dev on the same context. */
int efa_obs_phase_dev(efa_ctx *ctx, int M, long P, double *ym_dev,
                      double *Yp_dev, const double *ob_value,
                      const double *ob_error, const uint8_t *ob_assim,
                      int loc_mode, const double *ob_lat, const double *ob_lon,
                      const double *ob_halfwidth_km, double *prior_mean,
                      double *prior_var, double *post_mean, double *post_var,
                      uint8_t *assimilated);
/* out-of-place allowed: (xm_in, Xp_in) -> (xm_out, Xp_out); pass the same
 * pointers for in-place. */
int efa_state_phase_dev(efa_ctx *ctx, long rows, int M, const double *xm_in_dev,
                        const double *Xp_in_dev, double *xm_out_dev,
                        double *Xp_out_dev, const double *grid_lat,
                        const double *grid_lon, long ncol, long n_lead);

/* ---- a3+a5-a15 fused for resident full-member states ---------------------
 * prior members X_dev[rows*M] -> posterior members post_dev[rows*M] using the
 * trajectory recorded by efa_obs_phase_dev.  Equivalent to
 * efa_form_perts_dev + efa_state_phase_dev + efa_posterior_dev with one read
 * and one write of the state when the transform path applies.
 * post_dev may equal X_dev. */
int efa_state_cycle_dev(efa_ctx *ctx, long rows, int M, const double *X_dev,
                        double *post_dev, const double *grid_lat,
                        const double *grid_lon, long ncol, long n_lead);

/* ---- one whole cycle on resident prior members, ONE call ------------------
 * efa_obs_phase_dev followed by efa_state_cycle_dev (ensrf.py:50-149 on the obs
 * block, then on every state row; same arguments, same results bit for bit),
 * with one difference in how the device is driven: when the cycle is
 * unlocalised, takes the transform path, fits one persistent Phase-A launch and
 * X_dev / post_dev do not overlap, the state transform is put into the stream
 * BEHIND the Phase-A launch before the host has seen that launch's status, so
 * the device goes from Phase A to Phase B without a host round trip.  A launch
 * that then reports a fallback (bounded spin expired, cancellation guard) is
 * redone by the other Phase-A kernels and the transform enqueued again: a
 * wrong guess costs one wasted pass, never a result (the prior is only read).
 * obs_block_out 0 leaves ym_dev / Yp_dev as they came (the reference discards
 * the augmented obs rows: format_posterior_state keeps [:N], assimilation.py:168);
 * 1 returns the final obs block in them like efa_obs_phase_dev. */
int efa_ensrf_cycle_dev(efa_ctx *ctx, long rows, int M, long P,
                        const double *X_dev, double *post_dev, double *ym_dev,
                        double *Yp_dev, int obs_block_out,
                        const double *ob_value, const double *ob_error,
                        const uint8_t *ob_assim, int loc_mode,
                        const double *ob_lat, const double *ob_lon,
                        const double *ob_halfwidth_km, const double *grid_lat,
                        const double *grid_lon, long ncol, long n_lead,
                        double *prior_mean, double *prior_var,
                        double *post_mean, double *post_var,
                        uint8_t *assimilated);

/* ---- host-memory convenience: the augmented arrays of the reference ------
 * xbm[A], Xbp[A*M] (A = N + P) exactly as format_prior_state returns them
 * (assimilation.py:154), updated in place to the (xam, Xap) handed to
 * format_posterior_state (ensrf.py:151).  Copies to the GPU, runs
 * efa_ensrf_update_dev, copies back.  grid_lat/grid_lon have ncol entries,
 * N = n_lead*ncol. */
int efa_ensrf_update(efa_ctx *ctx, long A, long N, int M, long P, double *xbm,
                     double *Xbp, const double *ob_value,
                     const double *ob_error, const uint8_t *ob_assim,
                     int loc_mode, const double *ob_lat, const double *ob_lon,
                     const double *ob_halfwidth_km, const double *grid_lat,
                     const double *grid_lon, long ncol, long n_lead,
                     double *prior_mean, double *prior_var, double *post_mean,
                     double *post_var, uint8_t *assimilated);

/* ---- configs[4]: batched-obs dense contraction, float32 --------------------
 * C[i*P + k] = sum_m Xbp[i*M + m] * Ye[k*M + m]: the covariance numerators
 * `np.dot(Xbp, ye.T)` of ensrf.py:95 for P recorded obs-space rows at once, as
 * one (state x member).(member x obs) contraction on the matrix cores
 * (v_mfma_f32_32x32x2_f32: exact f32 FMA chains).  All pointers are device
 * memory; M must be a multiple of 4; divide by (M-1) for covariances. */
int efa_cov_contract_f32_dev(efa_ctx *ctx, long N, int M, long P,
                             const float *Xbp_f32_dev, const float *Ye_f32_dev,
                             float *C_f32_dev);

/* ---- measurement support --------------------------------------------------
 * Device time (ms) spent in the state-sweep kernels and in the obs-space
 * kernels during the most recent efa_ensrf_update_dev / efa_obs_phase_dev /
 * efa_state_phase_dev / efa_state_cycle_dev / efa_ensrf_cycle_dev call, measured with HIP events on
 * the context's stream, plus the number of state-sweep launches and the path
 * taken (EFA_PATH_SWEEP / EFA_PATH_TRANSFORM).  Timing is off by default;
 * enable with efa_ctx_set_option(ctx, "timing", 1): every state-phase call
 * then ends in a wait for its own end event.  "timing" 2 is the deferred
 * form for back-to-back cycles: no call waits for its events (an interval is
 * read when its events are next re-recorded, or here), and efa_last_timing
 * returns the SUMS of state_ms, obs_ms and state_launches over the calls since
 * the previous efa_last_timing (which it clears); it waits for the last
 * recorded events, so call it after the cycles of interest. */
int efa_last_timing(efa_ctx *ctx, double *state_ms, double *obs_ms,
                    long *state_launches, int *path_taken);

/* Fill rows of a resident state with the bench's synthetic ensemble
 * (SURVEY.md 8d): X[i,m] = mu_i + sigma*z_im with counter-based normal
 * deviates keyed by (seed, global row, member), so any sharding of the rows
 * produces identical data. */
int efa_fill_synthetic_dev(efa_ctx *ctx, long rows, long row_offset, int M,
                           uint64_t seed, double sigma, double *X_dev);

/* ---- multi-GPU, one process per GPU ----------------------------
 * The reference has no working multi-process path; its sketch (assimilation.py:186-193,
 * ensemble.py:98-106) computes the obs priors once and hands them to every
 * worker.  Here the state is sharded by (y,x) column; each rank calls
 * efa_forward_interp_dev / efa_forward_stencil_dev on its own columns and the
 * P x M partial sums are added over the ranks by ONE all-reduce
 * (ncclAllReduce, sum, float64, over xGMI) issued on the context's stream.
 * Afterwards every rank runs efa_obs_phase_dev on the identical obs block and
 * efa_state_cycle_dev on its own rows: no per-observation communication.
 *
 * The communicator lives inside the context.  librccl is opened with dlopen by
 * efa_comm_unique_id / efa_comm_init, so single-GPU callers never load it.
 * Rank 0 obtains an id and hands the EFA_COMM_ID_BYTES bytes to the other
 * ranks by any means (MPI, a file, torch.distributed); then every rank calls
 * efa_comm_init collectively. */
#define EFA_COMM_ID_BYTES 128
int efa_comm_unique_id(uint8_t *id_out /* [EFA_COMM_ID_BYTES] */);
int efa_comm_init(efa_ctx *ctx, const uint8_t *id, int rank, int world);
int efa_comm_destroy(efa_ctx *ctx);
/* buf_dev[count] <- sum over ranks, in place, on the context's stream (asynchronous:
 * ordered before whatever is issued on the context afterwards) */
int efa_allreduce_sum_dev(efa_ctx *ctx, double *buf_dev, long count);

/* Cost of the localised state sweep per block of 16 consecutive (y,x) columns
 * of a grid: block_count[b] = number of assimilated observations whose
 * Gaspari-Cohn weight (observation.py:117-130 on ensemble.py:254-267 distances) is
 * non-zero on at least one of columns 16b..16b+15 -- the length of the block's
 * active list in the one-pass sweep; block_pairs[b] (optional) = (column,
 * observation) pairs with a non-zero weight inside the block (the sweep's waves
 * skip an observation that is zero on their four columns, so its work follows the
 * pairs); *active_pairs = their total.  Host arrays of (ncol+15)/16 entries.
 * Used to cut the columns into contiguous shards of equal cost
 * (the reference's sketch cuts equal chunks, ensemble.py:98-106, which leaves
 * polar shards of a lat/lon grid with several times the work). */
int efa_gc_block_counts(efa_ctx *ctx, long ncol, const double *grid_lat,
                        const double *grid_lon, long P, const double *ob_lat,
                        const double *ob_lon, const double *ob_halfwidth_km,
                        const uint8_t *ob_assim, int32_t *block_count,
                        int32_t *block_pairs, uint64_t *active_pairs);

#ifdef __cplusplus
}
#endif
#endif /* EFA_HIP_H */
