// C-ABI layer of libefa_hip.so (see include/efa_hip.h): contexts, workspaces,
// the Phase A / Phase B drivers and the host-memory convenience entry point.
#include "../../include/efa_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and enums only: librccl is opened with dlopen when a communicator is asked for

#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "efa_internal.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define EFA_HIP(expr)                                                                       \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      return fail(EFA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                  __LINE__);                                                                \
  } while (0)

#define EFA_TRY(expr)          \
  do {                         \
    int _s = (expr);           \
    if (_s != EFA_OK) return _s; \
  } while (0)

// grow-only device buffer
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool view = false;  // p points into another DevBuf (carve): never freed here
  void carve(void* base, size_t bytes) {
    if (p && !view) (void)hipFree(p);
    p = base;
    cap = bytes;
    view = true;
  }
  int reserve(size_t bytes) {
    if (bytes <= cap) return EFA_OK;
    if (view) return fail(EFA_ERR_INVALID, "internal: reserve() on a carved buffer");
    if (p) {
      hipError_t e = hipFree(p);
      p = nullptr;
      cap = 0;
      if (e != hipSuccess) return fail(EFA_ERR_HIP, "hipFree failed: %s", hipGetErrorString(e));
    }
    size_t want = bytes + (bytes >> 3) + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      p = nullptr;
      return fail(EFA_ERR_HIP, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    cap = want;
    return EFA_OK;
  }
  void release() {
    if (p && !view) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    view = false;
  }
  template <typename T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

// grow-only pinned host staging: one asynchronous copy each way instead of one (synchronous, staged by the
// runtime) copy per pageable caller array
struct PinBuf {
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return EFA_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = bytes + (bytes >> 2) + 256;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) {
      p = nullptr;
      return fail(EFA_ERR_HIP, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    cap = want;
    return EFA_OK;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace

struct efa_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  long obs_batch = 64;
  long path = EFA_PATH_AUTO;
  long timing = 0;
  long use_gram = 2;       // persistent kernel's leader: 2 band leader (with and without localisation), 1 Gram leader step by step, 0 vector chain
  long use_pipeline = 1;   // persistent Phase-A kernel when it applies (else per-batch kernels)
  long spin_limit = 4000000;
  long spin_ms = -1;       // wall-time bound of the persistent Phase-A launch; -1: 100 ms + P/100 ms
  int cu_count = 0;
  hipStream_t dbg_stream = nullptr;  // diagnostic occupier (options debug_occupy_*)
  long dbg_occupy_blocks = 0;
  long pipe_debug = 0;
  long gc_onepass = 1;     // localised state sweep in one pass with per-column-block active lists

  // --- trajectory recorded by the last obs phase --------------------------
  bool have_traj = false;
  int M = 0;
  long P = 0;
  int loc_mode = EFA_LOC_NONE;
  long n_active = 0;
  bool have_transform = false;   // identity rows were carried: (T, w) valid
  std::vector<uint8_t> h_assim;  // host copy of ob_assim
  std::vector<double> h_hw;      // host copy of ob_halfwidth_km, sanitised for unassimilated obs
  DevBuf Ye_rec, coef;           // [P][M], [P][4]
  DevBuf traj, tw_mat, status, dbg;  // pipeline: trajectory records, GC obs-obs taper, status words, stamps
  const double* ye_ptr = nullptr;  // where Phase B reads the recorded ye rows
  long ye_stride = 0;
  int phase_a_kind = 0;          // 1 pipeline, 2 per-batch kernels
  DevBuf ob_pack, out_pack;  // the per-ob inputs / diagnostics below are carved out of these two allocations
  PinBuf pin_in, pin_out;    // their pinned host images: one H2D and one D2H per call
  PinBuf pin_fs;             // pinned image of the forward-operator stencil
  hipEvent_t ev_fs = nullptr;  // its last host-to-device copy
  size_t fs_valid_n = 0;       // the device copy fs_idx holds the pinned image's first fs_valid_n stencil entries ...
  const void* fs_valid_dev = nullptr;  // ... if fs_idx and pin_fs are still these allocations
  const void* fs_valid_pin = nullptr;
  DevBuf ob_val, ob_err, ob_asm, ob_lat, ob_lon, ob_hw, ob_errsq;  // device copies [P] ([P][4] the last)
  DevBuf d_prior_mean, d_prior_var, d_post_mean, d_post_var, d_assimilated;
  DevBuf Yw, ymw;  // obs block workspace [(P+M)][M], [(P+M)]
  DevBuf win_Y, win_m;  // rows of one Phase-A window + the transform rows (only when P exceeds one persistent launch)
  // --- state phase workspaces ---------------------------------------------
  DevBuf W;           // taper table [nb][ncol]
  DevBuf gc_cnt, gc_ub, gc_order, gc_obtrig, gc_off, gc_idx, gc_wts, gc_pairs;  // one-pass GC sweep: CSR active lists
  long gc_active_pairs = 0;  // (column, ob) pairs with a non-zero taper in the last one-pass sweep
  // What depends on the GEOMETRY of a localised cycle only -- the obs' positions, radii and assimilate flags, the column grid --
  // is kept from one cycle to the next while that geometry is unchanged (a fixed observing network on a fixed grid): the obs-obs
  // taper table of Phase A and the per-block active lists of the one-pass sweep (indices, tapers, hand-out order; the gains are
  // folded in by the sweep itself, cycle by cycle).  Compared by content on the host, never by pointer.
  std::vector<double> geo_lat, geo_lon, geo_hw;
  std::vector<uint8_t> geo_assim;
  long geo_serial = 0;       // bumped whenever the obs geometry of a call differs from the previous call's
  long grid_serial = 0;      // bumped whenever the device copy of the column grid is rewritten
  long tw_serial = -1, tw_Pw = -1, tw_Rw = -1;  // what the obs-obs taper table on the device was built from
  const void* tw_ptr = nullptr;
  bool gc_list_valid = false;
  long gc_list_geo = -1, gc_list_grid = -1, gc_list_ncol = -1, gc_list_P = -1;
  const void* gc_list_ptrs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  long geometry_reuse = 1;   // option "geometry_reuse" (0: rebuild every cycle)
  bool gc_pairs_pending = false;  // ... still on the device (read when asked for, or before the counter is cleared again: a read
                                  // behind the sweep would hold the host until the sweep is done, cycle after cycle)
  PinBuf pin_grid;           // pinned mirror of the column lat/lon on the device (glat | glon)
  long grid_ncol = -1;       // columns the mirror and the device copies hold (-1: none)
  bool grid_ready = false;   // efa_ensrf_cycle_dev brought the grid up to date ahead of Phase A: the state phase must not again
  DevBuf glat, glon;  // grid lat/lon [ncol]
  DevBuf xm_ws;       // means for efa_state_cycle_dev
  // --- f1: interpolation stencils -------------------------------------------------
  DevBuf fs_idx, fs_wts;  // efa_forward_stencil_dev staging
  DevBuf f_glat, f_glon, f_sl, f_cl, f_valids, f_var, f_time, f_lat, f_lon, f_near, f_idx, f_wts, f_status;
  long f_P = 0;       // observations of the stencil held in f_idx / f_wts (0: none)
  // --- host-memory API buffers ----------------------------------------------
  DevBuf h_xm, h_Xp, h_ym, h_Yp;
  // --- multi-GPU exchange step: an RCCL communicator owned by the context (efa_comm_init) -------------
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  DevBuf gcc_lat, gcc_lon, gcc_oblat, gcc_oblon, gcc_obhw, gcc_coef, gcc_trig, gcc_cnt, gcc_pairs;  // efa_gc_block_counts
  // --- timing -----------------------------------------------------------------
  hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // obs phase 0..1; state phase 2..3 and (the fused cycle's second pair) 4..5; 6: Phase A's results on the host (fused cycle)
  double state_ms = 0.0, obs_ms = 0.0;
  bool obs_ms_pending = false;  // ev[0] .. ev[obs_end_ev] of the last obs phase not read yet
  int obs_end_ev = 1;           // 1, or the start event of the state pair a speculative transform was put behind
  bool state_ms_pending = false;  // ev[2] .. ev[3] of the last state phase not read yet
  bool state_ms_pending2 = false; // ev[4] .. ev[5] likewise (efa_ensrf_cycle_dev alternates the pairs: it records a state phase's
                                  // events BEFORE the stream is synchronised, while the previous cycle's may still be unread)
  // efa_ensrf_cycle_dev: Phase B enqueued behind Phase A before Phase A's status is known
  struct Spec {
    bool armed = false, launched = false;
    const double* X = nullptr;
    double* post = nullptr;
    long rows = 0;
    int pair = 0;  // event pair of the launched transform
    bool obs_out = true;
  } spec;
  double state_ms_sum = 0.0, obs_ms_sum = 0.0;  // timing 2: sums since the previous efa_last_timing
  long state_launches_sum = 0;
  long state_launches = 0;
  int path_taken = 0;
};

namespace {

using namespace efa;

// "timing" 2 (deferred): no phase call waits for its own events -- the host may run ahead of the device from one cycle into the
// next.  An interval is read when its events are about to be recorded again (the calls in between have synchronised the stream
// since: the wait returns at once) or in efa_last_timing, and added to running sums.
void harvest_obs_ms(efa_ctx* c) {
  if (!c->obs_ms_pending) return;
  float ms = 0.f;
  if (hipEventSynchronize(c->ev[c->obs_end_ev]) == hipSuccess && hipEventElapsedTime(&ms, c->ev[0], c->ev[c->obs_end_ev]) == hipSuccess) {
    c->obs_ms = ms;
    c->obs_ms_sum += ms;
  } else {
    (void)hipGetLastError();
  }
  c->obs_ms_pending = false;
}
void harvest_state_pair(efa_ctx* c, int pair) {
  bool& pending = pair ? c->state_ms_pending2 : c->state_ms_pending;
  if (!pending) return;
  float ms = 0.f;
  if (hipEventSynchronize(c->ev[3 + 2 * pair]) == hipSuccess &&
      hipEventElapsedTime(&ms, c->ev[2 + 2 * pair], c->ev[3 + 2 * pair]) == hipSuccess) {
    c->state_ms = ms;
    c->state_ms_sum += ms;
  } else {
    (void)hipGetLastError();
  }
  pending = false;
}
void harvest_state_ms(efa_ctx* c) {
  harvest_state_pair(c, 0);
  harvest_state_pair(c, 1);
}
// end of a state-phase call: timing 1 waits and reads, timing 2 leaves the interval pending
int finish_state_timing(efa_ctx* c, hipStream_t s) {
  c->state_launches_sum += c->state_launches;
  if (!c->timing) return EFA_OK;
  EFA_HIP(hipEventRecord(c->ev[3], s));
  c->state_ms_pending = true;
  if (c->timing == 1) harvest_state_ms(c);
  return EFA_OK;
}

int use(efa_ctx* c) {
  if (!c) return fail(EFA_ERR_INVALID, "null context");
  EFA_HIP(hipSetDevice(c->device));
  return EFA_OK;
}

long effective_batch(const efa_ctx* c, int M) {
  long b = c->obs_batch;
  if (b < 1) b = 1;
  if (b > kMaxBatch) b = kMaxBatch;
  // LDS budgets: the sweep's image of the batch (ye rows + coefs, either lane layout) and the
  // diag kernel's ring (ye rows + scalars + the GC taper matrix) must fit one CU's 160 KiB.
  const long s4 = sweep_slots(M), s16 = 32L * ((M + 31) / 32);
  const long per_ob = ((s4 > s16 ? s4 : s16) + kCoefStride) * (long)sizeof(double);
  while (b > 1 && (b * per_ob + 64L * kMaxBatch * 8 > 150L * 1024 || (long)diag_lds_bytes((int)s4, (int)b, 1) > 150L * 1024)) --b;
  return b;
}

int h2d(efa_ctx* c, DevBuf& b, const void* src, size_t bytes) {
  EFA_TRY(b.reserve(bytes ? bytes : 8));
  if (bytes) EFA_HIP(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
  return EFA_OK;
}

int check_common(int M, long P) {
  if (M < 2) return fail(EFA_ERR_INVALID, "ensemble size M=%d must be >= 2 (covariance divides by M-1)", M);
  if (M > kMaxMembers) return fail(EFA_ERR_UNSUPPORTED, "ensemble size M=%d exceeds the built maximum %d", M, kMaxMembers);
  if (P < 0) return fail(EFA_ERR_INVALID, "negative observation count");
  return EFA_OK;
}

bool auto_transform(int M, long n_active, bool member_form);  // (with Phase B's path choice, below)

// ---- Phase A ---------------------------------------------------------------
int obs_phase(efa_ctx* c, int M, long P, double* ym_dev, double* Yp_dev, const double* ob_value,
              const double* ob_error, const uint8_t* ob_assim, int loc_mode, const double* ob_lat,
              const double* ob_lon, const double* ob_hw, double* prior_mean, double* prior_var,
              double* post_mean, double* post_var, uint8_t* assimilated) {
  EFA_TRY(check_common(M, P));
  if (loc_mode != EFA_LOC_NONE && loc_mode != EFA_LOC_GC) return fail(EFA_ERR_INVALID, "loc_mode %d", loc_mode);
  c->have_traj = false;
  c->M = M;
  c->P = P;
  c->loc_mode = loc_mode;
  c->n_active = 0;
  c->have_transform = false;
  c->spec.launched = false;
  harvest_obs_ms(c);
  c->obs_ms = 0.0;
  if (P == 0) {
    c->have_traj = true;
    c->h_assim.clear();
    return EFA_OK;
  }
  if (!ym_dev || !Yp_dev || !ob_value || !ob_error || !ob_assim)
    return fail(EFA_ERR_INVALID, "null observation array");
  if (loc_mode == EFA_LOC_GC) {
    if (!ob_lat || !ob_lon || !ob_hw) return fail(EFA_ERR_INVALID, "GC localisation needs ob_lat/ob_lon/ob_halfwidth_km");
    // the reference reads localize_radius only for obs it assimilates (ensrf.py:74-76 comes before :101):
    // an unassimilated ob may carry any radius; it is replaced by a harmless one before it goes to the device
    c->h_hw.assign(ob_hw, ob_hw + P);
    for (long k = 0; k < P; ++k) {
      if (!ob_assim[k]) {
        c->h_hw[k] = 1.0;
        continue;
      }
      if (!(ob_hw[k] == ob_hw[k]) || ob_hw[k] == 0.0)
        return fail(EFA_ERR_INVALID, "observation %ld: localize_radius must be a non-zero number for loc='GC' "
                    "(the reference raises in abs(None), observation.py:120)", k);
    }
    ob_hw = c->h_hw.data();
  }
  c->h_assim.assign(ob_assim, ob_assim + P);
  for (long k = 0; k < P; ++k) c->n_active += ob_assim[k] ? 1 : 0;
  if (loc_mode == EFA_LOC_GC) {
    const size_t nb8 = (size_t)P * sizeof(double);
    const bool same = (long)c->geo_lat.size() == P && std::memcmp(c->geo_lat.data(), ob_lat, nb8) == 0 &&
                      std::memcmp(c->geo_lon.data(), ob_lon, nb8) == 0 && std::memcmp(c->geo_hw.data(), ob_hw, nb8) == 0 &&
                      std::memcmp(c->geo_assim.data(), ob_assim, (size_t)P) == 0;
    if (!same) {
      c->geo_lat.assign(ob_lat, ob_lat + P);
      c->geo_lon.assign(ob_lon, ob_lon + P);
      c->geo_hw.assign(ob_hw, ob_hw + P);     // (sanitised above)
      c->geo_assim.assign(ob_assim, ob_assim + P);
      c->geo_serial++;
    }
  }

  const bool carry_T = (loc_mode == EFA_LOC_NONE) && transform_supported(M) && (c->path != EFA_PATH_SWEEP);
  const long extra = carry_T ? M : 0;
  const long R = P + extra;
  const size_t dP = (size_t)P * sizeof(double);

  size_t pack_bytes = 0;
  // per-ob inputs: [value | error | assim bytes | {error, sqrt(error), assimilate (1.0 / 0.0), 0} x P | lat | lon | halfwidth] in one
  // allocation, ONE H2D from pinned memory (the last three slots only with localisation); the four-double records are the band
  // leader's per-ob constants, fetched with wave-uniform loads
  {
    const bool gc = loc_mode == EFA_LOC_GC;
    const size_t slot = ((size_t)P * sizeof(double) + 255) & ~(size_t)255;
    const size_t total = 10 * slot;
    EFA_TRY(c->ob_pack.reserve(total));
    EFA_TRY(c->pin_in.reserve(total));
    char* hb = static_cast<char*>(c->pin_in.p);
    char* db = static_cast<char*>(c->ob_pack.p);
    std::memcpy(hb, ob_value, dP);
    std::memcpy(hb + slot, ob_error, dP);
    std::memcpy(hb + 2 * slot, ob_assim, (size_t)P);
    {
      double* ec = reinterpret_cast<double*>(hb + 3 * slot);
      for (long k = 0; k < P; ++k) {
        ec[4 * k] = ob_error[k];
        ec[4 * k + 1] = std::sqrt(ob_error[k]);
        ec[4 * k + 2] = ob_assim[k] ? 1.0 : 0.0;
        ec[4 * k + 3] = 0.0;
      }
    }
    if (gc) {
      std::memcpy(hb + 7 * slot, ob_lat, dP);
      std::memcpy(hb + 8 * slot, ob_lon, dP);
      std::memcpy(hb + 9 * slot, ob_hw, dP);
    }
    c->ob_val.carve(db, slot);
    c->ob_err.carve(db + slot, slot);
    c->ob_asm.carve(db + 2 * slot, slot);
    c->ob_errsq.carve(db + 3 * slot, 4 * slot);
    c->ob_lat.carve(db + 7 * slot, slot);
    c->ob_lon.carve(db + 8 * slot, slot);
    c->ob_hw.carve(db + 9 * slot, slot);
    pack_bytes = gc ? total : 7 * slot;  // goes to the device inside the prep launch below (read from the mapped pinned buffer)
  }
  EFA_TRY(c->Ye_rec.reserve((size_t)P * M * sizeof(double)));
  EFA_TRY(c->coef.reserve((size_t)P * kCoefStride * sizeof(double)));
  // per-ob diagnostics: [prior_mean | prior_var | post_mean | post_var | assimilated bytes], one D2H at the end
  const size_t oslot = ((size_t)P * sizeof(double) + 255) & ~(size_t)255;
  {
    EFA_TRY(c->out_pack.reserve(5 * oslot));
    EFA_TRY(c->pin_out.reserve(5 * oslot));
    char* db = static_cast<char*>(c->out_pack.p);
    c->d_prior_mean.carve(db, oslot);
    c->d_prior_var.carve(db + oslot, oslot);
    c->d_post_mean.carve(db + 2 * oslot, oslot);
    c->d_post_var.carve(db + 3 * oslot, oslot);
    c->d_assimilated.carve(db + 4 * oslot, oslot);
  }
  EFA_TRY(c->Yw.reserve((size_t)R * M * sizeof(double)));
  EFA_TRY(c->ymw.reserve((size_t)R * sizeof(double)));

  hipStream_t s = c->stream;
  if (c->timing) EFA_HIP(hipEventRecord(c->ev[0], s));
  double* Yw = c->Yw.as<double>();
  double* ymw = c->ymw.as<double>();
  // (the copies of the caller's block into the working rows, the identity rows, the sentinel fill of the records and the
  //  clearing of the status words are ONE launch: launch_phase_a_prep below, once the record stride is known)

  // ---- Phase A in WINDOWS of observations -----------------------------------------------------------------
  // A persistent launch keeps 64 obs rows per workgroup and needs its whole grid resident: at most kPipeMaxWGs * 64
  // rows (the window's obs + the M carried transform rows).  More observations are taken window by window: the
  // window's rows and the transform rows go through one persistent launch (in a workspace when the window is not the
  // whole block), and the rows of all OTHER observations -- earlier windows' (the reference keeps updating them,
  // ensrf.py:141 acts on every augmented row) and later ones' -- take the window's trajectory through the per-batch
  // sweep kernel, 64 obs per pass.  A window whose launch gives up (bounded spin, cancellation guard twice) is redone,
  // for its own observations only, by the per-batch kernels.
  // Without localisation a window that is not the whole block carries a SECOND set of identity rows: they come out as the
  // window's own transform (T_w, w_w), which then updates all other rows of the block in one k_transform pass instead of
  // one sweep pass per 64 obs.
  const long Wone = (long)kPipeMaxWGs * kPipeRowsPerWG - extra;            // one window covers the block up to here
  const long Wmax = (P <= Wone) ? Wone : Wone - extra;                       // else: two sets of extra rows per window
  const bool pipe_ok = c->use_pipeline && Wmax > 0 && pipeline_supported(M, (P <= Wone ? P + extra : Wmax + 2 * extra));
  const long TS_std = traj_stride(M), TS_band = band_traj_stride(M);
  const long TS = TS_std > TS_band ? TS_std : TS_band;
  const long B = effective_batch(c, M);
  bool traj_kind_band = false, any_pipeline = false, any_batch = false;
  bool diag_on_host = false;  // the diagnostics are already in pin_out (copied with the status words of the one launch that did it all)
  if (pipe_ok) {
    EFA_TRY(c->traj.reserve((size_t)P * TS * sizeof(unsigned long long)));
    EFA_TRY(c->status.reserve(3 * sizeof(int)));
  }
  EFA_HIP(launch_phase_a_prep(P, M, Yp_dev, ym_dev, Yw, ymw, carry_T ? 1 : 0, pipe_ok ? c->traj.as<unsigned long long>() : nullptr,
                              pipe_ok ? (size_t)P * TS : 0, kTrajSentinel, pipe_ok ? c->status.as<int>() : nullptr,
                              c->pin_in.p, c->ob_pack.p, pack_bytes, s));
  bool status_clear = pipe_ok;  // (cleared by the prep launch: the first window's launch needs no memset of its own)
  // rows [lo, hi) of the obs block take obs [b0, b0 + nb) from (Ye, ye_stride): the per-batch sweep
  auto sweep_rows = [&](long b0, int nb, const double* Ye, long ye_stride, long skip_lo, long skip_hi, long nrows) -> int {
    SweepArgs a{};
    a.Xin = Yw;
    a.xin = ymw;
    a.Xout = Yw;
    a.xout = ymw;
    a.nrows = nrows;
    a.M = M;
    a.Ye = Ye;
    a.ye_stride = ye_stride;
    a.coef = c->coef.as<double>() + (size_t)b0 * kCoefStride;
    a.nb = nb;
    a.taper_mode = (loc_mode == EFA_LOC_GC) ? kTaperObs : kTaperNone;
    a.row_lat = c->ob_lat.as<double>();
    a.row_lon = c->ob_lon.as<double>();
    a.ob_lat = c->ob_lat.as<double>() + b0;
    a.ob_lon = c->ob_lon.as<double>() + b0;
    a.ob_hw = c->ob_hw.as<double>() + b0;
    a.skip_lo = skip_lo;
    a.skip_hi = skip_hi;
    a.taper_rows = P;
    EFA_HIP(launch_sweep(a, s));
    return EFA_OK;
  };
  // obs [w0, w1) by the per-batch kernels (k_diag on the batch's own rows, k_sweep on every other row of the block)
  auto batch_window = [&](long w0, long w1) -> int {
    for (long b0 = w0; b0 < w1; b0 += B) {
      const int nb = (int)((w1 - b0 < B) ? (w1 - b0) : B);
      DiagArgs d{};
      d.Yp = Yw;
      d.ym = ymw;
      d.M = M;
      d.b0 = b0;
      d.nb = nb;
      d.ob_value = c->ob_val.as<double>();
      d.ob_error = c->ob_err.as<double>();
      d.ob_assim = c->ob_asm.as<uint8_t>();
      d.loc_mode = loc_mode;
      d.ob_lat = c->ob_lat.as<double>();
      d.ob_lon = c->ob_lon.as<double>();
      d.ob_hw = c->ob_hw.as<double>();
      d.Ye_rec = c->Ye_rec.as<double>();
      d.coef = c->coef.as<double>();
      d.prior_mean = c->d_prior_mean.as<double>();
      d.prior_var = c->d_prior_var.as<double>();
      d.post_mean = c->d_post_mean.as<double>();
      d.post_var = c->d_post_var.as<double>();
      d.assimilated = c->d_assimilated.as<uint8_t>();
      EFA_HIP(launch_diag(d, s));
      long act = 0;
      for (int k = 0; k < nb; ++k) act += ob_assim[b0 + k] ? 1 : 0;
      if (act == 0 || R == nb) continue;
      EFA_TRY(sweep_rows(b0, nb, c->Ye_rec.as<double>() + (size_t)b0 * M, M, b0, b0 + nb, R));
    }
    return EFA_OK;
  };
  const long nwin = pipe_ok ? (P + Wmax - 1) / Wmax : 1;
  for (long w = 0; w < nwin; ++w) {
    const long w0 = pipe_ok ? w * Wmax : 0, w1 = pipe_ok ? ((w0 + Wmax < P) ? w0 + Wmax : P) : P;
    const long Pw = w1 - w0, Rw = Pw + ((nwin == 1) ? extra : 2 * extra);
    bool done = false;
    const bool tw_fits = (loc_mode != EFA_LOC_GC) || ((size_t)Pw * (size_t)Rw * sizeof(double) <= ((size_t)3 << 30));
    if (pipe_ok && tw_fits) {
      // the launch's rows: the block itself when one window covers it, else a workspace [window rows | transform rows]
      const bool direct = (nwin == 1);
      double* Wy = Yw;
      double* Wm = ymw;
      if (!direct) {
        EFA_TRY(c->win_Y.reserve((size_t)Rw * M * sizeof(double)));
        EFA_TRY(c->win_m.reserve((size_t)Rw * sizeof(double)));
        Wy = c->win_Y.as<double>();
        Wm = c->win_m.as<double>();
      }
      auto stage_in = [&]() -> int {  // (also the restore after a failed attempt: the block keeps the pre-launch rows)
        if (direct) return EFA_OK;
        EFA_HIP(hipMemcpyAsync(Wy, Yw + (size_t)w0 * M, (size_t)Pw * M * sizeof(double), hipMemcpyDeviceToDevice, s));
        EFA_HIP(hipMemcpyAsync(Wm, ymw + w0, (size_t)Pw * sizeof(double), hipMemcpyDeviceToDevice, s));
        if (extra) {
          EFA_HIP(hipMemcpyAsync(Wy + (size_t)Pw * M, Yw + (size_t)P * M, (size_t)extra * M * sizeof(double), hipMemcpyDeviceToDevice, s));
          EFA_HIP(hipMemcpyAsync(Wm + Pw, ymw + P, (size_t)extra * sizeof(double), hipMemcpyDeviceToDevice, s));
          EFA_HIP(launch_set_identity(M, Wy + (size_t)(Pw + extra) * M, Wm + Pw + extra, s));  // the window's own transform
        }
        return EFA_OK;
      };
      EFA_TRY(stage_in());
      if (!status_clear) EFA_HIP(hipMemsetAsync(c->status.p, 0, 3 * sizeof(int), s));
      status_clear = false;
      PipeArgs pa{};
      pa.Yp = Wy;
      pa.ym = Wm;
      pa.R = Rw;
      pa.P = Pw;
      pa.M = M;
      pa.ob_value = c->ob_val.as<double>() + w0;
      pa.ob_error = c->ob_err.as<double>() + w0;
      pa.ob_assim = c->ob_asm.as<uint8_t>() + w0;
      pa.ob_errsq = c->ob_errsq.as<double>() + 4 * w0;
      pa.loc_mode = loc_mode;
      pa.tw = nullptr;
      if (loc_mode == EFA_LOC_GC) {
        EFA_TRY(c->tw_mat.reserve((size_t)Pw * Rw * sizeof(double)));
        EFA_TRY(c->gc_obtrig.reserve((size_t)Pw * 6 * sizeof(double)));
        const bool tw_ok = c->geometry_reuse && direct && c->tw_serial == c->geo_serial && c->tw_Pw == Pw && c->tw_Rw == Rw &&
                           c->tw_ptr == c->tw_mat.p;
        if (!tw_ok) {
          EFA_HIP(launch_obs_taper_matrix(Pw, Rw, c->ob_lat.as<double>() + w0, c->ob_lon.as<double>() + w0, c->ob_hw.as<double>() + w0,
                                          c->gc_obtrig.as<double>(), c->tw_mat.as<double>(), s));
          c->tw_serial = direct ? c->geo_serial : -1;  // (a window's table is not the whole block's)
          c->tw_Pw = Pw;
          c->tw_Rw = Rw;
          c->tw_ptr = c->tw_mat.p;
        }
        pa.tw = c->tw_mat.as<double>();
      }
      pa.coef = c->coef.as<double>() + (size_t)w0 * kCoefStride;
      pa.prior_mean = c->d_prior_mean.as<double>() + w0;
      pa.prior_var = c->d_prior_var.as<double>() + w0;
      pa.post_mean = c->d_post_mean.as<double>() + w0;
      pa.post_var = c->d_post_var.as<double>() + w0;
      pa.assimilated = c->d_assimilated.as<uint8_t>() + w0;
      pa.status = c->status.as<int>();
      pa.spin_limit = c->spin_limit;
      pa.spin_ticks = (c->spin_ms >= 0 ? c->spin_ms : 100 + Pw / 100) * 100000L;  // s_memrealtime runs at 100 MHz
      pa.cu_count = c->cu_count;
      pa.debug = (int)c->pipe_debug;
      pa.dbg = nullptr;
      if (c->pipe_debug & 4) {
        EFA_TRY(c->dbg.reserve((size_t)P * 8 * sizeof(unsigned long long)));
        if (w == 0) EFA_HIP(hipMemsetAsync(c->dbg.p, 0, (size_t)P * 8 * sizeof(unsigned long long), s));
        pa.dbg = c->dbg.as<unsigned long long>() + (size_t)w0 * 8;
      }
      // attempt 0: Gram-space leader (option "gram"); attempt 1: vector-chain pipeline.  A failed
      // attempt (bounded spin expired, or the Gram downdate's cancellation guard) may have let
      // finished workgroups write their rows back, so the launch's rows are restored before the next.
      // An attempt is skipped when its grid cannot be co-resident (occupancy query in the launcher), and after an
      // attempt whose bounded spins EXPIRED (status 1: some workgroups never became resident, e.g. another
      // kernel holds CUs) the other persistent kernel is not tried either: it has the same residency need.
      // All windows of a call must leave records of ONE layout (Phase B reads them with one stride): once a window
      // has run as the band leader, later windows do not fall back to the vector chain (they go to the per-batch kernels).
      int first_kind = (c->use_gram >= 2 && pipeline_band_supported(M, Rw, loc_mode)) ? 4
                       : (c->use_gram >= 1 && pipeline_gram_supported(M, Rw, loc_mode)) ? 3 : 1;
      if (any_pipeline && !traj_kind_band && first_kind == 4) first_kind = (c->use_gram >= 1 && pipeline_gram_supported(M, Rw, loc_mode)) ? 3 : 1;
      for (int attempt = (first_kind == 1) ? 1 : 0; attempt < 2 && !done; ++attempt) {
        const int kind = (attempt == 0) ? first_kind : 1;
        if (any_pipeline && traj_kind_band != (kind == 4)) break;
        const long TSk = (kind == 4) ? TS_band : TS_std;
        pa.traj = c->traj.as<unsigned long long>() + (size_t)w0 * TSk;
        const hipError_t le = kind == 4 ? launch_pipeline_band(pa, s) : kind == 3 ? launch_pipeline_gram(pa, s) : launch_pipeline(pa, s);
        if (le == hipErrorCooperativeLaunchTooLarge) {
          (void)hipGetLastError();
          continue;
        }
        EFA_HIP(le);
        // ONE host round trip per launch: the status words and -- when this launch is the whole Phase A -- the diagnostics
        // it wrote come back together, into pinned memory (a second copy + synchronise after the status was known left the
        // device idle for ~40 us before Phase B; a pageable destination made the status copy itself a staged one)
        int* st = reinterpret_cast<int*>(static_cast<char*>(c->pin_out.p) + 5 * oslot - 64);
        if (direct) EFA_HIP(launch_results_to_host(c->out_pack.p, c->pin_out.p, 4 * oslot + (size_t)P, c->status.as<int>(), st, s));
        else EFA_HIP(hipMemcpyAsync(st, c->status.p, 3 * sizeof(int), hipMemcpyDeviceToHost, s));
        // efa_ensrf_cycle_dev: the state transform goes into the stream HERE, behind the launch whose status is not known
        // yet -- it reads [T | w] from the launch's working rows and writes only the caller's posterior; a launch that
        // reports a fallback is redone below and the transform enqueued again (by the caller), so a wrong guess costs time, never
        // a result.  The device then runs Phase A -> Phase B with no host round trip in between.
        bool spec_now = false;
        if (c->spec.armed && c->spec.rows > 0 && direct && carry_T && c->n_active > 0 &&
            (c->path == EFA_PATH_TRANSFORM || (c->path == EFA_PATH_AUTO && auto_transform(M, c->n_active, true)))) {
          const int pr = c->state_ms_pending ? 1 : 0;
          if (c->timing) harvest_state_pair(c, pr);  // (both pairs unread cannot happen across the wait below; kept correct anyway)
          // ONE event between Phase A and the transform (each record idles the stream ~6 us): the status words and diagnostics are on
          // the host -- what the host waits for below --, the obs interval ends and the state interval of this pair begins
          EFA_HIP(hipEventRecord(c->ev[2 + 2 * pr], s));
          c->obs_end_ev = 2 + 2 * pr;
          TransformArgs t{};
          t.Xin = c->spec.X;
          t.Xout = c->spec.post;
          t.nrows = c->spec.rows;
          t.M = M;
          t.T = Yw + (size_t)P * M;
          t.w = ymw + P;
          t.fused_members = 1;
          EFA_HIP(launch_transform(t, s));
          if (c->timing) EFA_HIP(hipEventRecord(c->ev[3 + 2 * pr], s));
          c->spec.pair = pr;
          spec_now = true;
        }
        if (spec_now) EFA_HIP(hipEventSynchronize(c->ev[2 + 2 * c->spec.pair]));  // (not the stream: the transform behind it is to run while the host goes on)
        else EFA_HIP(hipStreamSynchronize(s));
        c->spec.launched = false;
        if (spec_now && c->timing) harvest_state_pair(c, 1 - c->spec.pair);  // the previous cycle's interval: complete by now
        if (st[0] == 0 && st[1] == 0) {
          done = true;
          any_pipeline = true;
          traj_kind_band = (kind == 4);
          c->phase_a_kind = kind;
          diag_on_host = direct;
          c->spec.launched = spec_now;
        } else {
          if (direct) {
            EFA_HIP(hipMemcpyAsync(Yw, Yp_dev, (size_t)P * M * sizeof(double), hipMemcpyDeviceToDevice, s));
            EFA_HIP(hipMemcpyAsync(ymw, ym_dev, dP, hipMemcpyDeviceToDevice, s));
            if (carry_T) EFA_HIP(launch_set_identity(M, Yw + (size_t)P * M, ymw + P, s));
          } else {
            EFA_TRY(stage_in());
          }
          if (st[2] == 0) break;  // not the Gram guard, so a spin expired: straight to the per-batch kernels
          if (attempt == 0) {
            EFA_HIP(launch_fill_u64(c->traj.as<unsigned long long>() + (size_t)w0 * TS, (size_t)Pw * TS, kTrajSentinel, s));
            EFA_HIP(hipMemsetAsync(c->status.p, 0, 3 * sizeof(int), s));
          }
        }
      }
      if (done && !direct) {
        // window rows and transform rows back into the block, then every other row of the block takes the window's records
        EFA_HIP(hipMemcpyAsync(Yw + (size_t)w0 * M, Wy, (size_t)Pw * M * sizeof(double), hipMemcpyDeviceToDevice, s));
        EFA_HIP(hipMemcpyAsync(ymw + w0, Wm, (size_t)Pw * sizeof(double), hipMemcpyDeviceToDevice, s));
        if (extra) {
          EFA_HIP(hipMemcpyAsync(Yw + (size_t)P * M, Wy + (size_t)Pw * M, (size_t)extra * M * sizeof(double), hipMemcpyDeviceToDevice, s));
          EFA_HIP(hipMemcpyAsync(ymw + P, Wm + Pw, (size_t)extra * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        const long TSk = traj_kind_band ? TS_band : TS_std;
        const double* yebase = reinterpret_cast<const double*>(c->traj.p);
        if (extra) {  // unlocalised: rows [0, w0) and [w1, P) through the window's transform, in place
          for (int part = 0; part < 2; ++part) {
            const long lo = part ? w1 : 0, hi = part ? P : w0;
            if (hi <= lo) continue;
            TransformArgs t{};
            t.Xin = Yw + (size_t)lo * M;
            t.xin = ymw + lo;
            t.Xout = Yw + (size_t)lo * M;
            t.xout = ymw + lo;
            t.nrows = hi - lo;
            t.M = M;
            t.T = Wy + (size_t)(Pw + extra) * M;
            t.w = Wm + Pw + extra;
            t.fused_members = 0;
            EFA_HIP(launch_transform(t, s));
          }
        }
        for (long b0 = w0; !extra && b0 < w1; b0 += B) {
          const int nb = (int)((w1 - b0 < B) ? (w1 - b0) : B);
          long act = 0;
          for (int k = 0; k < nb; ++k) act += ob_assim[b0 + k] ? 1 : 0;
          if (act == 0) continue;
          EFA_TRY(sweep_rows(b0, nb, yebase + (size_t)b0 * TSk, TSk, w0, w1, P));  // rows [0, P) but the window's own
        }
      }
    }
    if (!done) {
      if (any_pipeline) {
        // records of this window must look like the pipeline's (one stride for Phase B): the per-batch kernels write
        // dense ye rows, which are copied into the records' layout afterwards
        EFA_TRY(batch_window(w0, w1));
        const long TSk = traj_kind_band ? TS_band : TS_std;
        EFA_HIP(hipMemsetAsync(reinterpret_cast<double*>(c->traj.p) + (size_t)w0 * TSk, 0, (size_t)Pw * TSk * sizeof(double), s));
        EFA_HIP(hipMemcpy2DAsync(reinterpret_cast<double*>(c->traj.p) + (size_t)w0 * TSk, (size_t)TSk * sizeof(double),
                                 c->Ye_rec.as<double>() + (size_t)w0 * M, (size_t)M * sizeof(double), (size_t)M * sizeof(double),
                                 (size_t)Pw, hipMemcpyDeviceToDevice, s));
      } else if (w == 0) {
        // nothing has run as a pipeline: the whole call goes to the per-batch kernels
        EFA_TRY(batch_window(0, P));
        any_batch = true;
        break;
      } else {
        return fail(EFA_ERR_UNSUPPORTED, "internal: mixed Phase-A layouts");
      }
    }
  }
  if (any_batch || !any_pipeline) {
    c->ye_ptr = c->Ye_rec.as<double>();
    c->ye_stride = M;
    c->phase_a_kind = 2;
  } else {
    c->ye_ptr = reinterpret_cast<const double*>(c->traj.p);
    c->ye_stride = traj_kind_band ? TS_band : TS_std;
  }
  if (!c->spec.armed || c->spec.obs_out) {
    EFA_HIP(hipMemcpyAsync(Yp_dev, Yw, (size_t)P * M * sizeof(double), hipMemcpyDeviceToDevice, s));
    EFA_HIP(hipMemcpyAsync(ym_dev, ymw, dP, hipMemcpyDeviceToDevice, s));
  }
  if (c->timing && !c->spec.launched) {  // (behind a speculative transform the interval ended at the event in front of it)
    EFA_HIP(hipEventRecord(c->ev[1], s));
    c->obs_end_ev = 1;
  }

  // diagnostics back to the caller (ensrf.py:66,70,75,146-149)
  if (!diag_on_host) {
    EFA_HIP(hipMemcpyAsync(c->pin_out.p, c->out_pack.p, 4 * oslot + (size_t)P, hipMemcpyDeviceToHost, s));
    EFA_HIP(hipStreamSynchronize(s));
  }
  {
    const char* hb = static_cast<const char*>(c->pin_out.p);
    if (prior_mean) std::memcpy(prior_mean, hb, dP);
    if (prior_var) std::memcpy(prior_var, hb + oslot, dP);
    const double* pm = reinterpret_cast<const double*>(hb + 2 * oslot);
    const double* pv = reinterpret_cast<const double*>(hb + 3 * oslot);
    const uint8_t* as = reinterpret_cast<const uint8_t*>(hb + 4 * oslot);
    for (long k = 0; k < P; ++k) {
      if (assimilated) assimilated[k] = as[k];
      if (as[k]) {
        if (post_mean) post_mean[k] = pm[k];
        if (post_var) post_var[k] = pv[k];
      }
    }
  }
  if (c->timing) c->obs_ms_pending = true;  // read in efa_last_timing: the copies back to the caller's block may still be in flight
  c->have_transform = carry_T;
  c->have_traj = true;
  return EFA_OK;
}

// path "auto": one transform pass or sweep passes?  By FLOPS one transform pass is M/2 observations of sweep arithmetic (the rule of
// rounds 1-2), but the transform runs on the matrix cores at 49 TFLOP/s and the sweep on the vector ALUs at 10-20, and in MEMBER form
// (prior members in, posterior members out) the sweep path is three passes over the state -- form the perturbations, sweep, rebuild
// the members -- where the transform is one.  Measured at 1e7 x 100 (profiles/r03_auto_path.txt): member form 8 obs 10.6 ms by
// sweeps, 4.1 by the transform (48 obs: 16.9 vs 4.1); perturbation form 8 / 16 obs per sweep launch 3.4 / 4.5 ms vs 4.5.
// Above 136 members the transform re-reads the state once per group of 64 output columns: the flops rule stays.
bool auto_transform(int M, long n_active, bool member_form) {
  if (n_active <= 0) return false;
  if (M > 136) return n_active > M / 2;
  if (member_form) return true;
  return n_active > M / 8;
}
bool want_transform(const efa_ctx* c, bool member_form) {
  if (!c->have_transform) return false;
  if (c->path == EFA_PATH_TRANSFORM) return true;
  if (c->path == EFA_PATH_SWEEP) return false;
  return auto_transform(c->M, c->n_active, member_form);
}

int prepare_grid(efa_ctx* c, const double* grid_lat, const double* grid_lon, long ncol, long n_lead, long rows) {
  if (c->loc_mode != EFA_LOC_GC) return EFA_OK;
  if (!grid_lat || !grid_lon) return fail(EFA_ERR_INVALID, "GC localisation needs grid_lat/grid_lon");
  if (ncol <= 0 || n_lead <= 0 || ncol * n_lead != rows)
    return fail(EFA_ERR_INVALID, "rows=%ld must equal n_lead*ncol = %ld*%ld", rows, n_lead, ncol);
  if (c->grid_ready) {  // (the fused cycle did this before Phase A, while the device was still busy with the previous cycle)
    c->grid_ready = false;
    return EFA_OK;
  }
  EFA_TRY(h2d(c, c->glat, grid_lat, (size_t)ncol * sizeof(double)));
  EFA_TRY(h2d(c, c->glon, grid_lon, (size_t)ncol * sizeof(double)));
  c->grid_ncol = -1;
  c->grid_serial++;
  EFA_HIP(hipStreamSynchronize(c->stream));  // caller may reuse grid_lat/grid_lon on return
  return EFA_OK;
}

// The same grid ahead of Phase A (efa_ensrf_cycle_dev): compared with a pinned mirror of what the device holds and copied -- from
// the mirror, asynchronously -- only if it differs.  Cycle after cycle on one grid nothing is copied; the comparison (4 MB at
// configs[3]) is host time spent while the device still works on the previous cycle.
int prepare_grid_early(efa_ctx* c, int loc_mode, const double* grid_lat, const double* grid_lon, long ncol, long n_lead, long rows) {
  c->grid_ready = false;
  if (loc_mode != EFA_LOC_GC || rows <= 0) return EFA_OK;
  if (!grid_lat || !grid_lon) return fail(EFA_ERR_INVALID, "GC localisation needs grid_lat/grid_lon");
  if (ncol <= 0 || n_lead <= 0 || ncol * n_lead != rows)
    return fail(EFA_ERR_INVALID, "rows=%ld must equal n_lead*ncol = %ld*%ld", rows, n_lead, ncol);
  const size_t nb = (size_t)ncol * sizeof(double);
  const void* pin_before = c->pin_grid.p;
  EFA_TRY(c->pin_grid.reserve(2 * nb));
  char* pin = static_cast<char*>(c->pin_grid.p);
  const void *dl = c->glat.p, *dn = c->glon.p;
  EFA_TRY(c->glat.reserve(nb));
  EFA_TRY(c->glon.reserve(nb));
  const bool same = c->grid_ncol == ncol && pin_before == c->pin_grid.p && dl == c->glat.p && dn == c->glon.p &&
                    std::memcmp(pin, grid_lat, nb) == 0 && std::memcmp(pin + nb, grid_lon, nb) == 0;
  if (!same) {
    EFA_HIP(hipStreamSynchronize(c->stream));  // (an earlier copy out of the mirror may be in flight; a new grid is the rare case)
    std::memcpy(pin, grid_lat, nb);
    std::memcpy(pin + nb, grid_lon, nb);
    EFA_HIP(hipMemcpyAsync(c->glat.p, pin, nb, hipMemcpyHostToDevice, c->stream));
    EFA_HIP(hipMemcpyAsync(c->glon.p, pin + nb, nb, hipMemcpyHostToDevice, c->stream));
    c->grid_ncol = ncol;
    c->grid_serial++;
  }
  c->grid_ready = true;
  return EFA_OK;
}

int read_gc_pairs(efa_ctx* c) {
  if (!c->gc_pairs_pending) return EFA_OK;
  c->gc_pairs_pending = false;
  unsigned long long h_pairs = 0;
  EFA_HIP(hipMemcpyAsync(&h_pairs, c->gc_pairs.p, sizeof(h_pairs), hipMemcpyDeviceToHost, c->stream));
  EFA_HIP(hipStreamSynchronize(c->stream));
  c->gc_active_pairs = (long)h_pairs;
  return EFA_OK;
}

// ---- Phase B, localised, one pass (efa_gcsweep.hip) --------------------------------------
int state_gc_onepass(efa_ctx* c, long rows, const double* xm_in, const double* Xp_in, double* xm_out, double* Xp_out,
                     long ncol, long n_lead, int fused_members) {
  const int M = c->M;
  const long P = c->P;
  hipStream_t s = c->stream;
  const long nblk = gc_num_blocks(ncol);
  EFA_TRY(c->gc_cnt.reserve((size_t)nblk * sizeof(int)));
  EFA_TRY(c->gc_ub.reserve((size_t)nblk * sizeof(int)));
  EFA_TRY(c->gc_order.reserve((size_t)nblk * sizeof(int)));
  EFA_TRY(c->gc_obtrig.reserve((size_t)P * 6 * sizeof(double)));
  EFA_TRY(c->gc_off.reserve((size_t)(nblk + 1) * sizeof(long)));
  EFA_TRY(c->gc_pairs.reserve(sizeof(unsigned long long)));
  const void* ptrs[5] = {c->gc_off.p, c->gc_cnt.p, c->gc_order.p, c->gc_idx.p, c->gc_wts.p};
  const bool lists_ok = c->geometry_reuse && c->gc_list_valid && c->gc_list_geo == c->geo_serial && c->gc_list_grid == c->grid_serial &&
                        c->gc_list_ncol == ncol && c->gc_list_P == P && std::memcmp(ptrs, c->gc_list_ptrs, sizeof(ptrs)) == 0;
  if (!lists_ok) {
  c->gc_list_valid = false;
  EFA_TRY(read_gc_pairs(c));  // (the previous sweep's count, before the counter is cleared: that sweep is long done)
  EFA_HIP(hipMemsetAsync(c->gc_pairs.p, 0, sizeof(unsigned long long), s));
  EFA_HIP(launch_gc_bound(ncol, P, c->glat.as<double>(), c->ob_lat.as<double>(), c->ob_hw.as<double>(),
                          c->coef.as<double>(), c->gc_ub.as<int>(), c->gc_off.as<long>(), s));
  long cap = 0;  // the only host round trip of the build: 8 bytes, the capacity the lists need
  EFA_HIP(hipMemcpyAsync(&cap, c->gc_off.as<long>() + nblk, sizeof(long), hipMemcpyDeviceToHost, s));
  EFA_HIP(hipStreamSynchronize(s));
  EFA_TRY(c->gc_idx.reserve((size_t)(cap ? cap : 1) * sizeof(int)));
  EFA_TRY(c->gc_wts.reserve((size_t)(cap ? cap : 1) * 16 * sizeof(double)));
  EFA_HIP(launch_gc_fill(ncol, P, c->glat.as<double>(), c->glon.as<double>(), c->ob_lat.as<double>(),
                         c->ob_lon.as<double>(), c->ob_hw.as<double>(), c->coef.as<double>(), c->gc_obtrig.as<double>(),
                         c->gc_off.as<long>(), c->gc_cnt.as<int>(), c->gc_idx.as<int>(), c->gc_wts.as<double>(), c->gc_order.as<int>(),
                         c->gc_pairs.as<unsigned long long>(), s));
  c->gc_list_valid = true;
  c->gc_list_geo = c->geo_serial;
  c->gc_list_grid = c->grid_serial;
  c->gc_list_ncol = ncol;
  c->gc_list_P = P;
  c->gc_list_ptrs[0] = c->gc_off.p;
  c->gc_list_ptrs[1] = c->gc_cnt.p;
  c->gc_list_ptrs[2] = c->gc_order.p;
  c->gc_list_ptrs[3] = c->gc_idx.p;
  c->gc_list_ptrs[4] = c->gc_wts.p;
  c->gc_pairs_pending = true;  // read by read_gc_pairs when somebody asks (option "gc_active_pairs") or before the next build
  }
  GcSweepArgs g{};
  g.ncol = ncol;
  g.n_lead = n_lead;
  g.M = M;
  g.nblk = nblk;
  g.off = c->gc_off.as<long>();
  g.cnt = c->gc_cnt.as<int>();
  g.order = c->gc_order.as<int>();
  g.idx = c->gc_idx.as<int>();
  g.wts = c->gc_wts.as<double>();
  g.coef = c->coef.as<double>();
  g.Ye = c->ye_ptr;
  g.ye_stride = c->ye_stride;
  g.Xin = Xp_in;
  g.xin = xm_in;
  g.Xout = Xp_out;
  g.xout = xm_out;
  g.fused_members = fused_members;
  EFA_HIP(launch_sweep_gc(g, s));
  c->state_launches++;
  (void)rows;
  return EFA_OK;
}

// ---- Phase B (perturbation form) ------------------------------------------
int state_sweeps(efa_ctx* c, long rows, const double* xm_in, const double* Xp_in, double* xm_out, double* Xp_out,
                 long ncol) {
  const int M = c->M;
  const long P = c->P;
  hipStream_t s = c->stream;
  if (c->loc_mode == EFA_LOC_GC && c->gc_onepass && c->n_active > 0)  // every ensemble size the library accepts (2..256)
    return state_gc_onepass(c, rows, xm_in, Xp_in, xm_out, Xp_out, ncol, rows / ncol, 0);
  const long B = effective_batch(c, M);
  bool first = true;
  for (long b0 = 0; b0 < P; b0 += B) {
    const int nb = (int)((P - b0 < B) ? (P - b0) : B);
    long act = 0;
    for (int k = 0; k < nb; ++k) act += c->h_assim[b0 + k] ? 1 : 0;
    if (act == 0) continue;
    SweepArgs a{};
    a.Xin = first ? Xp_in : Xp_out;
    a.xin = first ? xm_in : xm_out;
    a.Xout = Xp_out;
    a.xout = xm_out;
    a.nrows = rows;
    a.M = M;
    a.Ye = c->ye_ptr + (size_t)b0 * c->ye_stride;
    a.ye_stride = c->ye_stride;
    a.coef = c->coef.as<double>() + (size_t)b0 * kCoefStride;
    a.nb = nb;
    a.skip_lo = a.skip_hi = -1;
    if (c->loc_mode == EFA_LOC_GC) {
      EFA_TRY(c->W.reserve((size_t)B * ncol * sizeof(double)));
      EFA_HIP(launch_taper_table(ncol, nb, c->glat.as<double>(), c->glon.as<double>(),
                                 c->ob_lat.as<double>() + b0, c->ob_lon.as<double>() + b0,
                                 c->ob_hw.as<double>() + b0, c->W.as<double>(), s));
      a.taper_mode = kTaperTable;
      a.W = c->W.as<double>();
      a.ncol = ncol;
    } else {
      a.taper_mode = kTaperNone;
    }
    EFA_HIP(launch_sweep(a, s));
    c->state_launches++;
    first = false;
  }
  if (first && Xp_out != Xp_in) {  // nothing assimilated: posterior == prior
    EFA_HIP(hipMemcpyAsync(Xp_out, Xp_in, (size_t)rows * M * sizeof(double), hipMemcpyDeviceToDevice, s));
    EFA_HIP(hipMemcpyAsync(xm_out, xm_in, (size_t)rows * sizeof(double), hipMemcpyDeviceToDevice, s));
  }
  return EFA_OK;
}

int state_phase(efa_ctx* c, long rows, int M, const double* xm_in, const double* Xp_in, double* xm_out,
                double* Xp_out, const double* grid_lat, const double* grid_lon, long ncol, long n_lead) {
  if (!c->have_traj) return fail(EFA_ERR_INVALID, "efa_state_phase_dev called before efa_obs_phase_dev");
  if (M != c->M) return fail(EFA_ERR_INVALID, "M=%d differs from the obs phase's M=%d", M, c->M);
  if (rows < 0) return fail(EFA_ERR_INVALID, "negative row count");
  harvest_state_ms(c);
  c->state_ms = 0.0;
  c->state_launches = 0;
  c->path_taken = EFA_PATH_SWEEP;
  if (rows == 0) return EFA_OK;
  if (!xm_in || !Xp_in || !xm_out || !Xp_out) return fail(EFA_ERR_INVALID, "null state pointer");
  EFA_TRY(prepare_grid(c, grid_lat, grid_lon, ncol, n_lead, rows));
  hipStream_t s = c->stream;
  if (c->timing) EFA_HIP(hipEventRecord(c->ev[2], s));
  if (c->P > 0 && c->n_active > 0 && want_transform(c, false)) {
    TransformArgs t{};
    t.Xin = Xp_in;
    t.xin = xm_in;
    t.Xout = Xp_out;
    t.xout = xm_out;
    t.nrows = rows;
    t.M = M;
    t.T = c->Yw.as<double>() + (size_t)c->P * M;
    t.w = c->ymw.as<double>() + c->P;
    t.fused_members = 0;
    EFA_HIP(launch_transform(t, s));
    c->state_launches = 1;
    c->path_taken = EFA_PATH_TRANSFORM;
  } else {
    EFA_TRY(state_sweeps(c, rows, xm_in, Xp_in, xm_out, Xp_out, ncol));
  }
  EFA_TRY(finish_state_timing(c, s));
  return EFA_OK;
}

}  // namespace

// ===========================================================================
// ---- RCCL, bound at run time: a single-GPU caller never loads it ---------------------------------------
namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int rccl_load() {
  if (g_rccl.lib) return EFA_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return fail(EFA_ERR_UNSUPPORTED, "librccl could not be opened: %s", dlerror());
  RcclApi a;
  a.lib = h;
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy || !a.GetErrorString) {
    dlclose(h);
    return fail(EFA_ERR_UNSUPPORTED, "librccl lacks an expected symbol");
  }
  g_rccl = a;
  return EFA_OK;
}
#define EFA_RCCL(expr)                                                                                       \
  do {                                                                                                       \
    ncclResult_t _r = (expr);                                                                                \
    if (_r != ncclSuccess) return fail(EFA_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(_r));      \
  } while (0)
}  // namespace

extern "C" {

int efa_abi_version(void) { return EFA_ABI_VERSION; }

const char* efa_last_error(void) { return g_last_error.c_str(); }

int efa_device_count(int* count) {
  if (!count) return fail(EFA_ERR_INVALID, "null count");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *count = n;
  return EFA_OK;
}

int efa_ctx_create(int device_id, efa_ctx** out) {
  if (!out) return fail(EFA_ERR_INVALID, "null out pointer");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(EFA_ERR_NO_DEVICE,
                "no HIP device visible (%s): libefa_hip has no CPU fallback and needs an MI355X (gfx950)",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  }
  if (device_id < 0 || device_id >= n) return fail(EFA_ERR_INVALID, "device_id %d out of range [0,%d)", device_id, n);
  hipDeviceProp_t prop;
  EFA_HIP(hipGetDeviceProperties(&prop, device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(EFA_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", device_id,
                prop.gcnArchName);
  EFA_HIP(hipSetDevice(device_id));
  efa_ctx* c = new (std::nothrow) efa_ctx();
  if (!c) return fail(EFA_ERR_INVALID, "out of host memory");
  c->device = device_id;
  c->cu_count = prop.multiProcessorCount;
  hipError_t es = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  if (es != hipSuccess) {
    delete c;
    return fail(EFA_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(es));
  }
  c->stream = c->own_stream;
  for (int i = 0; i < 7; ++i) {
    hipError_t ee = hipEventCreate(&c->ev[i]);
    if (ee != hipSuccess) {
      efa_ctx_destroy(c);
      return fail(EFA_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(ee));
    }
  }
  *out = c;
  return EFA_OK;
}

int efa_ctx_destroy(efa_ctx* c) {
  if (!c) return EFA_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  c->comm = nullptr;
  c->pin_in.release();
  c->pin_out.release();
  c->pin_fs.release();
  c->pin_grid.release();
  if (c->ev_fs) (void)hipEventDestroy(c->ev_fs);
  DevBuf* bufs[] = {&c->ob_pack, &c->out_pack, &c->Ye_rec, &c->coef, &c->ob_val, &c->ob_err, &c->ob_asm, &c->ob_lat, &c->ob_lon, &c->ob_hw, &c->ob_errsq,
                    &c->d_prior_mean, &c->d_prior_var, &c->d_post_mean, &c->d_post_var, &c->d_assimilated,
                    &c->Yw, &c->ymw, &c->win_Y, &c->win_m, &c->traj, &c->tw_mat, &c->status, &c->dbg, &c->W, &c->gc_cnt, &c->gc_ub, &c->gc_order, &c->gc_obtrig, &c->gc_off, &c->gc_idx, &c->gc_wts, &c->gc_pairs, &c->glat, &c->glon, &c->xm_ws, &c->fs_idx, &c->fs_wts, &c->f_glat, &c->f_glon, &c->f_sl, &c->f_cl, &c->f_valids, &c->f_var, &c->f_time, &c->f_lat, &c->f_lon, &c->f_near, &c->f_idx, &c->f_wts, &c->f_status, &c->h_xm, &c->h_Xp, &c->h_ym, &c->h_Yp,
                    &c->gcc_lat, &c->gcc_lon, &c->gcc_oblat, &c->gcc_oblon, &c->gcc_obhw, &c->gcc_coef, &c->gcc_trig, &c->gcc_cnt, &c->gcc_pairs};
  for (DevBuf* b : bufs) b->release();
  for (int i = 0; i < 7; ++i)
    if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  if (c->dbg_stream) {
    (void)hipStreamSynchronize(c->dbg_stream);
    (void)hipStreamDestroy(c->dbg_stream);
  }
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return EFA_OK;
}

int efa_ctx_set_stream(efa_ctx* c, void* hip_stream) {
  EFA_TRY(use(c));
  // NULL is a valid handle: the device's legacy default stream (what torch uses unless told otherwise)
  c->stream = reinterpret_cast<hipStream_t>(hip_stream);
  return EFA_OK;
}

int efa_ctx_set_option(efa_ctx* c, const char* key, long value) {
  EFA_TRY(use(c));
  if (!key) return fail(EFA_ERR_INVALID, "null option key");
  if (!strcmp(key, "obs_batch")) {
    if (value < 1 || value > efa::kMaxBatch) return fail(EFA_ERR_INVALID, "obs_batch must be in [1,%d]", efa::kMaxBatch);
    c->obs_batch = value;
  } else if (!strcmp(key, "path")) {
    if (value < EFA_PATH_AUTO || value > EFA_PATH_TRANSFORM) return fail(EFA_ERR_INVALID, "path must be 0,1,2");
    c->path = value;
  } else if (!strcmp(key, "timing")) {
    if (value < 0 || value > 2) return fail(EFA_ERR_INVALID, "timing must be 0, 1 or 2");
    harvest_obs_ms(c);
    harvest_state_ms(c);
    c->timing = value;
    c->state_ms_sum = c->obs_ms_sum = 0.0;
    c->state_launches_sum = 0;
  } else if (!strcmp(key, "gram")) {
    if (value < 0 || value > 2) return fail(EFA_ERR_INVALID, "gram must be 0, 1 or 2");
    c->use_gram = value;
  } else if (!strcmp(key, "pipeline")) {
    c->use_pipeline = value ? 1 : 0;
  } else if (!strcmp(key, "gc_onepass")) {
    c->gc_onepass = value ? 1 : 0;
  } else if (!strcmp(key, "geometry_reuse")) {
    c->geometry_reuse = value ? 1 : 0;
  } else if (!strcmp(key, "own_stream")) {
    c->stream = c->own_stream;  // back to the context's private non-blocking stream
  } else if (!strcmp(key, "pipe_debug")) {
    c->pipe_debug = value;
  } else if (!strcmp(key, "spin_limit")) {
    if (value < 1) return fail(EFA_ERR_INVALID, "spin_limit must be positive");
    c->spin_limit = value;
  } else if (!strcmp(key, "spin_ms")) {
    c->spin_ms = value;
  } else if (!strcmp(key, "debug_occupy_blocks")) {
    c->dbg_occupy_blocks = value;
  } else if (!strcmp(key, "debug_occupy_ms")) {
    // diagnostic: on a stream of its own, debug_occupy_blocks workgroups hold 120 KB of LDS each (one per CU, and no
    // persistent Phase-A workgroup fits beside one) for `value` ms; value 0 waits for them to finish
    if (!c->dbg_stream) EFA_HIP(hipStreamCreateWithFlags(&c->dbg_stream, hipStreamNonBlocking));
    if (value > 0) EFA_HIP(efa::launch_occupy((int)c->dbg_occupy_blocks, 120 * 1024, (double)value, c->dbg_stream));
    else EFA_HIP(hipStreamSynchronize(c->dbg_stream));
  } else if (!strcmp(key, "threads_hint")) {
  } else {
    return fail(EFA_ERR_INVALID, "unknown option '%s'", key);
  }
  return EFA_OK;
}

int efa_ctx_get_option(efa_ctx* c, const char* key, long* value) {
  EFA_TRY(use(c));
  if (!key || !value) return fail(EFA_ERR_INVALID, "null argument");
  if (!strcmp(key, "obs_batch")) *value = c->obs_batch;
  else if (!strcmp(key, "path")) *value = c->path;
  else if (!strcmp(key, "timing")) *value = c->timing;
  else if (!strcmp(key, "gram")) *value = c->use_gram;
  else if (!strcmp(key, "pipeline")) *value = c->use_pipeline;
  else if (!strcmp(key, "gc_onepass")) *value = c->gc_onepass;
  else if (!strcmp(key, "geometry_reuse")) *value = c->geometry_reuse;
  else if (!strcmp(key, "gc_active_pairs")) {
    EFA_TRY(read_gc_pairs(c));
    *value = c->gc_active_pairs;
  }
  else if (!strcmp(key, "spin_limit")) *value = c->spin_limit;
  else if (!strcmp(key, "spin_ms")) *value = c->spin_ms;
  else if (!strcmp(key, "cu_count")) *value = c->cu_count;
  else if (!strcmp(key, "phase_a_kind")) *value = c->phase_a_kind;
  else if (!strcmp(key, "pipe_dbg_addr")) *value = (long)reinterpret_cast<uintptr_t>(c->dbg.p);
  else if (!strcmp(key, "traj_addr")) *value = (long)reinterpret_cast<uintptr_t>(c->traj.p);  // (diagnostic tools only)
  else if (!strcmp(key, "device")) *value = c->device;
  else return fail(EFA_ERR_INVALID, "unknown option '%s'", key);
  return EFA_OK;
}

int efa_ctx_synchronize(efa_ctx* c) {
  EFA_TRY(use(c));
  EFA_HIP(hipStreamSynchronize(c->stream));
  return EFA_OK;
}

int efa_malloc(efa_ctx* c, size_t bytes, void** dev_out) {
  EFA_TRY(use(c));
  if (!dev_out) return fail(EFA_ERR_INVALID, "null out pointer");
  *dev_out = nullptr;
  EFA_HIP(hipMalloc(dev_out, bytes ? bytes : 8));
  return EFA_OK;
}

int efa_free(efa_ctx* c, void* dev) {
  EFA_TRY(use(c));
  if (dev) EFA_HIP(hipFree(dev));
  return EFA_OK;
}

int efa_memcpy_h2d(efa_ctx* c, void* dst_dev, const void* src, size_t bytes) {
  EFA_TRY(use(c));
  if (bytes) {
    EFA_HIP(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, c->stream));
    EFA_HIP(hipStreamSynchronize(c->stream));
  }
  return EFA_OK;
}

int efa_memcpy_d2h(efa_ctx* c, void* dst, const void* src_dev, size_t bytes) {
  EFA_TRY(use(c));
  if (bytes) {
    EFA_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    EFA_HIP(hipStreamSynchronize(c->stream));
  }
  return EFA_OK;
}

int efa_memcpy_d2d(efa_ctx* c, void* dst_dev, const void* src_dev, size_t bytes) {
  EFA_TRY(use(c));
  if (bytes) EFA_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, c->stream));
  return EFA_OK;
}

int efa_form_perts_dev(efa_ctx* c, long rows, int M, const double* X_dev, double scale, double* xm_dev,
                       double* Xp_dev) {
  EFA_TRY(use(c));
  if (rows < 0 || M < 1 || M > efa::kMaxMembers) return fail(EFA_ERR_INVALID, "bad shape rows=%ld M=%d", rows, M);
  if (rows && (!X_dev || !xm_dev || !Xp_dev)) return fail(EFA_ERR_INVALID, "null pointer");
  EFA_HIP(efa::launch_form_perts(rows, M, X_dev, scale, xm_dev, Xp_dev, c->stream));
  return EFA_OK;
}

int efa_posterior_dev(efa_ctx* c, long rows, int M, const double* xm_dev, const double* Xp_dev, double* post_dev) {
  EFA_TRY(use(c));
  if (rows < 0 || M < 1) return fail(EFA_ERR_INVALID, "bad shape rows=%ld M=%d", rows, M);
  if (rows && (!xm_dev || !Xp_dev || !post_dev)) return fail(EFA_ERR_INVALID, "null pointer");
  EFA_HIP(efa::launch_posterior(rows, M, xm_dev, Xp_dev, post_dev, c->stream));
  return EFA_OK;
}

int efa_forward_stencil_dev(efa_ctx* c, long rows, long row_offset, int M, const double* X_dev, long P, int npt,
                            const int64_t* idx, const double* wts, double* HX_dev) {
  EFA_TRY(use(c));
  if (rows < 0 || M < 1 || P < 0 || npt < 1) return fail(EFA_ERR_INVALID, "bad shape");
  if (P == 0) return EFA_OK;
  if (!X_dev || !idx || !wts || !HX_dev) return fail(EFA_ERR_INVALID, "null pointer");
  // staging of the stencil in grow-only context buffers (a hipMalloc/hipFree pair per call costs more than the kernel)
  // The caller's arrays are copied into pinned memory (free to be reused on return) and go to the device as ONE
  // asynchronous copy: no stream synchronisation here.  The pinned image is reused by the next call, which first waits
  // for this copy's event (long complete by then).
  const size_t n = (size_t)P * npt;
  const size_t half = (n * sizeof(int64_t) + 255) & ~(size_t)255;
  EFA_TRY(c->fs_idx.reserve(2 * half));
  EFA_TRY(c->pin_fs.reserve(2 * half));
  if (!c->ev_fs) EFA_HIP(hipEventCreateWithFlags(&c->ev_fs, hipEventDisableTiming));
  else EFA_HIP(hipEventSynchronize(c->ev_fs));
  // A fixed observing network hands over the same stencil cycle after cycle: when the pinned image still holds exactly these
  // indices and weights, the device copy made from it is current and nothing is copied (the copy itself is 8 us at 1e4 obs, but
  // a copy between two kernels idles the stream for ~10 us on either side).
  char* pin = static_cast<char*>(c->pin_fs.p);
  const bool same = c->fs_valid_n == n && c->fs_valid_dev == c->fs_idx.p && c->fs_valid_pin == c->pin_fs.p && std::memcmp(pin, idx, n * sizeof(int64_t)) == 0 &&
                    std::memcmp(pin + half, wts, n * sizeof(double)) == 0;
  if (!same) {
    std::memcpy(pin, idx, n * sizeof(int64_t));
    std::memcpy(pin + half, wts, n * sizeof(double));
    EFA_HIP(hipMemcpyAsync(c->fs_idx.p, c->pin_fs.p, half + n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    EFA_HIP(hipEventRecord(c->ev_fs, c->stream));
    c->fs_valid_n = n;
    c->fs_valid_dev = c->fs_idx.p;
    c->fs_valid_pin = c->pin_fs.p;
  }
  EFA_HIP(efa::launch_forward_stencil(rows, row_offset, M, X_dev, P, npt, c->fs_idx.as<int64_t>(),
                                      reinterpret_cast<const double*>(static_cast<const char*>(c->fs_idx.p) + half), HX_dev,
                                      c->stream));
  return EFA_OK;
}

int efa_interp_stencils(efa_ctx* c, int nvar, int nt, int ny, int nx, int latlon_1d, long n_grid, const double* grid_lat,
                        const double* grid_lon, const double* valid_times, long P, const int32_t* ob_var,
                        const double* ob_time, const double* ob_lat, const double* ob_lon, int64_t* sten_idx,
                        double* sten_wts, uint8_t* ob_status) {
  EFA_TRY(use(c));
  c->f_P = 0;
  if (nvar < 1 || nt < 1 || ny < 1 || nx < 1 || n_grid < 1 || P < 0)
    return fail(EFA_ERR_INVALID, "efa_interp_stencils: bad shape nvar=%d nt=%d ny=%d nx=%d n_grid=%ld P=%ld", nvar, nt, ny, nx, n_grid, P);
  if (!latlon_1d && n_grid != (long)ny * nx)
    return fail(EFA_ERR_INVALID, "efa_interp_stencils: 2-D lat/lon need n_grid = ny*nx = %ld, got %ld", (long)ny * nx, n_grid);
  if (P == 0) return EFA_OK;
  if (!grid_lat || !grid_lon || !valid_times || !ob_var || !ob_time || !ob_lat || !ob_lon)
    return fail(EFA_ERR_INVALID, "efa_interp_stencils: null input array");
  for (int i = 1; i < nt; ++i)
    if (!(valid_times[i] > valid_times[i - 1])) return fail(EFA_ERR_INVALID, "efa_interp_stencils: valid_times must ascend");
  const size_t dG = (size_t)n_grid * sizeof(double), dP = (size_t)P * sizeof(double);
  EFA_TRY(h2d(c, c->f_glat, grid_lat, dG));
  EFA_TRY(h2d(c, c->f_glon, grid_lon, dG));
  EFA_TRY(h2d(c, c->f_valids, valid_times, (size_t)nt * sizeof(double)));
  EFA_TRY(h2d(c, c->f_var, ob_var, (size_t)P * sizeof(int32_t)));
  EFA_TRY(h2d(c, c->f_time, ob_time, dP));
  EFA_TRY(h2d(c, c->f_lat, ob_lat, dP));
  EFA_TRY(h2d(c, c->f_lon, ob_lon, dP));
  EFA_TRY(c->f_sl.reserve(dG));
  EFA_TRY(c->f_cl.reserve(dG));
  EFA_TRY(c->f_near.reserve((size_t)P * 4 * sizeof(long)));
  EFA_TRY(c->f_idx.reserve((size_t)P * 8 * sizeof(long)));
  EFA_TRY(c->f_wts.reserve((size_t)P * 8 * sizeof(double)));
  EFA_TRY(c->f_status.reserve((size_t)P));
  efa::InterpArgs a{};
  a.P = P;
  a.nvar = nvar;
  a.nt = nt;
  a.ny = ny;
  a.nx = nx;
  a.latlon_1d = latlon_1d ? 1 : 0;
  a.n_grid = n_grid;
  a.glat = c->f_glat.as<double>();
  a.glon = c->f_glon.as<double>();
  a.sl = c->f_sl.as<double>();
  a.cl = c->f_cl.as<double>();
  a.valids = c->f_valids.as<double>();
  a.ob_var = c->f_var.as<int>();
  a.ob_time = c->f_time.as<double>();
  a.ob_lat = c->f_lat.as<double>();
  a.ob_lon = c->f_lon.as<double>();
  a.nearest = c->f_near.as<long>();
  a.sten_idx = c->f_idx.as<long>();
  a.sten_wts = c->f_wts.as<double>();
  a.status = c->f_status.as<unsigned char>();
  hipStream_t s = c->stream;
  EFA_HIP(efa::launch_interp_stencils(a, s));
  if (sten_idx) EFA_HIP(hipMemcpyAsync(sten_idx, c->f_idx.p, (size_t)P * 8 * sizeof(int64_t), hipMemcpyDeviceToHost, s));
  if (sten_wts) EFA_HIP(hipMemcpyAsync(sten_wts, c->f_wts.p, (size_t)P * 8 * sizeof(double), hipMemcpyDeviceToHost, s));
  if (ob_status) EFA_HIP(hipMemcpyAsync(ob_status, c->f_status.p, (size_t)P, hipMemcpyDeviceToHost, s));
  EFA_HIP(hipStreamSynchronize(s));  // the caller may reuse its input arrays on return
  c->f_P = P;
  return EFA_OK;
}

int efa_forward_interp_dev(efa_ctx* c, long ncol, long col_lo, long col_hi, long n_lead, int M, const double* X_dev,
                           double* HX_dev) {
  EFA_TRY(use(c));
  if (c->f_P <= 0) return fail(EFA_ERR_INVALID, "efa_forward_interp_dev called before efa_interp_stencils");
  if (ncol < 1 || col_lo < 0 || col_hi > ncol || col_lo > col_hi || n_lead < 1 || M < 1)
    return fail(EFA_ERR_INVALID, "efa_forward_interp_dev: bad shard [%ld,%ld) of %ld columns, n_lead=%ld, M=%d", col_lo, col_hi, ncol, n_lead, M);
  if (!X_dev || !HX_dev) return fail(EFA_ERR_INVALID, "null pointer");
  EFA_HIP(efa::launch_forward_cols(ncol, col_lo, col_hi, n_lead, M, X_dev, c->f_P, 8, c->f_idx.as<long>(), c->f_wts.as<double>(),
                                   HX_dev, c->stream));
  return EFA_OK;
}

int efa_obs_phase_dev(efa_ctx* c, int M, long P, double* ym_dev, double* Yp_dev, const double* ob_value,
                      const double* ob_error, const uint8_t* ob_assim, int loc_mode, const double* ob_lat,
                      const double* ob_lon, const double* ob_halfwidth_km, double* prior_mean, double* prior_var,
                      double* post_mean, double* post_var, uint8_t* assimilated) {
  EFA_TRY(use(c));
  return obs_phase(c, M, P, ym_dev, Yp_dev, ob_value, ob_error, ob_assim, loc_mode, ob_lat, ob_lon, ob_halfwidth_km,
                   prior_mean, prior_var, post_mean, post_var, assimilated);
}

int efa_state_phase_dev(efa_ctx* c, long rows, int M, const double* xm_in_dev, const double* Xp_in_dev,
                        double* xm_out_dev, double* Xp_out_dev, const double* grid_lat, const double* grid_lon,
                        long ncol, long n_lead) {
  EFA_TRY(use(c));
  return state_phase(c, rows, M, xm_in_dev, Xp_in_dev, xm_out_dev, Xp_out_dev, grid_lat, grid_lon, ncol, n_lead);
}

int efa_state_cycle_dev(efa_ctx* c, long rows, int M, const double* X_dev, double* post_dev, const double* grid_lat,
                        const double* grid_lon, long ncol, long n_lead) {
  EFA_TRY(use(c));
  if (!c->have_traj) return fail(EFA_ERR_INVALID, "efa_state_cycle_dev called before efa_obs_phase_dev");
  if (M != c->M) return fail(EFA_ERR_INVALID, "M=%d differs from the obs phase's M=%d", M, c->M);
  harvest_state_ms(c);
  c->state_ms = 0.0;
  c->state_launches = 0;
  c->path_taken = EFA_PATH_SWEEP;
  if (rows <= 0) return rows == 0 ? EFA_OK : fail(EFA_ERR_INVALID, "negative row count");
  if (!X_dev || !post_dev) return fail(EFA_ERR_INVALID, "null state pointer");
  EFA_TRY(prepare_grid(c, grid_lat, grid_lon, ncol, n_lead, rows));
  hipStream_t s = c->stream;
  if (c->timing) EFA_HIP(hipEventRecord(c->ev[2], s));
  if (c->P > 0 && c->n_active > 0 && want_transform(c, true)) {
    efa::TransformArgs t{};
    t.Xin = X_dev;
    t.Xout = post_dev;
    t.nrows = rows;
    t.M = M;
    t.T = c->Yw.as<double>() + (size_t)c->P * M;
    t.w = c->ymw.as<double>() + c->P;
    t.fused_members = 1;
    EFA_HIP(efa::launch_transform(t, s));
    c->state_launches = 1;
    c->path_taken = EFA_PATH_TRANSFORM;
  } else if (c->loc_mode == EFA_LOC_GC && c->gc_onepass && c->n_active > 0) {
    // localised: prior members -> posterior members in one read + one write of the state
    EFA_TRY(state_gc_onepass(c, rows, nullptr, X_dev, nullptr, post_dev, ncol, n_lead, 1));
  } else {
    EFA_TRY(c->xm_ws.reserve((size_t)rows * sizeof(double)));
    double* xm = c->xm_ws.as<double>();
    EFA_HIP(efa::launch_form_perts(rows, M, X_dev, 1.0, xm, post_dev, s));
    EFA_TRY(state_sweeps(c, rows, xm, post_dev, xm, post_dev, ncol));
    EFA_HIP(efa::launch_posterior(rows, M, xm, post_dev, post_dev, s));
  }
  EFA_TRY(finish_state_timing(c, s));
  return EFA_OK;
}

int efa_ensrf_update_dev(efa_ctx* c, long rows, int M, long P, double* xm_dev, double* Xp_dev, double* ym_dev,
                         double* Yp_dev, const double* ob_value, const double* ob_error, const uint8_t* ob_assim,
                         int loc_mode, const double* ob_lat, const double* ob_lon, const double* ob_halfwidth_km,
                         const double* grid_lat, const double* grid_lon, long ncol, long n_lead, double* prior_mean,
                         double* prior_var, double* post_mean, double* post_var, uint8_t* assimilated) {
  EFA_TRY(use(c));
  EFA_TRY(obs_phase(c, M, P, ym_dev, Yp_dev, ob_value, ob_error, ob_assim, loc_mode, ob_lat, ob_lon,
                    ob_halfwidth_km, prior_mean, prior_var, post_mean, post_var, assimilated));
  EFA_TRY(state_phase(c, rows, M, xm_dev, Xp_dev, xm_dev, Xp_dev, grid_lat, grid_lon, ncol, n_lead));
  EFA_HIP(hipStreamSynchronize(c->stream));
  return EFA_OK;
}

int efa_ensrf_cycle_dev(efa_ctx* c, long rows, int M, long P, const double* X_dev, double* post_dev, double* ym_dev,
                        double* Yp_dev, int obs_block_out, const double* ob_value, const double* ob_error,
                        const uint8_t* ob_assim, int loc_mode, const double* ob_lat, const double* ob_lon,
                        const double* ob_halfwidth_km, const double* grid_lat, const double* grid_lon, long ncol, long n_lead,
                        double* prior_mean, double* prior_var, double* post_mean, double* post_var, uint8_t* assimilated) {
  EFA_TRY(use(c));
  if (rows < 0) return fail(EFA_ERR_INVALID, "negative row count");
  if (rows > 0 && (!X_dev || !post_dev)) return fail(EFA_ERR_INVALID, "null state pointer");
  // Phase B may go into the stream before Phase A's status is known only if a wrong guess cannot cost the prior:
  // separate prior and posterior buffers (a redone Phase A needs the transform run again on the untouched prior)
  const char* xb = reinterpret_cast<const char*>(X_dev);
  const char* pb = reinterpret_cast<const char*>(post_dev);
  const size_t bytes = (size_t)rows * (size_t)(M > 0 ? M : 0) * sizeof(double);
  const bool disjoint = rows > 0 && (xb + bytes <= pb || pb + bytes <= xb);
  c->spec = efa_ctx::Spec{};
  c->spec.armed = true;
  c->spec.obs_out = obs_block_out != 0;
  c->spec.X = X_dev;
  c->spec.post = post_dev;
  c->spec.rows = (disjoint && loc_mode == EFA_LOC_NONE) ? rows : 0;  // 0: armed only for the optional obs-block copy
  {
    const int rg = prepare_grid_early(c, loc_mode, grid_lat, grid_lon, ncol, n_lead, rows);
    if (rg != EFA_OK) {
      c->spec = efa_ctx::Spec{};
      return rg;
    }
  }
  const int rc = obs_phase(c, M, P, ym_dev, Yp_dev, ob_value, ob_error, ob_assim, loc_mode, ob_lat, ob_lon, ob_halfwidth_km,
                           prior_mean, prior_var, post_mean, post_var, assimilated);
  const bool launched = c->spec.launched;
  const int pair = c->spec.pair;
  c->spec = efa_ctx::Spec{};
  if (rc != EFA_OK) {
    c->grid_ready = false;
    return rc;
  }
  if (launched) {  // Phase B is in the stream already, behind the launch that turned out fine
    c->state_ms = 0.0;
    c->state_launches = 1;
    c->state_launches_sum += 1;
    c->path_taken = EFA_PATH_TRANSFORM;
    if (c->timing) {
      (pair ? c->state_ms_pending2 : c->state_ms_pending) = true;
      if (c->timing == 1) harvest_state_pair(c, pair);
    }
    return EFA_OK;
  }
  return efa_state_cycle_dev(c, rows, M, X_dev, post_dev, grid_lat, grid_lon, ncol, n_lead);
}

int efa_ensrf_update(efa_ctx* c, long A, long N, int M, long P, double* xbm, double* Xbp, const double* ob_value,
                     const double* ob_error, const uint8_t* ob_assim, int loc_mode, const double* ob_lat,
                     const double* ob_lon, const double* ob_halfwidth_km, const double* grid_lat,
                     const double* grid_lon, long ncol, long n_lead, double* prior_mean, double* prior_var,
                     double* post_mean, double* post_var, uint8_t* assimilated) {
  EFA_TRY(use(c));
  if (N < 0 || P < 0 || A != N + P) return fail(EFA_ERR_INVALID, "A=%ld must equal N+P=%ld+%ld", A, N, P);
  EFA_TRY(check_common(M, P));
  if (A && (!xbm || !Xbp)) return fail(EFA_ERR_INVALID, "null xbm/Xbp");
  const size_t rowb = (size_t)M * sizeof(double);
  EFA_TRY(c->h_xm.reserve((size_t)(N ? N : 1) * sizeof(double)));
  EFA_TRY(c->h_Xp.reserve((size_t)(N ? N : 1) * rowb));
  EFA_TRY(c->h_ym.reserve((size_t)(P ? P : 1) * sizeof(double)));
  EFA_TRY(c->h_Yp.reserve((size_t)(P ? P : 1) * rowb));
  hipStream_t s = c->stream;
  if (N) {
    EFA_HIP(hipMemcpyAsync(c->h_xm.p, xbm, (size_t)N * sizeof(double), hipMemcpyHostToDevice, s));
    EFA_HIP(hipMemcpyAsync(c->h_Xp.p, Xbp, (size_t)N * rowb, hipMemcpyHostToDevice, s));
  }
  if (P) {
    EFA_HIP(hipMemcpyAsync(c->h_ym.p, xbm + N, (size_t)P * sizeof(double), hipMemcpyHostToDevice, s));
    EFA_HIP(hipMemcpyAsync(c->h_Yp.p, Xbp + (size_t)N * M, (size_t)P * rowb, hipMemcpyHostToDevice, s));
  }
  EFA_TRY(efa_ensrf_update_dev(c, N, M, P, c->h_xm.as<double>(), c->h_Xp.as<double>(), c->h_ym.as<double>(),
                               c->h_Yp.as<double>(), ob_value, ob_error, ob_assim, loc_mode, ob_lat, ob_lon,
                               ob_halfwidth_km, grid_lat, grid_lon, ncol, n_lead, prior_mean, prior_var, post_mean,
                               post_var, assimilated));
  if (N) {
    EFA_HIP(hipMemcpyAsync(xbm, c->h_xm.p, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, s));
    EFA_HIP(hipMemcpyAsync(Xbp, c->h_Xp.p, (size_t)N * rowb, hipMemcpyDeviceToHost, s));
  }
  if (P) {
    EFA_HIP(hipMemcpyAsync(xbm + N, c->h_ym.p, (size_t)P * sizeof(double), hipMemcpyDeviceToHost, s));
    EFA_HIP(hipMemcpyAsync(Xbp + (size_t)N * M, c->h_Yp.p, (size_t)P * rowb, hipMemcpyDeviceToHost, s));
  }
  EFA_HIP(hipStreamSynchronize(s));
  return EFA_OK;
}

int efa_cov_contract_f32_dev(efa_ctx* c, long N, int M, long P, const float* Xbp_f32_dev, const float* Ye_f32_dev,
                             float* C_f32_dev) {
  EFA_TRY(use(c));
  if (N < 0 || P < 0 || M < 4 || (M & 3) != 0)
    return fail(EFA_ERR_INVALID, "efa_cov_contract_f32_dev: need N,P >= 0 and M a positive multiple of 4 (M=%d)", M);
  if (N == 0 || P == 0) return EFA_OK;
  if (!Xbp_f32_dev || !Ye_f32_dev || !C_f32_dev) return fail(EFA_ERR_INVALID, "null pointer");
  if ((reinterpret_cast<uintptr_t>(Xbp_f32_dev) & 15u) || (reinterpret_cast<uintptr_t>(Ye_f32_dev) & 15u))
    return fail(EFA_ERR_INVALID, "operands must be 16-byte aligned");
  EFA_HIP(efa::launch_contract_f32(N, M, P, Xbp_f32_dev, Ye_f32_dev, C_f32_dev, c->stream));
  return EFA_OK;
}

int efa_last_timing(efa_ctx* c, double* state_ms, double* obs_ms, long* state_launches, int* path_taken) {
  if (!c) return fail(EFA_ERR_INVALID, "null context");
  harvest_obs_ms(c);
  harvest_state_ms(c);
  if (c->timing == 2) {  // deferred: the sums over the calls since the previous efa_last_timing
    if (state_ms) *state_ms = c->state_ms_sum;
    if (obs_ms) *obs_ms = c->obs_ms_sum;
    if (state_launches) *state_launches = c->state_launches_sum;
    c->state_ms_sum = c->obs_ms_sum = 0.0;
    c->state_launches_sum = 0;
  } else {
    if (state_ms) *state_ms = c->state_ms;
    if (obs_ms) *obs_ms = c->obs_ms;
    if (state_launches) *state_launches = c->state_launches;
  }
  if (path_taken) *path_taken = c->path_taken;
  return EFA_OK;
}

int efa_fill_synthetic_dev(efa_ctx* c, long rows, long row_offset, int M, uint64_t seed, double sigma,
                           double* X_dev) {
  EFA_TRY(use(c));
  if (rows < 0 || M < 1) return fail(EFA_ERR_INVALID, "bad shape");
  if (rows && !X_dev) return fail(EFA_ERR_INVALID, "null pointer");
  EFA_HIP(efa::launch_fill_synthetic(rows, row_offset, M, seed, sigma, X_dev, c->stream));
  return EFA_OK;
}

// ---- the one exchange step, owned by the library ------------------------------------------
int efa_comm_unique_id(uint8_t* id_out) {
  if (!id_out) return fail(EFA_ERR_INVALID, "null id");
  EFA_TRY(rccl_load());
  static_assert(sizeof(ncclUniqueId) == EFA_COMM_ID_BYTES, "EFA_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
  ncclUniqueId id;
  EFA_RCCL(g_rccl.GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof(id));
  return EFA_OK;
}

int efa_comm_init(efa_ctx* c, const uint8_t* id, int rank, int world) {
  EFA_TRY(use(c));
  if (!id || world < 1 || rank < 0 || rank >= world) return fail(EFA_ERR_INVALID, "bad communicator arguments (rank %d of %d)", rank, world);
  if (c->comm) return fail(EFA_ERR_INVALID, "the context already owns a communicator (efa_comm_destroy first)");
  EFA_TRY(rccl_load());
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  EFA_RCCL(g_rccl.CommInitRank(&c->comm, world, uid, rank));
  c->comm_rank = rank;
  c->comm_world = world;
  return EFA_OK;
}

int efa_comm_destroy(efa_ctx* c) {
  EFA_TRY(use(c));
  if (!c->comm) return EFA_OK;
  EFA_HIP(hipStreamSynchronize(c->stream));
  EFA_RCCL(g_rccl.CommDestroy(c->comm));
  c->comm = nullptr;
  c->comm_rank = 0;
  c->comm_world = 1;
  return EFA_OK;
}

int efa_allreduce_sum_dev(efa_ctx* c, double* buf_dev, long count) {
  EFA_TRY(use(c));
  if (count < 0 || (count && !buf_dev)) return fail(EFA_ERR_INVALID, "bad buffer");
  if (!c->comm) return fail(EFA_ERR_INVALID, "no communicator: call efa_comm_init first");
  if (count == 0) return EFA_OK;
  EFA_RCCL(g_rccl.AllReduce(buf_dev, buf_dev, (size_t)count, ncclDouble, ncclSum, c->comm, c->stream));
  return EFA_OK;
}

// ---- cost of a column block under Gaspari-Cohn localisation (the sharding plan of SURVEY.md 8e) ------------
int efa_gc_block_counts(efa_ctx* c, long ncol, const double* grid_lat, const double* grid_lon, long P, const double* ob_lat,
                        const double* ob_lon, const double* ob_halfwidth_km, const uint8_t* ob_assim, int32_t* block_count,
                        int32_t* block_pairs, uint64_t* active_pairs) {
  EFA_TRY(use(c));
  if (ncol <= 0 || P < 0) return fail(EFA_ERR_INVALID, "bad shape");
  if (!grid_lat || !grid_lon || !block_count || (P && (!ob_lat || !ob_lon || !ob_halfwidth_km || !ob_assim)))
    return fail(EFA_ERR_INVALID, "null pointer");
  const long nblk = efa::gc_num_blocks(ncol);
  hipStream_t s = c->stream;
  std::vector<double> coef((size_t)(P ? P : 1) * efa::kCoefStride, 0.0), hw((size_t)(P ? P : 1), 1.0);
  for (long k = 0; k < P; ++k) {
    const bool on = ob_assim[k] != 0;
    coef[(size_t)k * efa::kCoefStride + 3] = on ? 1.0 : 0.0;
    if (on) {
      if (!(ob_halfwidth_km[k] == ob_halfwidth_km[k]) || ob_halfwidth_km[k] == 0.0)
        return fail(EFA_ERR_INVALID, "observation %ld: localize_radius must be a non-zero number", k);
      hw[k] = ob_halfwidth_km[k];
    }
  }
  EFA_TRY(h2d(c, c->gcc_lat, grid_lat, (size_t)ncol * sizeof(double)));
  EFA_TRY(h2d(c, c->gcc_lon, grid_lon, (size_t)ncol * sizeof(double)));
  EFA_TRY(h2d(c, c->gcc_oblat, ob_lat, (size_t)P * sizeof(double)));
  EFA_TRY(h2d(c, c->gcc_oblon, ob_lon, (size_t)P * sizeof(double)));
  EFA_TRY(h2d(c, c->gcc_obhw, hw.data(), (size_t)P * sizeof(double)));
  EFA_TRY(h2d(c, c->gcc_coef, coef.data(), (size_t)P * efa::kCoefStride * sizeof(double)));
  EFA_TRY(c->gcc_trig.reserve((size_t)(P ? P : 1) * 6 * sizeof(double)));
  EFA_TRY(c->gcc_cnt.reserve((size_t)2 * nblk * sizeof(int)));  // [counts | pairs]
  EFA_TRY(c->gcc_pairs.reserve(sizeof(unsigned long long)));
  EFA_HIP(hipMemsetAsync(c->gcc_pairs.p, 0, sizeof(unsigned long long), s));
  EFA_HIP(hipMemsetAsync(c->gcc_cnt.p, 0, (size_t)2 * nblk * sizeof(int), s));
  if (P > 0)
    EFA_HIP(efa::launch_gc_count(ncol, P, c->gcc_lat.as<double>(), c->gcc_lon.as<double>(), c->gcc_oblat.as<double>(),
                                 c->gcc_oblon.as<double>(), c->gcc_obhw.as<double>(), c->gcc_coef.as<double>(),
                                 c->gcc_trig.as<double>(), c->gcc_cnt.as<int>(), c->gcc_cnt.as<int>() + nblk,
                                 c->gcc_pairs.as<unsigned long long>(), s));
  unsigned long long pairs = 0;
  EFA_HIP(hipMemcpyAsync(block_count, c->gcc_cnt.p, (size_t)nblk * sizeof(int), hipMemcpyDeviceToHost, s));
  if (block_pairs)
    EFA_HIP(hipMemcpyAsync(block_pairs, c->gcc_cnt.as<int>() + nblk, (size_t)nblk * sizeof(int), hipMemcpyDeviceToHost, s));
  EFA_HIP(hipMemcpyAsync(&pairs, c->gcc_pairs.p, sizeof(pairs), hipMemcpyDeviceToHost, s));
  EFA_HIP(hipStreamSynchronize(s));
  if (active_pairs) *active_pairs = pairs;
  return EFA_OK;
}

}  // extern "C"
