"""Multi-GPU EnSRF: state sharded by grid point, one process per GPU.

The reference has no working multi-process path; its dead sketch
(efa_xray/assimilation/assimilation.py:186-222, efa_xray/state/ensemble.py:59-107)
shows the intent this module implements: split the state by location into
contiguous equal chunks (last chunk takes the remainder, ensemble.py:98-106),
compute the obs-space priors once and give them to every worker
(assimilation.py:186-193).

Rows of the state are independent given the obs-space trajectory (DESIGN.md,
"F1"), so after ONE sum all-reduce of the forward-operator output HX (P x M
doubles; each rank contributes the stencil points it owns) every rank holds the
identical obs block, runs Phase A redundantly and sweeps only its own rows.
There is no per-observation communication.  With `torch.distributed` backend
"nccl" the all-reduce is RCCL over xGMI.

The arithmetic is delegated to an *engine* with the methods of
`HipEngine`; the product engine is the HIP library.  (Tests drive the same host
logic on CPU ranks with gloo by passing their own engine.)
"""
import numpy as np

from efa_xray_amd import _lib


def column_bounds(ncol, world_size):
    """[(lo, hi)] per rank: contiguous equal chunks of the (y, x) columns, the
    last rank takes the remainder (ensemble.py:98-106)."""
    chunk = ncol // world_size
    if chunk == 0:
        raise ValueError("more ranks (%d) than columns (%d)" % (world_size, ncol))
    return [(r * chunk, (r + 1) * chunk if r != world_size - 1 else ncol) for r in range(world_size)]


def shard_rows(n_lead, ncol, lo, hi):
    """Global state-vector rows (order of to_vect(): lead-major, then column)
    owned by the column shard [lo, hi)."""
    return (np.arange(n_lead, dtype=np.int64)[:, None] * ncol + np.arange(lo, hi, dtype=np.int64)[None, :]).reshape(-1)


def localize_stencil(idx, wts, n_lead, ncol, lo, hi):
    """Map global stencil rows to the shard's local rows; points owned by other
    shards get weight 0 (they are added by the all-reduce)."""
    idx = np.asarray(idx, dtype=np.int64)
    wts = np.asarray(wts, dtype=np.float64)
    lead = idx // ncol
    col = idx % ncol
    own = (col >= lo) & (col < hi) & (lead < n_lead)
    lidx = np.where(own, lead * (hi - lo) + (col - lo), 0)
    return np.ascontiguousarray(lidx), np.ascontiguousarray(np.where(own, wts, 0.0))


class HipEngine(object):
    """The product engine: libefa_hip on this rank's GPU, buffers as torch CUDA
    tensors (torch is plumbing: allocation, stream, RCCL)."""

    def __init__(self, device):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(device)
        self.ctx = _lib.Context(device)
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def empty(self, shape):
        return self.torch.empty(shape, dtype=self.torch.float64, device=self.device)

    def forward_stencil(self, rows, M, X, lidx, lwts, HX):
        self.ctx.forward_stencil(rows, 0, M, X.data_ptr(), lidx, lwts, HX.data_ptr())

    def form_perts(self, rows, M, X, xm, Xp):
        self.ctx.form_perts(rows, M, X.data_ptr(), xm.data_ptr(), Xp.data_ptr())

    def inflate(self, rows, M, X, factor):
        """X <- mean + factor * (X - mean), in place on the resident shard (assimilation.py:62-69)."""
        xm = self.empty((max(rows, 1),))
        self.ctx.form_perts(rows, M, X.data_ptr(), xm.data_ptr(), X.data_ptr(), scale=factor)
        self.ctx.posterior(rows, M, xm.data_ptr(), X.data_ptr(), X.data_ptr())

    def obs_phase(self, M, P, ym, Yp, ob):
        return self.ctx.obs_phase(M, P, ym.data_ptr(), Yp.data_ptr(), ob["value"], ob["error"], ob["assim"],
                                  _lib.LOC_GC if ob.get("loc") == "GC" else _lib.LOC_NONE,
                                  ob.get("lat"), ob.get("lon"), ob.get("halfwidth"))

    def state_cycle(self, rows, M, X, post, grid_lat, grid_lon, n_lead):
        self.ctx.state_cycle(rows, M, X.data_ptr(), post.data_ptr(), grid_lat, grid_lon, n_lead)


class ShardedEnSRF(object):
    """One rank's part of a sharded EnSRF cycle.

    X_local : engine buffer (rows_local x M), rows_local = n_lead * (hi - lo),
              prior member values of this shard (resident on the rank's device).
    sten_idx/sten_wts : (P, npt) linear forward operator in GLOBAL rows.
    ob : dict(value, error, assim[, loc='GC', lat, lon, halfwidth]) -- identical on all ranks.
    grid_lat/grid_lon : (ncol,) per-column lat/lon of the GLOBAL grid (GC only).
    """

    def __init__(self, engine, n_lead, ncol, M, rank=0, world_size=1, group=None):
        self.engine = engine
        self.n_lead, self.ncol, self.M = int(n_lead), int(ncol), int(M)
        self.rank, self.world_size, self.group = rank, world_size, group
        self.lo, self.hi = column_bounds(self.ncol, world_size)[rank]
        self.rows_local = self.n_lead * (self.hi - self.lo)

    def local_rows(self):
        return shard_rows(self.n_lead, self.ncol, self.lo, self.hi)

    def all_reduce_sum(self, t):
        if self.world_size > 1:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def partial_estimates(self, X_local, sten_idx, sten_wts, inflation=None):
        """Stage 1: this shard's contribution to the obs-space prior ensemble HX (P x M): the
        stencil points it owns, zeros elsewhere.  `inflation` (a float) first inflates the
        resident shard in place about its ensemble mean -- the constant form of
        `Assimilation.inflate_state` (assimilation.py:62-69), which the reference applies
        before the obs priors are computed (assimilation.py:131-138)."""
        eng, M = self.engine, self.M
        if inflation is not None:
            eng.inflate(self.rows_local, M, X_local, float(inflation))
        P = int(np.asarray(sten_idx).shape[0])
        lidx, lwts = localize_stencil(sten_idx, sten_wts, self.n_lead, self.ncol, self.lo, self.hi)
        HX = eng.empty((P, M))
        eng.forward_stencil(self.rows_local, M, X_local, lidx, lwts, HX)
        return HX

    def assimilate(self, X_local, post_local, HX, ob, grid_lat=None, grid_lon=None):
        """Stage 2, after HX has been summed over the shards: obs-space priors, Phase A
        (replicated: identical on every rank) and the sweep of this shard's rows."""
        eng, M = self.engine, self.M
        P = int(HX.shape[0])
        ym = eng.empty((max(P, 1),))
        eng.form_perts(P, M, HX, ym, HX)                                   # assimilation.py:46-48
        diag = eng.obs_phase(M, P, ym, HX, ob)                             # identical on every rank
        glat = glon = None
        if ob.get("loc") == "GC":
            glat = np.ascontiguousarray(np.asarray(grid_lat, dtype=np.float64).reshape(-1)[self.lo:self.hi])
            glon = np.ascontiguousarray(np.asarray(grid_lon, dtype=np.float64).reshape(-1)[self.lo:self.hi])
        eng.state_cycle(self.rows_local, M, X_local, post_local, glat, glon, self.n_lead)
        return diag

    def update(self, X_local, post_local, sten_idx, sten_wts, ob, grid_lat=None, grid_lon=None, inflation=None):
        HX = self.partial_estimates(X_local, sten_idx, sten_wts, inflation)
        self.all_reduce_sum(HX)                                            # the one exchange step
        return self.assimilate(X_local, post_local, HX, ob, grid_lat, grid_lon)
