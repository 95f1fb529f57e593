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
There is no per-observation communication.  The all-reduce is RCCL over xGMI,
issued by the library itself on the context's stream (`efa_comm_init` /
`efa_allreduce_sum_dev`, include/efa_hip.h); `torch.distributed` is only the
rendezvous that carries the 128-byte communicator id to the other ranks (and the
barrier of the bench).

Partition.  Without localisation every column costs the same and the split is the
reference sketch's: contiguous equal chunks.  With Gaspari-Cohn localisation the
work of a column is the number of observations whose taper reaches it, which on a
lat/lon grid grows several-fold towards the poles; `balanced_column_bounds` cuts
the same contiguous order at equal cumulative cost instead
(`efa_gc_block_counts` counts the active list of every 16-column block on the device).

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


GC_BLOCK = 16          # columns per block of the one-pass localised sweep (efa_gcsweep.hip)
# A block's read + write of its rows (and its staging), in (column, ob) pairs.  Calibrated on configs[3] on one MI355X by the measured
# per-shard state phases of the 8-way split (tools/shard_balance.py): 512 (round 3's first estimate: 0.41 us of one CU per pair, 0.22 ms of
# one CU's HBM share per block) left the equatorial shards -- many blocks, short lists -- 4 % slower than the polar ones with the
# row-per-lane sweep (16.68 vs 15.95 ms); 900 tips it the other way; at 760 the eight shards take 16.03-16.33 ms (max / mean 1.011).
GC_FIXED_COST = 760.0


def balanced_column_bounds(block_pairs, ncol, world_size, block=GC_BLOCK, fixed=GC_FIXED_COST):
    """[(lo, hi)] per rank: contiguous column ranges, cut on block boundaries so that every rank gets the same
    share of sum(block_pairs + fixed).  `block_pairs[b]` = (column, observation) pairs with a non-zero taper in
    block b: the sweep's waves skip an ob that is zero on their columns, so its work follows the pairs.  Falls back
    to equal chunks when there are fewer blocks than ranks."""
    cost = np.asarray(block_pairs, dtype=np.float64) + float(fixed)
    nblk = cost.shape[0]
    assert nblk == (ncol + block - 1) // block
    if nblk < world_size:
        return column_bounds(ncol, world_size)
    cum = np.concatenate(([0.0], np.cumsum(cost)))
    cuts = [0]
    for r in range(1, world_size):
        target = cum[-1] * r / world_size
        b = int(np.searchsorted(cum, target))            # first block boundary at or beyond the target
        if b > 0 and target - cum[b - 1] < cum[b] - target:
            b -= 1                                        # the nearer of the two boundaries
        b = min(max(b, cuts[-1] + 1), nblk - (world_size - r))   # every rank keeps at least one block
        cuts.append(b)
    cuts.append(nblk)
    return [(cuts[r] * block, min(cuts[r + 1] * block, ncol)) for r in range(world_size)]


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
        self.has_comm = False
        self.fused_cycle = True     # ShardedEnSRF.assimilate: Phase A + state phase as one library call (efa_ensrf_cycle_dev)

    def init_comm(self, rank, world_size, group=None):
        """Create the library-owned RCCL communicator.  Collective over `group` (torch.distributed, any
        backend): rank 0's id travels as a broadcast object; at world size 1 nothing else is needed."""
        cid = [self.ctx.comm_unique_id() if rank == 0 else None]
        if world_size > 1:
            import torch.distributed as dist
            dist.broadcast_object_list(cid, src=0, group=group)
        self.ctx.comm_init(cid[0], rank, world_size)
        self.has_comm = True

    def all_reduce_sum(self, t):
        """Sum over the ranks through the context's own communicator, on the context's stream."""
        assert self.has_comm and t.is_contiguous()
        self.ctx.allreduce_sum(t.data_ptr(), t.numel())
        return t

    def gc_block_counts(self, grid_lat, grid_lon, ob):
        return self.ctx.gc_block_counts(grid_lat, grid_lon, ob["lat"], ob["lon"], ob["halfwidth"], ob["assim"])

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

    def cycle(self, rows, M, P, X, post, ym, Yp, ob, grid_lat, grid_lon, n_lead):
        """obs_phase + state_cycle as ONE library call (efa_ensrf_cycle_dev): no host round trip between Phase A and Phase B
        when the transform applies; the augmented obs rows are not copied back (the reference discards them)."""
        return self.ctx.ensrf_cycle(rows, M, P, X.data_ptr(), post.data_ptr(), ym.data_ptr(), Yp.data_ptr(),
                                    ob["value"], ob["error"], ob["assim"],
                                    _lib.LOC_GC if ob.get("loc") == "GC" else _lib.LOC_NONE,
                                    ob.get("lat"), ob.get("lon"), ob.get("halfwidth"), grid_lat, grid_lon, n_lead)


class ShardedEnSRF(object):
    """One rank's part of a sharded EnSRF cycle.

    X_local : engine buffer (rows_local x M), rows_local = n_lead * (hi - lo),
              prior member values of this shard (resident on the rank's device).
    sten_idx/sten_wts : (P, npt) linear forward operator in GLOBAL rows.
    ob : dict(value, error, assim[, loc='GC', lat, lon, halfwidth]) -- identical on all ranks.
    grid_lat/grid_lon : (ncol,) per-column lat/lon of the GLOBAL grid (GC only).
    """

    def __init__(self, engine, n_lead, ncol, M, rank=0, world_size=1, group=None, bounds=None):
        self.engine = engine
        self.n_lead, self.ncol, self.M = int(n_lead), int(ncol), int(M)
        self.rank, self.world_size, self.group = rank, world_size, group
        self.bounds = list(bounds) if bounds is not None else column_bounds(self.ncol, world_size)
        assert len(self.bounds) == world_size and self.bounds[0][0] == 0 and self.bounds[-1][1] == self.ncol
        assert all(a[1] == b[0] for a, b in zip(self.bounds[:-1], self.bounds[1:]))
        self.lo, self.hi = self.bounds[rank]
        self.rows_local = self.n_lead * (self.hi - self.lo)

    @classmethod
    def balanced(cls, engine, n_lead, ncol, M, ob, grid_lat, grid_lon, rank=0, world_size=1, group=None):
        """Shards of equal COST for a localised cycle: every rank counts the active lists of the global grid's
        column blocks (the same deterministic count everywhere, a few ms on the device) and cuts the contiguous
        column order at equal cumulative cost.  Without localisation: the equal split."""
        if ob.get("loc") != "GC" or world_size == 1:
            return cls(engine, n_lead, ncol, M, rank, world_size, group)
        _, blk_pairs, _ = engine.gc_block_counts(grid_lat, grid_lon, ob)
        return cls(engine, n_lead, ncol, M, rank, world_size, group,
                   bounds=balanced_column_bounds(blk_pairs, int(ncol), world_size))

    def local_rows(self):
        return shard_rows(self.n_lead, self.ncol, self.lo, self.hi)

    def all_reduce_sum(self, t):
        """The one exchange step.  An engine that owns a communicator (HipEngine after `init_comm`: RCCL inside
        libefa_hip) performs it; otherwise `torch.distributed` does (CPU ranks under gloo in the tests)."""
        if self.world_size > 1:
            if getattr(self.engine, "has_comm", False):
                self.engine.all_reduce_sum(t)
            else:
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
        # the shard-local form of a stencil is kept while the caller hands over the same stencil (cycle after cycle with a
        # fixed observing network): compared by content (two 80 KB compares at 1e4 obs), not by identity
        idx_a, wts_a = np.asarray(sten_idx), np.asarray(sten_wts)
        cached = getattr(self, "_sten_src", None)
        if cached is None or cached[0].shape != idx_a.shape or not (np.array_equal(cached[0], idx_a) and np.array_equal(cached[1], wts_a)):
            self._sten_local = localize_stencil(idx_a, wts_a, self.n_lead, self.ncol, self.lo, self.hi)
            self._sten_src = (idx_a.copy(), wts_a.copy())
        lidx, lwts = self._sten_local
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
        glat = glon = None
        if ob.get("loc") == "GC":
            glat = np.ascontiguousarray(np.asarray(grid_lat, dtype=np.float64).reshape(-1)[self.lo:self.hi])
            glon = np.ascontiguousarray(np.asarray(grid_lon, dtype=np.float64).reshape(-1)[self.lo:self.hi])
        if getattr(eng, "fused_cycle", False):                             # one library call for both phases
            return eng.cycle(self.rows_local, M, P, X_local, post_local, ym, HX, ob, glat, glon, self.n_lead)
        diag = eng.obs_phase(M, P, ym, HX, ob)                             # identical on every rank
        eng.state_cycle(self.rows_local, M, X_local, post_local, glat, glon, self.n_lead)
        return diag

    def update(self, X_local, post_local, sten_idx, sten_wts, ob, grid_lat=None, grid_lon=None, inflation=None):
        HX = self.partial_estimates(X_local, sten_idx, sten_wts, inflation)
        self.all_reduce_sum(HX)                                            # the one exchange step
        return self.assimilate(X_local, post_local, HX, ob, grid_lat, grid_lon)
