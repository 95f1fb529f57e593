"""N>1 host path on CPU: world_size 2 (and 3) with the gloo backend.

The product's sharding logic (efa_xray_amd.distributed.ShardedEnSRF: column
bounds, stencil localisation, the single HX all-reduce, replicated Phase A,
per-shard sweep) runs unchanged; only the arithmetic engine is replaced by a
test engine built on the CPU oracle, because the product engine needs a GPU.
Gathered shard posteriors must equal the unsharded oracle.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


class OracleEngine(object):
    """Test double for efa_xray_amd.distributed.HipEngine (CPU tensors + oracle)."""

    def __init__(self):
        from oracle import ensrf_oracle as orc
        self.orc = orc
        self.rec = None

    def empty(self, shape):
        return torch.zeros(shape, dtype=torch.float64)

    def forward_stencil(self, rows, M, X, lidx, lwts, HX):
        x = X.numpy()
        out = np.zeros((lidx.shape[0], M))
        for k in range(lidx.shape[0]):
            for j in range(lidx.shape[1]):
                if lwts[k, j] != 0.0:
                    out[k] += lwts[k, j] * x[lidx[k, j]]
        HX.copy_(torch.from_numpy(out))

    def form_perts(self, rows, M, X, xm, Xp):
        x = X.numpy().copy()
        m = x.mean(axis=1)
        xm[:rows].copy_(torch.from_numpy(m))
        Xp.copy_(torch.from_numpy(x - m[:, None]))

    def obs_phase(self, M, P, ym, Yp, ob):
        self.rec = dict(ym=ym.numpy()[:P].copy(), Yp=Yp.numpy().copy(), ob=ob)
        kw = {}
        if ob.get("loc") == "GC":   # obs-space only: no state rows, grid irrelevant
            kw = dict(loc="GC", ob_lat=ob["lat"], ob_lon=ob["lon"], ob_halfwidth=ob["halfwidth"],
                      grid_lat=np.zeros(0), grid_lon=np.zeros(0), state_shape=(1, 1, 1, 0))
        _, _, diag = self.orc.ensrf_update(self.rec["ym"], self.rec["Yp"], 0, ob["value"], ob["error"],
                                           ob["assim"], **kw)
        return diag

    def gc_block_counts(self, grid_lat, grid_lon, ob):
        return numpy_block_counts(self.orc, grid_lat, grid_lon, ob)

    def state_cycle(self, rows, M, X, post, grid_lat, grid_lon, n_lead):
        x = X.numpy()
        xm = x.mean(axis=1)
        xbm = np.hstack((xm, self.rec["ym"]))
        Xbp = np.vstack((x - xm[:, None], self.rec["Yp"]))
        ob = self.rec["ob"]
        kw = {}
        if ob.get("loc") == "GC":
            kw = dict(loc="GC", ob_lat=ob["lat"], ob_lon=ob["lon"], ob_halfwidth=ob["halfwidth"],
                      grid_lat=grid_lat, grid_lon=grid_lon, state_shape=(n_lead, 1, 1, rows // n_lead))
        xam, Xap, _ = self.orc.ensrf_update(xbm, Xbp, rows, ob["value"], ob["error"], ob["assim"], **kw)
        post.copy_(torch.from_numpy(self.orc.format_posterior_state(xam, Xap, rows)))


def numpy_block_counts(orc, grid_lat, grid_lon, ob, block=16):
    """What efa_gc_block_counts computes, with the oracle's distance and taper (ensemble.py:254-267,
    observation.py:117-130): per block of 16 columns the assimilated obs with a non-zero weight on any of them."""
    glat = np.asarray(grid_lat, dtype=np.float64).reshape(-1)
    glon = np.asarray(grid_lon, dtype=np.float64).reshape(-1)
    ncol = glat.shape[0]
    nblk = (ncol + block - 1) // block
    cnt = np.zeros(nblk, dtype=np.int64)
    blk_pairs = np.zeros(nblk, dtype=np.int64)
    for k in np.nonzero(np.asarray(ob["assim"]))[0]:
        w = orc.gaspari_cohn(orc.distance_to_point(glat, glon, ob["lat"][k], ob["lon"][k]), ob["halfwidth"][k]) != 0.0
        pad = np.zeros(nblk * block, dtype=bool)
        pad[:ncol] = w
        cnt += pad.reshape(nblk, block).any(axis=1)
        blk_pairs += pad.reshape(nblk, block).sum(axis=1)
    return cnt, blk_pairs, int(blk_pairs.sum())


def _problem(loc):
    rng = np.random.default_rng(77)
    n_lead, ny, nx, M, P = 3, 7, 9, 12, 20
    ncol = ny * nx
    N = n_lead * ncol
    X = rng.standard_normal((N, 1)) + 2.0 * rng.standard_normal((N, M))
    idx = rng.integers(0, N, (P, 4))
    wts = rng.random((P, 4))
    wts /= wts.sum(axis=1, keepdims=True)
    lat, lon = np.meshgrid(np.linspace(10, 60, ny), np.linspace(100, 180, nx), indexing="ij")
    col0 = idx[:, 0] % ncol
    ob = dict(value=rng.standard_normal(P), error=rng.uniform(0.5, 1.5, P), assim=rng.random(P) > 0.15)
    if loc:
        ob.update(loc="GC", lat=lat.reshape(-1)[col0] + 0.1, lon=lon.reshape(-1)[col0] - 0.1,
                  halfwidth=rng.uniform(1500, 4000, P))
    return dict(n_lead=n_lead, ncol=ncol, ny=ny, nx=nx, M=M, P=P, N=N, X=X, idx=idx, wts=wts, lat=lat, lon=lon, ob=ob)


def _worker(rank, world, port, loc, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from efa_xray_amd.distributed import ShardedEnSRF
        pr = _problem(loc)
        if loc == "balanced":   # cost-balanced contiguous shards (Gaspari-Cohn): cut at 16-column block boundaries
            sh = ShardedEnSRF.balanced(OracleEngine(), pr["n_lead"], pr["ncol"], pr["M"], pr["ob"], pr["lat"].reshape(-1),
                                       pr["lon"].reshape(-1), rank=rank, world_size=world)
            assert sh.bounds != ShardedEnSRF(OracleEngine(), pr["n_lead"], pr["ncol"], pr["M"], rank, world).bounds
        else:
            sh = ShardedEnSRF(OracleEngine(), pr["n_lead"], pr["ncol"], pr["M"], rank=rank, world_size=world)
        rows = sh.local_rows()
        Xl = torch.from_numpy(np.ascontiguousarray(pr["X"][rows]))
        post = torch.zeros_like(Xl)
        diag = sh.update(Xl, post, pr["idx"], pr["wts"], pr["ob"], pr["lat"].reshape(-1), pr["lon"].reshape(-1))
        # gather shard posteriors on rank 0
        gathered = [None] * world
        dist.all_gather_object(gathered, (rows, post.numpy(), diag["post_mean"]))
        if rank == 0:
            full = np.zeros((pr["N"], pr["M"]))
            for r_rows, r_post, _ in gathered:
                full[r_rows] = r_post
            q.put((full, [g[2] for g in gathered]))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,loc", [(2, False), (2, True), (3, True), (3, "balanced")])
def test_sharded_cycle_equals_unsharded_oracle(world, loc):
    from oracle import ensrf_oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, loc, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, post_means = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    pr = _problem(loc)
    HX = np.array([(pr["wts"][k][:, None] * pr["X"][pr["idx"][k]]).sum(axis=0) for k in range(pr["P"])])
    kw = {}
    if loc:
        ob = pr["ob"]
        kw = dict(loc="GC", ob_lat=ob["lat"], ob_lon=ob["lon"], ob_halfwidth=ob["halfwidth"], grid_lat=pr["lat"],
                  grid_lon=pr["lon"], state_shape=(pr["n_lead"], 1, pr["ny"], pr["nx"]))
    ref_post, _, _, diag = orc.ensrf_cycle(pr["X"], HX, pr["ob"]["value"], pr["ob"]["error"], pr["ob"]["assim"], **kw)
    np.testing.assert_allclose(full, ref_post, rtol=1e-11, atol=1e-12)
    for pm in post_means:        # Phase A is replicated: every rank reports the same diagnostics
        np.testing.assert_allclose(pm, diag["post_mean"], rtol=1e-11, equal_nan=True)


def test_column_bounds_and_stencil_localisation():
    from efa_xray_amd.distributed import column_bounds, shard_rows, localize_stencil
    assert column_bounds(10, 3) == [(0, 3), (3, 6), (6, 10)]      # last takes the remainder
    assert column_bounds(8, 1) == [(0, 8)]
    with pytest.raises(ValueError):
        column_bounds(2, 3)
    n_lead, ncol = 3, 10
    allrows = np.concatenate([shard_rows(n_lead, ncol, lo, hi) for lo, hi in column_bounds(ncol, 3)])
    assert sorted(allrows.tolist()) == list(range(n_lead * ncol))
    idx = np.array([[0, 13, 29, 7]])
    wts = np.array([[0.1, 0.2, 0.3, 0.4]])
    tot = 0.0
    for lo, hi in column_bounds(ncol, 3):
        li, lw = localize_stencil(idx, wts, n_lead, ncol, lo, hi)
        rows = shard_rows(n_lead, ncol, lo, hi)
        for j in range(4):
            if lw[0, j] != 0:
                assert rows[li[0, j]] == idx[0, j]
        tot += lw.sum()
    assert abs(tot - 1.0) < 1e-15          # every stencil point owned exactly once


def test_cost_balanced_column_bounds_on_a_global_grid():
    """Gaspari-Cohn work per column grows towards the poles of a lat/lon grid (obs drawn uniformly over grid points
    are dense per km^2 there and every footprint spans many longitudes): the equal split of ensemble.py:98-106 leaves
    the polar ranks with a multiple of the mean work, the cost-balanced contiguous split stays within 5 %."""
    from oracle import ensrf_oracle as orc
    from efa_xray_amd.distributed import balanced_column_bounds, column_bounds, GC_FIXED_COST
    ny, nx, P = 61, 120, 1500
    lat, lon = np.meshgrid(np.linspace(-90, 90, ny), np.linspace(0, 357, nx), indexing="ij")
    glat, glon = lat.reshape(-1), lon.reshape(-1)
    ncol = ny * nx
    rng = np.random.default_rng(9)
    col = rng.choice(ncol, P, replace=False)
    ob = dict(lat=glat[col], lon=glon[col], halfwidth=np.full(P, 1000.0), assim=rng.random(P) > 0.1)
    cnt, bp, pairs = numpy_block_counts(orc, glat, glon, ob)
    assert pairs > 0 and bp.max() > 2 * np.median(bp)
    cost = bp + GC_FIXED_COST

    def shares(bounds):
        return np.array([cost[lo // 16:(hi + 15) // 16].sum() for lo, hi in bounds])

    for world in (2, 3, 4, 8):
        b = balanced_column_bounds(bp, ncol, world)
        assert b[0][0] == 0 and b[-1][1] == ncol and all(x[1] == y[0] for x, y in zip(b[:-1], b[1:]))
        assert all(lo % 16 == 0 for lo, _ in b) and all(hi > lo for lo, hi in b)
        s = shares(b)
        assert s.max() / s.mean() <= 1.05, (world, s)
    eq = column_bounds(ncol, 8)
    eq = [(lo - lo % 16, hi - hi % 16 if hi != ncol else hi) for lo, hi in eq]
    s_eq = shares(eq)
    assert s_eq.max() / s_eq.mean() > 1.3       # what the equal split costs
    # degenerate inputs: fewer blocks than ranks -> the equal split; no obs -> equal blocks
    assert balanced_column_bounds(np.zeros(1), 10, 3) == column_bounds(10, 3)
    b0 = balanced_column_bounds(np.zeros(8), 128, 4)
    assert b0 == [(0, 32), (32, 64), (64, 96), (96, 128)]


class _CommEngine:
    """Stand-in for HipEngine in bench.init_library_comm: the communicator comes up on rank 0 only."""
    def __init__(self, rank):
        self.device = torch.device("cpu")
        self.has_comm = False
        self.rank = rank

    def init_comm(self, rank, world):
        if rank != 0:
            raise RuntimeError("no communicator on this rank")
        self.has_comm = True

    def all_reduce_sum(self, t):
        t.fill_(3.0)     # what a 2-rank sum of rank + 1 gives
        return t


def _comm_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench

        class _T:                      # torch with a synchronize() that works without a GPU
            def __getattr__(self, n):
                return getattr(torch, n)
            class cuda:                # noqa: E301
                @staticmethod
                def synchronize():
                    pass
        eng = _CommEngine(rank)
        how = bench.init_library_comm(eng, rank, world, dist, _T(), timeout_s=30.0)
        q.put((rank, how, eng.has_comm))
    finally:
        dist.destroy_process_group()


def test_bench_falls_back_to_torch_all_reduce_when_one_rank_has_no_library_communicator():
    """bench.py, N > 1: the ranks agree on ONE collective -- the library's own RCCL communicator only if it came up
    (and passed a probe all-reduce) on every rank, otherwise torch.distributed's all-reduce everywhere."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, how, has_comm in got:
        assert how.startswith("torch.distributed all_reduce") and not has_comm, (rank, how, has_comm)
