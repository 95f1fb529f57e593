"""The multi-GPU product path on ONE device: `ShardedEnSRF` + `HipEngine` (torch CUDA tensors,
libefa_hip on torch's stream) at world_size 1 against the oracle and against `EnSRF.update()`,
and as two logical column shards through two engines with an in-process sum standing in for the
RCCL all-reduce -- bit-equal to the unsharded run (rows are independent given the obs-space
trajectory; reference precedent ensemble.py:98-106, assimilation.py:186-193).
The real N>1 rendezvous is covered on CPU by tests/test_distributed_gloo.py."""
import numpy as np
import pytest

from oracle import ensrf_oracle as orc
from test_gpu_parity import assert_parity

pytestmark = pytest.mark.gpu


def _problem(loc, row_pick, seed=5):
    rng = np.random.default_rng(seed)
    n_lead, ny, nx, M, P = 3, 24, 40, 30, 90
    ncol = ny * nx
    N = n_lead * ncol
    X = rng.standard_normal((N, 1)) + 2.0 * rng.standard_normal((N, M))
    if row_pick:
        idx = rng.choice(N, P, replace=False)[:, None].astype(np.int64)
        wts = np.ones((P, 1))
    else:
        idx = rng.integers(0, N, (P, 4)).astype(np.int64)
        wts = rng.random((P, 4))
        wts /= wts.sum(axis=1, keepdims=True)
    lat, lon = np.meshgrid(np.linspace(10, 60, ny), np.linspace(100, 180, nx), indexing="ij")
    col0 = idx[:, 0] % ncol
    ob = dict(value=rng.standard_normal(P), error=rng.uniform(0.5, 1.5, P), assim=rng.random(P) > 0.1)
    if loc:
        ob.update(loc="GC", lat=lat.reshape(-1)[col0] + 0.1, lon=lon.reshape(-1)[col0] - 0.1,
                  halfwidth=rng.uniform(800, 3000, P))
    return dict(n_lead=n_lead, ncol=ncol, ny=ny, nx=nx, M=M, P=P, N=N, X=X, idx=idx, wts=wts, lat=lat, lon=lon, ob=ob)


def _oracle(pr, X=None):
    X = pr["X"] if X is None else X
    HX = np.array([(pr["wts"][k][:, None] * X[pr["idx"][k]]).sum(axis=0) for k in range(pr["P"])])
    kw = {}
    ob = pr["ob"]
    if ob.get("loc"):
        kw = dict(loc="GC", ob_lat=ob["lat"], ob_lon=ob["lon"], ob_halfwidth=ob["halfwidth"], grid_lat=pr["lat"],
                  grid_lon=pr["lon"], state_shape=(pr["n_lead"], 1, pr["ny"], pr["nx"]))
    post, _, _, diag = orc.ensrf_cycle(X, HX, ob["value"], ob["error"], ob["assim"], **kw)
    return post, diag


@pytest.mark.parametrize("loc", [False, True])
def test_sharded_ensrf_world_size_1_on_hip_engine(loc):
    import torch
    from efa_xray_amd.distributed import ShardedEnSRF, HipEngine
    pr = _problem(loc, row_pick=False)
    eng = HipEngine(0)
    sh = ShardedEnSRF(eng, pr["n_lead"], pr["ncol"], pr["M"], rank=0, world_size=1)
    X = torch.from_numpy(pr["X"]).to(eng.device)
    post = torch.empty_like(X)
    diag = sh.update(X, post, pr["idx"], pr["wts"], pr["ob"], pr["lat"].reshape(-1), pr["lon"].reshape(-1))
    torch.cuda.synchronize()
    ref_post, ref_diag = _oracle(pr)
    assert_parity(post.cpu().numpy(), ref_post, "post")
    for key in ("prior_mean", "prior_var", "post_mean", "post_var"):
        assert_parity(diag[key], ref_diag[key], key)
    assert np.array_equal(diag["assimilated"], ref_diag["assimilated"])
    assert np.array_equal(X.cpu().numpy(), pr["X"]), "the resident prior must not change"
    # constant inflation of the resident shard, then the same cycle (assimilation.py:62-69,131-138)
    diag_i = sh.update(X, post, pr["idx"], pr["wts"], pr["ob"], pr["lat"].reshape(-1), pr["lon"].reshape(-1),
                       inflation=1.3)
    torch.cuda.synchronize()
    Xi = orc.inflate_constant(pr["X"], 1.3)
    assert_parity(X.cpu().numpy(), Xi, "inflated resident prior")
    ref_post_i, ref_diag_i = _oracle(pr, Xi)
    assert_parity(post.cpu().numpy(), ref_post_i, "post (inflation 1.3)")
    assert_parity(diag_i["prior_var"], ref_diag_i["prior_var"], "prior_var (inflation 1.3)")
    eng.ctx.close()


@pytest.mark.parametrize("loc", [False, True])
def test_two_logical_column_shards_equal_unsharded_bit_for_bit(loc):
    import torch
    from efa_xray_amd.distributed import ShardedEnSRF, HipEngine
    pr = _problem(loc, row_pick=True, seed=6)
    glat, glon = pr["lat"].reshape(-1), pr["lon"].reshape(-1)
    # unsharded
    e0 = HipEngine(0)
    s0 = ShardedEnSRF(e0, pr["n_lead"], pr["ncol"], pr["M"])
    X0 = torch.from_numpy(pr["X"]).to(e0.device)
    P0 = torch.empty_like(X0)
    d0 = s0.update(X0, P0, pr["idx"], pr["wts"], pr["ob"], glat, glon)
    torch.cuda.synchronize()
    full = P0.cpu().numpy()
    # two logical ranks, each with its own engine (context + workspaces), on the same device
    engines = [HipEngine(0), HipEngine(0)]
    # (partial_estimates / assimilate never touch torch.distributed: the sum below stands in for it)
    shards = [ShardedEnSRF(engines[r], pr["n_lead"], pr["ncol"], pr["M"], rank=r, world_size=2) for r in range(2)]
    Xl = [torch.from_numpy(np.ascontiguousarray(pr["X"][sh.local_rows()])).to(engines[0].device) for sh in shards]
    Pl = [torch.empty_like(x) for x in Xl]
    parts = [sh.partial_estimates(x, pr["idx"], pr["wts"]) for sh, x in zip(shards, Xl)]
    torch.cuda.synchronize()
    total = parts[0] + parts[1]                # the all-reduce, in process
    torch.cuda.synchronize()
    out = np.empty_like(full)
    for sh, x, p in zip(shards, Xl, Pl):
        d = sh.assimilate(x, p, total.clone(), pr["ob"], glat, glon)
        torch.cuda.synchronize()
        out[sh.local_rows()] = p.cpu().numpy()
        for key in ("prior_mean", "prior_var", "post_mean", "post_var"):
            assert np.array_equal(d[key], d0[key], equal_nan=True), key   # Phase A is replicated
    assert np.array_equal(out, full)
    ref_post, _ = _oracle(pr)
    assert_parity(out, ref_post, "sharded post vs oracle")
    for e in engines + [e0]:
        e.ctx.close()


def _global_gc_problem(seed=8, ny=91, nx=180, n_lead=3, M=24, P=400):
    """Obs drawn uniformly over the grid points of a global lat/lon grid: dense per km^2 near the poles."""
    rng = np.random.default_rng(seed)
    lat, lon = np.meshgrid(np.linspace(-90, 90, ny), np.linspace(0, 358, nx), indexing="ij")
    ncol = ny * nx
    N = n_lead * ncol
    X = rng.standard_normal((N, 1)) + 2.0 * rng.standard_normal((N, M))
    idx = rng.choice(N, P, replace=False)[:, None].astype(np.int64)
    col0 = idx[:, 0] % ncol
    ob = dict(value=rng.standard_normal(P), error=rng.uniform(0.5, 1.5, P), assim=rng.random(P) > 0.1, loc="GC",
              lat=lat.reshape(-1)[col0], lon=lon.reshape(-1)[col0], halfwidth=rng.uniform(700, 1300, P))
    return dict(n_lead=n_lead, ncol=ncol, ny=ny, nx=nx, M=M, P=P, N=N, X=X, idx=idx, wts=np.ones((P, 1)), lat=lat, lon=lon,
                ob=ob)


def test_gc_block_counts_match_the_oracle_taper():
    """`efa_gc_block_counts` (the cost of a 16-column block of the localised sweep) against a NumPy count with the
    oracle's distance_to_point + gaspari_cohn (ensemble.py:254-267, observation.py:117-130)."""
    from efa_xray_amd import _lib
    from test_distributed_gloo import numpy_block_counts
    pr = _global_gc_problem(P=300)
    ctx = _lib.get_context(0)
    ob = pr["ob"]
    cnt, bp, pairs = ctx.gc_block_counts(pr["lat"], pr["lon"], ob["lat"], ob["lon"], ob["halfwidth"], ob["assim"])
    ref_cnt, ref_bp, ref_pairs = numpy_block_counts(orc, pr["lat"], pr["lon"], ob)
    assert cnt.shape == ref_cnt.shape and pairs == ref_pairs
    assert np.array_equal(cnt, ref_cnt) and np.array_equal(bp, ref_bp)
    # an unassimilated ob may carry any radius (ensrf.py:74-76 precedes :101); an assimilated one may not
    hw = ob["halfwidth"].copy()
    hw[~ob["assim"]] = np.nan
    cnt2, _, _ = ctx.gc_block_counts(pr["lat"], pr["lon"], ob["lat"], ob["lon"], hw, ob["assim"])
    assert np.array_equal(cnt2, cnt)
    hw[np.nonzero(ob["assim"])[0][0]] = np.nan
    with pytest.raises(_lib.EfaError):
        ctx.gc_block_counts(pr["lat"], pr["lon"], ob["lat"], ob["lon"], hw, ob["assim"])


@pytest.mark.parametrize("world", [2, 4])
def test_cost_balanced_logical_shards_equal_unsharded_and_are_balanced(world):
    """Cost-balanced contiguous column shards of a global Gaspari-Cohn cycle, as `world` logical ranks on one GPU:
    bit-equal to the unsharded run, and the cost of each shard's one-pass sweep -- the (column, ob) pairs it finds plus
    the fixed cost of its column blocks -- is within 5 % of the mean; under the equal split of ensemble.py:98-106 it is not."""
    import torch
    from efa_xray_amd.distributed import ShardedEnSRF, HipEngine, column_bounds, GC_FIXED_COST
    pr = _global_gc_problem(P=1500)
    glat, glon = pr["lat"].reshape(-1), pr["lon"].reshape(-1)
    e0 = HipEngine(0)
    s0 = ShardedEnSRF(e0, pr["n_lead"], pr["ncol"], pr["M"])
    X0 = torch.from_numpy(pr["X"]).to(e0.device)
    P0 = torch.empty_like(X0)
    s0.update(X0, P0, pr["idx"], pr["wts"], pr["ob"], glat, glon)
    torch.cuda.synchronize()
    full = P0.cpu().numpy()
    total_pairs = e0.ctx.get_option("gc_active_pairs")

    def run(make):
        engines = [HipEngine(0) for _ in range(world)]
        shards = [make(engines[r], r) for r in range(world)]
        Xl = [torch.from_numpy(np.ascontiguousarray(pr["X"][sh.local_rows()])).to(e0.device) for sh in shards]
        parts = [sh.partial_estimates(x, pr["idx"], pr["wts"]) for sh, x in zip(shards, Xl)]
        torch.cuda.synchronize()
        total = sum(parts[1:], parts[0].clone())
        out = np.empty_like(full)
        pairs = []
        for sh, x in zip(shards, Xl):
            p = torch.empty_like(x)
            sh.assimilate(x, p, total.clone(), pr["ob"], glat, glon)
            torch.cuda.synchronize()
            out[sh.local_rows()] = p.cpu().numpy()
            pairs.append(sh.engine.ctx.get_option("gc_active_pairs"))
        for e in engines:
            e.ctx.close()
        cost = np.array(pairs, dtype=np.float64) + GC_FIXED_COST * np.array([(sh.hi - sh.lo + 15) // 16 for sh in shards])
        return out, np.array(pairs, dtype=np.float64), [sh.bounds for sh in shards], cost

    out, pairs, bounds, cost = run(lambda e, r: ShardedEnSRF.balanced(e, pr["n_lead"], pr["ncol"], pr["M"], pr["ob"], glat, glon,
                                                                      rank=r, world_size=world))
    assert all(b == bounds[0] for b in bounds), "every rank must derive the same plan"
    assert np.array_equal(out, full)
    assert pairs.sum() == total_pairs
    assert cost.max() / cost.mean() <= 1.05, (pairs, cost)
    out_eq, pairs_eq, _, cost_eq = run(lambda e, r: ShardedEnSRF(e, pr["n_lead"], pr["ncol"], pr["M"], rank=r, world_size=world))
    assert np.array_equal(out_eq, full)
    assert bounds[0] != column_bounds(pr["ncol"], world)
    if world == 4:
        assert cost_eq.max() / cost_eq.mean() > 1.15, (pairs_eq, cost_eq)
    e0.ctx.close()


def test_library_owned_rccl_all_reduce_world_size_1():
    """`efa_comm_unique_id` / `efa_comm_init` / `efa_allreduce_sum_dev` / `efa_comm_destroy`: the exchange step of
    SURVEY.md 8(e) inside the C ABI, on librccl opened at run time.  One GPU allows world size 1 only (RCCL refuses two
    ranks on one device): the sum over one rank returns the buffer unchanged, on the context's stream, and
    `ShardedEnSRF` on a `HipEngine` with a communicator gives the oracle's posterior."""
    import torch
    from efa_xray_amd import _lib
    from efa_xray_amd.distributed import ShardedEnSRF, HipEngine
    eng = HipEngine(0)
    with pytest.raises(_lib.EfaError):
        eng.ctx.allreduce_sum(torch.zeros(4, dtype=torch.float64, device=eng.device).data_ptr(), 4)   # no communicator yet
    eng.init_comm(0, 1)
    with pytest.raises(_lib.EfaError):
        eng.ctx.comm_init(eng.ctx.comm_unique_id(), 0, 1)            # a context owns one communicator
    rng = np.random.default_rng(3)
    h = rng.standard_normal((500, 100))
    t = torch.from_numpy(h).to(eng.device)
    eng.all_reduce_sum(t)
    eng.ctx.synchronize()
    assert np.array_equal(t.cpu().numpy(), h)
    pr = _problem(True, row_pick=False)
    sh = ShardedEnSRF(eng, pr["n_lead"], pr["ncol"], pr["M"], rank=0, world_size=1)
    sh.world_size = 2          # force the exchange step through the library's communicator (of size 1)
    X = torch.from_numpy(pr["X"]).to(eng.device)
    post = torch.empty_like(X)
    sh.update(X, post, pr["idx"], pr["wts"], pr["ob"], pr["lat"].reshape(-1), pr["lon"].reshape(-1))
    torch.cuda.synchronize()
    ref_post, _ = _oracle(pr)
    assert_parity(post.cpu().numpy(), ref_post, "post through the library's all-reduce")
    eng.ctx.comm_destroy()
    eng.ctx.comm_destroy()                                           # idempotent
    eng.ctx.close()
