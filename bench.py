#!/usr/bin/env python3
"""bench.py -- EnSRF cov+update throughput on MI355X (driver contract).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--scaling strong --workload cfg4]

One "step" = one full serial-EnSRF assimilation cycle on data already resident in HBM, run
through the product's multi-GPU class (`efa_xray_amd.distributed.ShardedEnSRF` on a
`HipEngine`, also at N=1): forward-operator gather of the obs-space priors, (N>1: ONE RCCL
all-reduce of them), obs-space priors, Phase A over the obs block, and the state sweep
prior members -> posterior members.

The state is ONE global grid partitioned by (y,x) column into contiguous blocks: equal chunks, last
rank takes the remainder, without localisation (`column_bounds`; reference precedent
ensemble.py:98-106); chunks of equal COST under Gaspari-Cohn localisation (`balanced_column_bounds`).
The HX all-reduce is the library's own RCCL collective (`efa_allreduce_sum_dev`).

Defaults.  N = 1: the headline (BASELINE.json's metric: 1e7 x 100 x 1e4 obs, loc=None).
           N > 1: STRONG scaling of configs[3]'s one global 3-D state (`--workload cfg4`): value = P / (max over
           ranks of the cycle time); after the timed region rank 0 alone runs the same global problem unsharded
           so that the line carries a measured `speedup_vs_one_gpu`.  `--workload headline --scaling strong`
           gives the headline's strong scaling (Phase A is replicated: Amdahl-bound), `--scaling weak` (loc=None
           workloads only) the old series in which every rank holds the workload's rows.

Prints ONE JSON line on rank 0 (fields documented in DESIGN.md section 6).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
HBM_COPY_GBPS = 6290.0
FP64_PEAK_TFLOPS = 78.6     # MI355X fp64 vector = matrix peak (spec, at 2.4 GHz)

WORKLOADS = {
    # name: rows (global for strong scaling, per GPU for weak), members, obs, loc, (n_lead, ny, nx) for GC
    "headline": dict(rows=10_000_000, M=100, P=10_000, loc=None,
                     desc="headline (BASELINE.json metric): 1e7 state x 100 members x 1e4 obs, float64, loc=None"),
    "cfg2": dict(rows=512 * 512, M=50, P=1_000, loc=None,
                 desc="configs[1]: 512x512 grid x 50 members x 1000 obs, float64, loc=None"),
    "cfg3": dict(rows=4 * 37 * 361 * 720, M=80, P=5_000, loc="GC", n_lead=148, ny=361, nx=720, radius_km=1000.0,
                 desc="configs[2]: (lat=361,lon=720,lev=37,vars=4) x 80 members x 5000 obs, GC 1000 km"),
    "cfg4": dict(rows=4 * 37 * 361 * 720, M=100, P=10_000, loc="GC", n_lead=148, ny=361, nx=720, radius_km=1000.0,
                 desc="configs[3]: (lat=361,lon=720,lev=37,vars=4) x 100 members x 10000 obs, GC 1000 km, "
                      "state sharded by grid point"),
    "small": dict(rows=200_000, M=100, P=500, loc=None, desc="small smoke workload"),
    "small_gc": dict(rows=8 * 90 * 180, M=40, P=300, loc="GC", n_lead=8, ny=90, nx=180, radius_km=1000.0,
                     desc="small GC smoke workload"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: headline at --gpus 1, cfg4 (configs[3], one global state) at --gpus > 1")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"], help="default: strong (the same thing at N = 1)")
    ap.add_argument("--no-one-gpu-reference", action="store_true",
                    help="N > 1: skip the unsharded run of the same workload on rank 0 after the timed region")
    ap.add_argument("--no-balance", action="store_true", help="GC workloads: equal column chunks instead of equal cost")
    ap.add_argument("--rows", type=int, default=None, help="override the workload's row count (loc=None workloads)")
    ap.add_argument("--obs", type=int, default=None, help="override observation count")
    ap.add_argument("--path", default="auto", choices=["auto", "sweep", "transform"])
    ap.add_argument("--obs-batch", type=int, default=None)
    ap.add_argument("--gram", type=int, default=None, help="Phase-A leader in Gram space (library default if omitted)")
    ap.add_argument("--split-phases", action="store_true",
                    help="Phase A and the state phase as two library calls with a host round trip in between (A/B of efa_ensrf_cycle_dev)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-api", action="store_true", help="skip the PCIe-inclusive EnSRF.update() timing (N = 1)")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the CPU-baseline sample (0: sized to --cpu-seconds)")
    ap.add_argument("--cpu-obs", type=int, default=8, help="timed observations of the CPU baseline (BASELINE.md 3)")
    ap.add_argument("--cpu-warm", type=int, default=2, help="warm-up observations of the CPU baseline")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="time budget of the CPU baseline")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------
# CPU baseline: the oracle, on this box's host cores, on a bounded sample
# ----------------------------------------------------------------------------------------------
def _oracle_prefix_seconds(orc, xbm, Xbp, rows, val, err, n, loc_kw):
    """Wall time of assimilating the first n obs of the sample (augmented rows trimmed to n)."""
    A = rows + n
    kw = {}
    if loc_kw:
        kw = dict(loc="GC", ob_lat=loc_kw["ob_lat"][:n], ob_lon=loc_kw["ob_lon"][:n], ob_halfwidth=loc_kw["hw"][:n],
                  grid_lat=loc_kw["lat"], grid_lon=loc_kw["lon"], state_shape=loc_kw["state_shape"])
    t0 = time.perf_counter()
    orc.ensrf_update(xbm[:A], Xbp[:A], rows, val[:n], err[:n], np.ones(n, dtype=bool), faithful_cost=True, **kw)
    return time.perf_counter() - t0


def cpu_baseline(fetch_sample, rows_full, M, nwarm, ntimed, budget_s, fixed_rows, loc_info):
    """`fetch_sample(n_units)` returns (X_sample, HX_sample, val, err, loc_kw, rows_sample) for the first n_units
    rows (loc=None) or latitude circles (GC) of the bench's own synthetic state.  The oracle runs in
    faithful_cost mode (the reference's one-hot row picks and temporaries, i.e. its memory traffic) for
    `nwarm` warm-up obs, then `ntimed` timed obs (2 + 8); the per-ob cost is linear in rows and
    independent of the ob index, so the sample's rate is scaled by rows_sample / rows_full.
    The sample is the largest that keeps the whole measurement inside `budget_s` (probed on a small one)."""
    from oracle import ensrf_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    unit_rows = loc_info["unit_rows"]          # rows per sample unit (1, or n_lead*nx for a latitude circle)
    max_units = loc_info["max_units"]
    nobs = nwarm + ntimed

    def measure(units):
        Xs, HXs, val, err, loc_kw, rows_s = fetch_sample(units, nobs)
        xbm, Xbp = orc.format_prior_state(Xs, HXs)
        _oracle_prefix_seconds(orc, xbm, Xbp, rows_s, val, err, nwarm, loc_kw)           # warm-up, untimed
        t_warm = _oracle_prefix_seconds(orc, xbm, Xbp, rows_s, val, err, nwarm, loc_kw)  # the same obs again
        t_all = _oracle_prefix_seconds(orc, xbm, Xbp, rows_s, val, err, nobs, loc_kw)    # ... plus the timed ones
        return max(t_all - t_warm, 1e-9) / ntimed, rows_s

    if fixed_rows:
        units = max(1, min(max_units, fixed_rows // unit_rows))
    else:
        probe_units = max(1, min(max_units, 200_000 // unit_rows))
        per_ob, rows_p = measure(probe_units)
        per_row = per_ob / rows_p
        # the measurement costs (3*nwarm + ntimed) obs plus forming the sample (~2 obs' worth)
        units = int(budget_s / (per_row * unit_rows * (3 * nwarm + ntimed + 2)))
        units = max(probe_units, min(max_units, units))
    per_ob, rows_s = measure(units)
    return dict(value=(1.0 / per_ob) * rows_s / rows_full, unit="obs/s", cores=int(cores), kind="port",
                sample="NumPy oracle (faithful_cost: the reference's passes and temporaries) on the first %d rows x %d "
                       "members of the same synthetic state, %d warm-up + %d timed obs (%.3f s/ob on the sample), scaled "
                       "linearly to %d rows" % (rows_s, M, nwarm, ntimed, per_ob, rows_full))


def host_api_timing(device):
    """configs[1] (512 x 512 grid x 50 members x 1 000 obs, loc=None) through the Python API from HOST memory:
    `EnSRF(state, obs).update()` with plain `Observation`s -- upload of the state, forward operator on the device,
    Phase A, Phase B, download into a new state object.  The PCIe-inclusive time; never the `value`."""
    from efa_xray_amd import EnsembleState, Observation, EnSRF
    rng = np.random.default_rng(77)
    ny = nx = 512
    M, P = 50, 1000
    lat, lon = np.meshgrid(np.linspace(20, 60, ny), np.linspace(200, 280, nx), indexing="ij")
    arr = rng.standard_normal((1, 1, ny, nx, 1)) + 3.0 * rng.standard_normal((1, 1, ny, nx, M))
    state = EnsembleState.from_array(arr, lat, lon)
    obs = [Observation(value=float(rng.standard_normal()), obtype="var0", time=0, error=1.0, lat=float(rng.uniform(21, 59)),
                       lon=float(rng.uniform(201, 279)), assimilate_this=True) for _ in range(P)]
    times, dev = [], []
    for _ in range(4):
        flt = EnSRF(state, obs, verbose=False, loc=None, device=device)
        t0 = time.perf_counter()
        post, _ = flt.update()
        times.append(time.perf_counter() - t0)
        dev.append(flt.last_timing["obs_ms"] + flt.last_timing["state_ms"])
    assert np.isfinite(post.to_vect()[:64]).all()
    nbytes = arr.nbytes
    return {"workload": "configs[1] through EnSRF.update() from host memory (state %.0f MB up, %.0f MB down, pageable)" % (nbytes / 1e6, nbytes / 1e6),
            "pcie_inclusive_ms": 1e3 * min(times[1:]), "device_phase_ms": float(np.median(dev[1:])),
            "host_passes_over_state": "1 upload (slab by slab from the variables) + 1 download (into the new state's arrays)"}


def load_traffic(workload, path_name):
    """HBM bytes per launch of the state-sweep kernel from separate rocprofv3 --pmc passes
    (profiles/traffic.json, written by tools/prof_summary.py; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950 wide streaming reads)."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(p)).get("%s:%s" % (workload, path_name))
    except Exception:
        return None


def measure(args, wl, strong, world, rank, eng, dist, torch, seed, balance=True):
    """One workload on `world` ranks (this process is `rank`): inputs generated on the device, warm-up, then
    `args.steps` timed cycles between barriers.  Returns the raw numbers of THIS rank plus the shard object."""
    from efa_xray_amd.distributed import ShardedEnSRF
    ctx = eng.ctx
    M, P, loc = wl["M"], wl["P"], wl["loc"]
    if loc == "GC":
        if not strong and world > 1:
            raise SystemExit("--scaling weak is defined for loc=None workloads only (a localised cycle has ONE globe)")
        n_lead, ny, nx = wl["n_lead"], wl["ny"], wl["nx"]
        lat2, lon2 = np.meshgrid(np.linspace(-90, 90, ny), np.linspace(0, 360 - 360.0 / nx, nx), indexing="ij")
        glat, glon = lat2.reshape(-1), lon2.reshape(-1)
        ncol_g = ny * nx
    else:
        n_lead = 1
        ncol_g = wl["rows"] if strong else wl["rows"] * world
        glat = glon = lat2 = lon2 = None
    rows_g = n_lead * ncol_g
    rng = np.random.default_rng(3000 + seed)
    pick = rng.choice(rows_g, P, replace=False).astype(np.int64)
    idx = pick[:, None].copy()
    wts = np.ones((P, 1))
    err = np.ones(P)
    ob = dict(value=None, error=err, assim=np.ones(P, dtype=bool))
    if loc == "GC":
        col = pick % ncol_g
        ob.update(loc="GC", lat=glat[col], lon=glon[col], halfwidth=np.full(P, wl["radius_km"]))
    if loc == "GC" and balance and world > 1:
        sh = ShardedEnSRF.balanced(eng, n_lead, ncol_g, M, ob, glat, glon, rank=rank, world_size=world)
    else:
        sh = ShardedEnSRF(eng, n_lead, ncol_g, M, rank=rank, world_size=world)
    rows = sh.rows_local
    ncol_l = sh.hi - sh.lo

    # synthetic inputs, generated on device per shard; keyed by GLOBAL row
    X = eng.empty((rows, M))
    post = eng.empty((rows, M))
    for lead in range(n_lead):   # local rows of one lead are one contiguous run of global rows
        ctx.fill_synthetic(ncol_l, lead * ncol_g + sh.lo, M, seed, 3.0, X.data_ptr() + lead * ncol_l * M * 8)
    HX0 = sh.partial_estimates(X, idx, wts)
    sh.all_reduce_sum(HX0)
    torch.cuda.synchronize()
    hx_host = HX0.cpu().numpy()
    ob["value"] = hx_host.mean(axis=1) + np.random.default_rng(4000 + seed).standard_normal(P) * np.sqrt(err)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        sh.update(X, post, idx, wts, ob, glat, glon)
    sync_all()
    # Phase times come from HIP events the library records on its stream around every phase of the timed steps ("timing" 2:
    # deferred -- no call waits for its own events, the sums are read ONCE after the closing synchronisation; with "timing" 1
    # every state phase ended in an event wait, so the host could not prepare the next cycle while the transform ran)
    ctx.last_timing()                      # (clears the sums of the warm-up steps)
    # the interpreter's cyclic garbage collector stays out of the timed region, as in timeit: a full collection with torch
    # imported takes ~30 ms, four cycles' worth, and falls into one run in a few (tools/cycle_jitter.py: 1 of 400 cycles)
    import gc
    gc.collect()
    gc.disable()
    try:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            d = sh.update(X, post, idx, wts, ob, glat, glon)
        sync_all()
        elapsed = time.perf_counter() - t0
    finally:
        gc.enable()
    t = ctx.last_timing()
    state_ms, obs_ms, launches, path_taken = t["state_ms"], t["obs_ms"], t["state_launches"], t["path"]
    my_elapsed = elapsed
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    chk = post[:4096].cpu().numpy()
    assert np.isfinite(chk).all(), "non-finite posterior"
    pairs = float(ctx.get_option("gc_active_pairs")) if loc == "GC" else None
    return dict(sh=sh, X=X, post=post, pick=pick, glat=glat, glon=glon, lat2=lat2, lon2=lon2, n_lead=n_lead, ncol_g=ncol_g,
                rows_g=rows_g, rows=rows, ncol_l=ncol_l, elapsed=elapsed, my_elapsed=my_elapsed, state_ms=state_ms, obs_ms=obs_ms,
                launches=launches, path_taken=path_taken, n_active=int(d["assimilated"].sum()), pairs=pairs,
                kind=ctx.get_option("phase_a_kind"))


def init_library_comm(eng, rank, world, dist, torch, timeout_s=120.0):
    """The library's own RCCL communicator (efa_comm_init) with one checked all-reduce, created in a helper thread so that a
    communicator that cannot be brought up on this node (it has only ever run at world size 1 where this was built) costs a
    timeout, not the run: the ranks then agree (one MIN all-reduce through torch.distributed) to use torch.distributed's
    own RCCL all-reduce for the exchange step instead.  Both are RCCL on the GPUs; which one ran is reported in `collective`."""
    import threading
    ok = [False]
    err = [None]

    def bring_up():
        try:
            eng.init_comm(rank, world)
            probe = torch.full((1024,), float(rank + 1), dtype=torch.float64, device=eng.device)
            eng.all_reduce_sum(probe)
            torch.cuda.synchronize()
            ok[0] = bool((probe == world * (world + 1) / 2.0).all().item())
            if not ok[0]:
                err[0] = "probe all-reduce returned %r" % probe[0].item()
        except Exception as e:          # noqa: BLE001 -- any failure means: use the other collective
            err[0] = repr(e)

    t = threading.Thread(target=bring_up, daemon=True)
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        err[0] = "efa_comm_init did not return within %.0f s" % timeout_s
    flag = torch.tensor([1.0 if (ok[0] and not t.is_alive()) else 0.0], device=eng.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if flag.item() == 1.0:
        return "efa_allreduce_sum_dev (RCCL ncclAllReduce sum f64 on the context stream)"
    eng.has_comm = False
    if err[0] is not None:
        sys.stderr.write("[bench] rank %d: library communicator unavailable (%s); using torch.distributed all_reduce\n" % (rank, err[0]))
    return "torch.distributed all_reduce (RCCL) -- the library's own communicator could not be brought up on this node"


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)\n"
                             % (args.gpus, world))
        if args.gpus != 1 or world != 1:
            sys.exit(2)
    if args.workload is None:
        args.workload = "headline" if world == 1 else "cfg4"
    if args.scaling is None:
        args.scaling = "strong"

    import torch            # plumbing only: device memory, the stream, the rendezvous, barriers
    import torch.distributed as dist
    from efa_xray_amd.distributed import HipEngine

    # EFA_BENCH_REHEARSE=1: every rank on cuda:0 with the gloo backend -- a rehearsal of the N > 1 code path
    # (sharding, the HX all-reduce, barriers, max-over-ranks) on a one-GPU box; its timings mean nothing and the
    # driver never sets it.  Normal runs: one rank per GPU; torch.distributed ("nccl") carries the barriers and the
    # communicator id, the HX all-reduce itself is the library's RCCL collective.
    rehearse = os.environ.get("EFA_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    wl = dict(WORKLOADS[args.workload])
    if args.rows:
        if wl.get("loc"):
            raise SystemExit("--rows cannot be combined with a GC workload")
        wl["rows"] = args.rows
    if args.obs:
        wl["P"] = args.obs
    M, P, loc = wl["M"], wl["P"], wl["loc"]
    strong = args.scaling == "strong"

    eng = HipEngine(local_rank)
    ctx = eng.ctx
    ctx.set_option("path", {"auto": 0, "sweep": 1, "transform": 2}[args.path])
    if args.obs_batch:
        ctx.set_option("obs_batch", args.obs_batch)
    if args.gram is not None:
        ctx.set_option("gram", args.gram)
    ctx.set_option("timing", 2)
    if args.split_phases:
        eng.fused_cycle = False
    collective = "none (one rank)"
    if world > 1 and not rehearse:
        collective = init_library_comm(eng, rank, world, dist, torch)
    elif world > 1:
        collective = "torch.distributed gloo (rehearsal)"

    seed = 1000 + sorted(WORKLOADS).index(args.workload)
    r = measure(args, wl, strong, world, rank, eng, dist, torch, seed, balance=not args.no_balance)
    sh, X, post, pick = r["sh"], r["X"], r["post"], r["pick"]
    glat, glon, lat2, lon2 = r["glat"], r["glon"], r["lat2"], r["lon2"]
    n_lead, ncol_g, rows_g, rows, ncol_l = r["n_lead"], r["ncol_g"], r["rows_g"], r["rows"], r["ncol_l"]
    elapsed, state_ms, obs_ms, launches, path_taken = r["elapsed"], r["state_ms"], r["obs_ms"], r["launches"], r["path_taken"]
    n_active = r["n_active"]
    nx_g = wl.get("nx")

    # per-rank view (imbalance is visible here): rows, (column, ob) pairs, phase times, own wall time
    mine = dict(rank=rank, columns=[int(sh.lo), int(sh.hi)], rows=int(rows), active_pairs=r["pairs"],
                obs_phase_ms=obs_ms / args.steps, state_phase_ms=state_ms / args.steps,
                cycle_ms=1e3 * r["my_elapsed"] / args.steps)
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    # N > 1: the SAME global problem, unsharded, on rank 0 alone (the other ranks wait at the barrier)
    one_gpu = None
    if world > 1 and strong and not args.no_one_gpu_reference and not rehearse:
        X = post = r = None
        torch.cuda.empty_cache()
        if rank == 0:
            r1 = measure(args, wl, True, 1, 0, eng, dist, torch, seed)
            one_gpu = dict(ms_per_step=1e3 * r1["elapsed"] / args.steps, value=r1["n_active"] / (r1["elapsed"] / args.steps),
                           obs_phase_ms=r1["obs_ms"] / args.steps, state_phase_ms=r1["state_ms"] / args.steps)
            r1 = None
        dist.barrier()

    if rank == 0:
        sec_per_step = elapsed / args.steps
        ms_per_step = 1e3 * sec_per_step
        value = n_active * (1 if strong else world) / sec_per_step
        path_name = {1: "sweep", 2: "transform"}.get(path_taken, "sweep")
        kind = ctx.get_option("phase_a_kind")
        pa_kernel = {1: "k_pipe", 2: "k_diag+k_sweep", 3: "k_pipe_gram", 4: "k_pipe_band"}.get(kind, "?")
        state_step_ms = state_ms / args.steps
        obs_step_ms = obs_ms / args.steps
        avg_launch_ms = state_ms / max(launches, 1)
        launches_per_step = launches / float(args.steps)
        obs_per_launch = n_active * args.steps / max(launches, 1)
        # algorithmic bytes per assimilated ob = 16*N*(M+1) (per shard)
        bytes_per_ob = 16.0 * rows * (M + 1)
        # physical minimum of ONE state-sweep launch: read + write of the member block (and, in perturbation
        # form, the mean); the fused transform / one-pass GC launches carry means in registers only
        if loc == "GC":
            # members in / members out with an even M <= 104: the row-per-lane form of the one-pass sweep (efa_gcsweep.hip)
            kernel = ("k_sweep_gc_lane" if (M <= 104 and M % 2 == 0) else "k_sweep_gc") if ctx.get_option("gc_onepass") else "k_sweep"
            one_pass = bool(ctx.get_option("gc_onepass"))
            phys_bytes = 16.0 * rows * M if one_pass else 16.0 * rows * (M + 1)
        elif path_name == "transform":
            kernel, phys_bytes = "k_transform", 16.0 * rows * M
        else:
            kernel, phys_bytes = "k_sweep", 16.0 * rows * (M + 1)
        phys_gbps = phys_bytes / (avg_launch_ms * 1e-3) / 1e9
        if loc == "GC":
            pairs = per_rank[0]["active_pairs"]                          # (column, ob) pairs with taper != 0, rank 0's shard
            flops_step = 4.0 * M * pairs * n_lead
            bytes_touched = 16.0 * (M + 1) * pairs * n_lead             # SURVEY.md 8d, localised configs
        elif path_name == "transform":
            pairs, bytes_touched = None, None
            flops_step = 2.0 * rows * M * (M + 1)
        else:
            pairs, bytes_touched = None, None
            flops_step = 4.0 * rows * M * n_active
        fp64_tflops = flops_step / (state_step_ms * 1e-3) / 1e12
        dominant = pa_kernel if obs_step_ms > state_step_ms else kernel
        out = {
            "metric": "obs assimilated/sec on cov+update (EnSRF cycle, %d state x %d members x %d obs)"
                      % (rows_g if strong else rows, M, P),
            "value": value, "unit": "obs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic" if not rehearse else "synthetic; REHEARSAL (all ranks on one GPU, gloo): timings are not a measurement",
            "config": {"workload": wl["desc"], "rows_per_gpu": rows, "rows_global": rows_g, "members": M, "obs": P,
                       "loc": loc or "none", "path": path_name, "obs_batch": ctx.get_option("obs_batch"),
                       "phase_a": {1: "pipeline", 2: "per-batch", 3: "pipeline-gram", 4: "pipeline-band"}.get(kind, "?"),
                       "sharding": "one global grid split by (y,x) column over the ranks (ShardedEnSRF: %s), obs block "
                                   "replicated, one all-reduce of HX per cycle, no per-observation communication"
                                   % ("contiguous chunks of equal Gaspari-Cohn cost" if (loc == "GC" and world > 1 and not args.no_balance)
                                      else "contiguous equal chunks"),
                       "collective": collective},
            # whole-cycle rates: algorithmic (effective, counts every ob's nominal pass) and physical
            "GBps_algorithmic": bytes_per_ob * n_active * world / sec_per_step / 1e9,
            "phase_ms": {"obs_phase": obs_step_ms, "state_phase": state_step_ms,
                         "other": ms_per_step - obs_step_ms - state_step_ms},
            # roofline of the state-sweep kernel (the cov+update sweep the metric names).  `achieved` / `frac` are
            # PHYSICAL: the bytes one launch must move at minimum / its average duration (HIP events on the launch
            # stream, inside the library) against the 8 TB/s spec.  The algorithmic figure of SURVEY.md 8d is
            # `effective_GBps` (>> peak when one launch assimilates many obs).
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": phys_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": phys_gbps / HBM_PEAK_GBPS, "frac_of_copy_ceiling": phys_gbps / HBM_COPY_GBPS,
                         "traffic": load_traffic(args.workload, "gc" if loc == "GC" else path_name) if world == 1 else None,
                         "bytes_per_launch_min": phys_bytes, "avg_launch_ms": avg_launch_ms,
                         "launches_per_step": launches_per_step, "obs_per_launch": obs_per_launch,
                         "effective_GBps": bytes_per_ob * obs_per_launch / (avg_launch_ms * 1e-3) / 1e9,
                         "fp64_TFLOPs": fp64_tflops, "fp64_frac": fp64_tflops / FP64_PEAK_TFLOPS,
                         "cycle_frac": phys_bytes * launches_per_step / sec_per_step / 1e9 / HBM_PEAK_GBPS,
                         "dominant_kernel_by_time": dominant},
            # Phase A (the serial chain): ~10 MB of HBM traffic per launch, bound by the step latency of ONE
            # workgroup's chain, so neither roofline applies; reported as time and shader cycles per observation
            "phase_a": {"kernel": pa_kernel,
                        "bound": "latency of the serial per-observation chain (one workgroup leads at a time)",
                        "ms": obs_step_ms, "us_per_ob": 1e3 * obs_step_ms / max(P, 1),
                        "cycles_per_ob_at_2.4GHz": 2400.0 * (1e3 * obs_step_ms / max(P, 1)),
                        "share_of_cycle": obs_step_ms / ms_per_step},
        }
        if loc == "GC":
            out["roofline"]["active_column_ob_pairs"] = pairs
            out["roofline"]["bytes_touched"] = bytes_touched
            out["roofline"]["bytes_touched_GBps"] = bytes_touched / (state_step_ms * 1e-3) / 1e9
        if world > 1:
            out["per_rank"] = per_rank
            st = np.array([p["state_phase_ms"] for p in per_rank])
            out["imbalance"] = {"state_phase_max_over_mean": float(st.max() / st.mean())}
            if loc == "GC":
                pr_ = np.array([p["active_pairs"] for p in per_rank], dtype=np.float64)
                out["imbalance"]["active_pairs_max_over_mean"] = float(pr_.max() / pr_.mean())
            if one_gpu is not None:
                out["one_gpu_same_workload"] = one_gpu
                out["speedup_vs_one_gpu"] = value / one_gpu["value"]
        if world == 1 and not args.no_cpu_baseline:
            if loc == "GC":
                unit_rows, max_units = n_lead * nx_g, wl["ny"]
            else:
                unit_rows, max_units = 1, rows

            def fetch_sample(units, nobs):
                if loc == "GC":
                    ncs = units * nx_g                                   # the first `units` latitude circles
                    Xs = torch.cat([X[l * ncol_l:l * ncol_l + ncs] for l in range(n_lead)]).cpu().numpy()
                    rows_s = n_lead * ncs
                    scol = pick[:nobs] % ncs
                    srow = (pick[:nobs] // ncol_g) % n_lead * ncs + scol
                    lk = dict(ob_lat=glat[scol], ob_lon=glon[scol], hw=np.full(nobs, wl["radius_km"]),
                              lat=lat2[:units], lon=lon2[:units], state_shape=(n_lead, 1, units, nx_g))
                else:
                    Xs = X[:units].cpu().numpy()
                    rows_s = units
                    srow = pick[:nobs] % units
                    lk = None
                s_hx = Xs[srow]
                return Xs, s_hx, s_hx.mean(axis=1) + 0.5, np.ones(nobs), lk, rows_s

            out["cpu_baseline"] = cpu_baseline(fetch_sample, rows, M, args.cpu_warm, min(args.cpu_obs, max(P - args.cpu_warm, 1)),
                                               args.cpu_seconds, args.cpu_rows,
                                               dict(unit_rows=unit_rows, max_units=max_units))
        if world == 1 and args.workload == "headline" and not args.no_host_api:
            X = post = r = None
            torch.cuda.empty_cache()
            out["host_api"] = host_api_timing(local_rank)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        if "could not be brought up" in collective:
            # a helper thread may still sit inside the communicator's bring-up: leave without running its destructors
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
