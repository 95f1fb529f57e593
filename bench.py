#!/usr/bin/env python3
"""bench.py -- EnSRF cov+update throughput on MI355X (driver contract).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full serial-EnSRF assimilation cycle of the headline workload
(BASELINE.json metric: 1e7 state x 100 members, 1e4 obs, float64, no
localisation) on data already resident in HBM: forward-operator gather of the
obs-space priors, (N>1: RCCL all-reduce of them), Phase A over the obs block,
and the state sweep prior-members -> posterior-members.  With N GPUs every rank
holds its own 1e7-row shard of an N*1e7-row state (weak scaling) and
assimilates all P obs into it; value = P*N / time.

Prints ONE JSON line on rank 0 (fields documented in DESIGN.md section
"Measurement").
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

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6     # MI355X fp64 vector = matrix peak (spec)

WORKLOADS = {
    # name: rows per GPU, members, obs, loc, (n_lead, ny, nx) for GC
    "headline": dict(rows=10_000_000, M=100, P=10_000, loc=None,
                     desc="headline (BASELINE.json metric): 1e7 state x 100 members x 1e4 obs, float64, loc=None"),
    "cfg2": dict(rows=512 * 512, M=50, P=1_000, loc=None,
                 desc="configs[1]: 512x512 grid x 50 members x 1000 obs, float64, loc=None"),
    "cfg3": dict(rows=4 * 37 * 361 * 720, M=80, P=5_000, loc="GC", n_lead=148, ny=361, nx=720, radius_km=1000.0,
                 desc="configs[2]: (lat=361,lon=720,lev=37,vars=4) x 80 members x 5000 obs, GC 1000 km"),
    "small": dict(rows=200_000, M=100, P=500, loc=None, desc="small smoke workload"),
    "small_gc": dict(rows=8 * 90 * 180, M=40, P=300, loc="GC", n_lead=8, ny=90, nx=180, radius_km=1000.0,
                     desc="small GC smoke workload"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--rows", type=int, default=None, help="override rows per GPU")
    ap.add_argument("--obs", type=int, default=None, help="override observation count")
    ap.add_argument("--path", default="auto", choices=["auto", "sweep", "transform"])
    ap.add_argument("--obs-batch", type=int, default=None)
    ap.add_argument("--gram", type=int, default=None, help="Phase-A leader in Gram space (library default if omitted)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=1_000_000)
    ap.add_argument("--cpu-obs", type=int, default=4)
    return ap.parse_args()


def cpu_baseline(sample_X, sample_HX, val, err, rows_full, loc_kw):
    """The oracle (NumPy restatement of the reference, faithful_cost=True so it
    performs the reference's passes over the matrix) timed on this box's host
    cores on a bounded sample, scaled linearly in rows (per-ob cost
    is linear in rows and independent of the ob index)."""
    from oracle import ensrf_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    nwarm = 1
    P = sample_HX.shape[0]
    rows = sample_X.shape[0]
    xbm, Xbp = orc.format_prior_state(sample_X, sample_HX)
    asm = np.ones(P, dtype=bool)
    # time the loop on a prefix of nwarm obs and on all P obs; the difference is P-nwarm warm obs
    t0 = time.perf_counter()
    _run_prefix(orc, xbm, Xbp, rows, val, err, asm, nwarm, loc_kw)
    t1 = time.perf_counter()
    _run_prefix(orc, xbm, Xbp, rows, val, err, asm, P, loc_kw)
    t2 = time.perf_counter()
    per_ob = ((t2 - t1) - (t1 - t0)) / (P - nwarm)
    obs_per_s_sample = 1.0 / per_ob
    return dict(value=obs_per_s_sample * rows / rows_full, unit="obs/s", cores=int(cores), kind="port",
                sample="NumPy oracle (faithful_cost) on the first %d rows x %d members of the same synthetic "
                       "state, %d obs timed after %d warm-up (%.2f s/ob), scaled linearly to %d rows"
                       % (rows, sample_X.shape[1], P - nwarm, nwarm, per_ob, rows_full))


def _run_prefix(orc, xbm, Xbp, rows, val, err, asm, n, loc_kw):
    """Assimilate the first n obs of the sample (augmented rows trimmed to n)."""
    A = rows + n
    kw = {}
    if loc_kw:
        kw = dict(loc="GC", ob_lat=loc_kw["ob_lat"][:n], ob_lon=loc_kw["ob_lon"][:n],
                  ob_halfwidth=loc_kw["hw"][:n], grid_lat=loc_kw["lat"], grid_lon=loc_kw["lon"],
                  state_shape=loc_kw["state_shape"])
    return orc.ensrf_update(xbm[:A], Xbp[:A], rows, val[:n], err[:n], asm[:n], faithful_cost=True, **kw)


def load_traffic(workload, path_name):
    """HBM bytes per launch of the dominant kernel from a separate rocprofv3
    --pmc pass (profiles/traffic.json, written by tools/prof_summary.py)."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(p))
        return t.get("%s:%s" % (workload, path_name))
    except Exception:
        return None


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

    import torch            # plumbing only: device selection, RCCL, barriers
    import torch.distributed as dist
    from efa_xray_amd import _lib

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    wl = dict(WORKLOADS[args.workload])
    if args.rows:
        wl["rows"] = args.rows
        if wl.get("loc"):
            raise SystemExit("--rows cannot be combined with a GC workload")
    if args.obs:
        wl["P"] = args.obs
    rows, M, P = wl["rows"], wl["M"], wl["P"]
    loc = wl["loc"]
    row_offset = rank * rows
    rows_global = rows * world

    ctx = _lib.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_option("path", {"auto": 0, "sweep": 1, "transform": 2}[args.path])
    if args.obs_batch:
        ctx.set_option("obs_batch", args.obs_batch)
    if args.gram is not None:
        ctx.set_option("gram", args.gram)
    ctx.set_option("timing", 1)

    # ---- synthetic inputs, generated on device per shard -------
    seed = 1000 + sorted(WORKLOADS).index(args.workload)
    X = torch.empty((rows, M), dtype=torch.float64, device=dev)
    post = torch.empty((rows, M), dtype=torch.float64, device=dev)
    ctx.fill_synthetic(rows, row_offset, M, seed, 3.0, X.data_ptr())
    rng = np.random.default_rng(3000 + seed)
    pick = rng.choice(rows_global, P, replace=False).astype(np.int64)
    idx = pick[:, None].copy()
    wts = np.ones((P, 1))
    err = np.ones(P)
    asm = np.ones(P, dtype=bool)
    loc_mode = _lib.LOC_GC if loc == "GC" else _lib.LOC_NONE
    ob_lat = ob_lon = hw = glat = glon = None
    n_lead = 1
    loc_kw = None
    if loc == "GC":
        # shard by (y,x) columns: this rank owns columns [c0, c0+ncol) of the global ny*nx*world grid
        n_lead, ny, nx = wl["n_lead"], wl["ny"], wl["nx"]
        lat2, lon2 = np.meshgrid(np.linspace(-90, 90, ny), np.linspace(0, 360 - 360.0 / nx, nx), indexing="ij")
        glat, glon = lat2.reshape(-1), lon2.reshape(-1)
        ncol = ny * nx
        # weak scaling: every rank holds a full copy-sized grid shard of a world-times-larger state
        col = (pick % rows) % ncol
        ob_lat, ob_lon = glat[col], glon[col]
        hw = np.full(P, wl["radius_km"])
        loc_kw = dict(ob_lat=ob_lat, ob_lon=ob_lon, hw=hw)
    HX = torch.empty((P, M), dtype=torch.float64, device=dev)
    ym = torch.empty((P,), dtype=torch.float64, device=dev)

    def forward():
        ctx.forward_stencil(rows, row_offset, M, X.data_ptr(), idx, wts, HX.data_ptr())
        if world > 1:
            dist.all_reduce(HX, op=dist.ReduceOp.SUM)      # the one exchange step

    forward()
    torch.cuda.synchronize()
    hx_host = HX.cpu().numpy()
    val = hx_host.mean(axis=1) + np.random.default_rng(4000 + seed).standard_normal(P) * np.sqrt(err)

    def step():
        forward()
        ctx.form_perts(P, M, HX.data_ptr(), ym.data_ptr(), HX.data_ptr())
        d = ctx.obs_phase(M, P, ym.data_ptr(), HX.data_ptr(), val, err, asm, loc_mode, ob_lat, ob_lon, hw)
        ctx.state_cycle(rows, M, X.data_ptr(), post.data_ptr(), glat, glon, n_lead)
        return d

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    state_ms = obs_ms = 0.0
    launches = 0
    path_taken = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        d = step()
        t = ctx.last_timing()
        state_ms += t["state_ms"]
        obs_ms += t["obs_ms"]
        launches += t["state_launches"]
        path_taken = t["path"]
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    n_active = int(d["assimilated"].sum())

    # sanity: the posterior must be finite and the variance must have shrunk
    chk = post[:4096].cpu().numpy()
    assert np.isfinite(chk).all(), "non-finite posterior"

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = n_active * world / (elapsed / args.steps)
        bytes_per_ob = 16.0 * rows * (M + 1)                 # per-shard algorithmic bytes
        path_name = {1: "sweep", 2: "transform"}.get(path_taken, "sweep")
        avg_launch_ms = state_ms / max(launches, 1)
        obs_per_launch = n_active * args.steps / max(launches, 1)
        achieved = bytes_per_ob * obs_per_launch / (avg_launch_ms * 1e-3) / 1e9
        phys_bytes = 16.0 * rows * M if path_name == "transform" else 16.0 * rows * (M + 1)
        flops_launch = (2.0 * rows * M * (M + 1)) if path_name == "transform" else 4.0 * rows * M * obs_per_launch
        out = {
            "metric": "obs assimilated/sec on cov+update (EnSRF cycle, %d state x %d members x %d obs)" % (rows, M, P),
            "value": value, "unit": "obs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl["desc"], "rows_per_gpu": rows, "members": M, "obs": P,
                       "loc": loc or "none", "path": path_name, "obs_batch": ctx.get_option("obs_batch"),
                       "phase_a": {1: "pipeline", 2: "per-batch", 3: "pipeline-gram"}.get(ctx.get_option("phase_a_kind"), "?"),
                       "sharding": "state rows by grid point, obs block replicated, one all-reduce of HX per cycle"},
            "GBps_algorithmic": bytes_per_ob * n_active * world / (elapsed / args.steps) / 1e9,
            "phase_ms": {"obs_phase": obs_ms / args.steps, "state_phase": state_ms / args.steps},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": load_traffic(args.workload, path_name),
                         "kernel": "k_transform" if path_name == "transform" else "k_sweep",
                         "avg_launch_ms": avg_launch_ms, "obs_per_launch": obs_per_launch,
                         "launch_physical_min_bytes": phys_bytes,
                         "launch_physical_GBps": phys_bytes / (avg_launch_ms * 1e-3) / 1e9,
                         "launch_physical_frac": phys_bytes / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "launch_fp64_TFLOPs": flops_launch / (avg_launch_ms * 1e-3) / 1e12,
                         "launch_fp64_frac": flops_launch / (avg_launch_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS},
        }
        # Phase A (the serial chain) is the longer of the two launches at the headline size: it moves ~95 MB
        # and is bound by the step latency of ONE workgroup, so neither roofline applies; reported for scale
        kind = ctx.get_option("phase_a_kind")
        out["phase_a"] = {"kernel": {1: "k_pipe", 2: "k_diag+k_sweep", 3: "k_pipe_gram"}.get(kind, "?"),
                          "bound": "latency of the serial per-observation chain (one workgroup leads at a time)",
                          "ms": obs_ms / args.steps, "us_per_ob": 1e3 * obs_ms / args.steps / max(P, 1)}
        if loc:
            # a localised sweep touches only the rows inside each ob's support: the dense flop count does not apply
            out["roofline"]["launch_fp64_TFLOPs"] = None
            out["roofline"]["launch_fp64_frac"] = None
            out["roofline"]["kernel"] = "k_sweep_gc"
        if world == 1 and not args.no_cpu_baseline:
            n_cpu_rows = min(args.cpu_rows, rows)
            n_cpu_obs = min(args.cpu_obs + 1, P)
            sample = X[:n_cpu_rows].cpu().numpy()
            # sample obs: the first obs of the list, re-based onto sample rows so the loop is well-posed
            srows = (pick[:n_cpu_obs] % n_cpu_rows)
            s_hx = sample[srows]
            s_val = s_hx.mean(axis=1) + 0.5
            lk = None
            if loc == "GC":
                raise SystemExit("cpu baseline for GC workloads: use --no-cpu-baseline")
            out["cpu_baseline"] = cpu_baseline(sample, s_hx, s_val, err[:n_cpu_obs], rows, lk)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
