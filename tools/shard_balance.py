#!/usr/bin/env python3
"""Measured per-shard cost of a strong-scaling split on ONE GPU: every logical rank of a `world`-way split of a
workload's global state runs its own cycle alone on the device (ShardedEnSRF + HipEngine, rank r of world; the HX
all-reduce is replaced by a forward-operator gather that sees the whole stencil: row-pick obs owned elsewhere are read
from a regenerated row).  The slowest shard bounds the N-GPU cycle (plus the all-reduce, ~0.1-0.2 ms): a projection
from measured shard times, not a scaling measurement.

    python tools/shard_balance.py --workload cfg4 --world 8 [--equal]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg4")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import torch
    from efa_xray_amd.distributed import ShardedEnSRF, HipEngine
    wl = bench.WORKLOADS[args.workload]
    M, P, loc = wl["M"], wl["P"], wl["loc"]
    seed = 1000 + sorted(bench.WORKLOADS).index(args.workload)
    eng = HipEngine(0)
    ctx = eng.ctx
    ctx.set_option("timing", 2)   # deferred: sums over the timed cycles, read once (no event wait inside the cycles)
    if loc == "GC":
        n_lead, ny, nx = wl["n_lead"], wl["ny"], wl["nx"]
        lat2, lon2 = np.meshgrid(np.linspace(-90, 90, ny), np.linspace(0, 360 - 360.0 / nx, nx), indexing="ij")
        glat, glon = lat2.reshape(-1), lon2.reshape(-1)
        ncol_g = ny * nx
    else:
        n_lead, ncol_g, glat, glon = 1, wl["rows"], None, None
    rows_g = n_lead * ncol_g
    rng = np.random.default_rng(3000 + seed)
    pick = rng.choice(rows_g, P, replace=False).astype(np.int64)
    ob = dict(value=None, error=np.ones(P), assim=np.ones(P, dtype=bool))
    if loc == "GC":
        col = pick % ncol_g
        ob.update(loc="GC", lat=glat[col], lon=glon[col], halfwidth=np.full(P, wl["radius_km"]))
    # the obs-space prior ensemble of the GLOBAL state: regenerate each picked row on its own (keyed by global row)
    HX = eng.empty((P, M))
    one = eng.empty((1, M))
    for k in range(P):
        ctx.fill_synthetic(1, int(pick[k]), M, seed, 3.0, HX.data_ptr() + k * M * 8)
    torch.cuda.synchronize()
    hx = HX.cpu().numpy()
    ob["value"] = hx.mean(axis=1) + np.random.default_rng(4000 + seed).standard_normal(P)
    out = {"workload": wl["desc"], "world": args.world, "splits": {}}
    for name in (("balanced", "equal") if loc == "GC" else ("equal",)):
        res = []
        for r in range(args.world):
            if name == "balanced":
                sh = ShardedEnSRF.balanced(eng, n_lead, ncol_g, M, ob, glat, glon, rank=r, world_size=args.world)
            else:
                sh = ShardedEnSRF(eng, n_lead, ncol_g, M, rank=r, world_size=args.world)
            rows, ncol_l = sh.rows_local, sh.hi - sh.lo
            X = eng.empty((rows, M))
            post = eng.empty((rows, M))
            for lead in range(n_lead):
                ctx.fill_synthetic(ncol_l, lead * ncol_g + sh.lo, M, seed, 3.0, X.data_ptr() + lead * ncol_l * M * 8)
            import time
            hx_copies = [HX.clone() for _ in range(args.steps + 1)]
            d = sh.assimilate(X, post, hx_copies[0], ob, glat, glon)     # warm-up (builds what depends on geometry only)
            torch.cuda.synchronize()
            ctx.last_timing()
            t0 = time.perf_counter()
            for it in range(args.steps):
                d = sh.assimilate(X, post, hx_copies[it + 1], ob, glat, glon)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3 / args.steps
            t = ctx.last_timing()
            st, ob_ms = t["state_ms"], t["obs_ms"]
            res.append(dict(rank=r, columns=[int(sh.lo), int(sh.hi)], rows=int(rows),
                            active_pairs=(float(ctx.get_option("gc_active_pairs")) if loc == "GC" else None),
                            state_phase_ms=st / args.steps, obs_phase_ms=ob_ms / args.steps, cycle_ms=wall))
            del hx_copies
            del X, post
            torch.cuda.empty_cache()
        stv = np.array([x["state_phase_ms"] for x in res])
        obv = np.array([x["obs_phase_ms"] for x in res])
        cyc = np.array([x["cycle_ms"] for x in res])
        out["splits"][name] = dict(per_rank=res, state_phase_max_ms=float(stv.max()), state_phase_mean_ms=float(stv.mean()),
                                   max_over_mean=float(stv.max() / stv.mean()), obs_phase_ms=float(obv.mean()),
                                   projected_cycle_ms=float(stv.max() + obv.mean()),
                                   cycle_max_ms=float(cyc.max()), cycle_mean_ms=float(cyc.mean()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
