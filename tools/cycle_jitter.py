"""Diagnostic: host-side time between the returns of consecutive headline cycles (the host waits once per cycle, for Phase A's
results, so the deltas are the cycle times): median, percentiles and the largest outliers over a long run."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import bench
from efa_xray_amd.distributed import ShardedEnSRF, HipEngine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
wl = bench.WORKLOADS["headline"]
M, P = wl["M"], wl["P"]
eng = HipEngine(0)
ctx = eng.ctx
rows = wl["rows"]
sh = ShardedEnSRF(eng, 1, rows, M)
X = eng.empty((rows, M)); post = eng.empty((rows, M))
ctx.fill_synthetic(rows, 0, M, 1000, 3.0, X.data_ptr())
rng = np.random.default_rng(3000)
pick = rng.choice(rows, P, replace=False).astype(np.int64)
idx = pick[:, None].copy(); wts = np.ones((P, 1))
ob = dict(value=None, error=np.ones(P), assim=np.ones(P, dtype=bool))
HX0 = sh.partial_estimates(X, idx, wts); torch.cuda.synchronize()
ob["value"] = HX0.cpu().numpy().mean(axis=1) + rng.standard_normal(P)
for _ in range(5):
    sh.update(X, post, idx, wts, ob)
torch.cuda.synchronize()
import gc
for label, gc_on in (("gc enabled", True), ("gc disabled", False)):
    if not gc_on:
        gc.disable()
    ts = [time.perf_counter()]
    for _ in range(steps):
        sh.update(X, post, idx, wts, ob)
        ts.append(time.perf_counter())
    torch.cuda.synchronize()
    tot = (time.perf_counter() - ts[0]) * 1e3 / steps
    d = np.diff(np.array(ts)) * 1e3
    print("%s: mean %.3f ms/cycle; deltas median %.3f p90 %.3f p99 %.3f max %.3f; ten largest %s" % (
        label, tot, np.median(d), np.percentile(d, 90), np.percentile(d, 99), d.max(), np.round(np.sort(d)[-10:], 2)))
    gc.enable()
