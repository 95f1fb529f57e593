"""Diagnostic: the state transform (member form, efa_ensrf_cycle_dev's Phase B) over ensemble sizes at a fixed 8 GB state:
ms per launch, physical HBM fraction (16 x rows x M bytes / 8 TB/s) and fp64 TFLOP/s (2 x rows x M^2)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
from efa_xray_amd import _lib

ctx = _lib.get_context(0)
ctx.set_option("timing", 1)
ctx.set_option("path", 2)
rng = np.random.default_rng(0)
for M in (16, 20, 32, 40, 50, 64, 80, 96, 100, 104, 112, 120, 128, 136, 160, 200, 256):
    rows = int(8e9 / 8 / M)
    P = 64
    X = ctx.empty((rows, M))
    post = ctx.empty((rows, M))
    ctx.fill_synthetic(rows, 0, M, 7, 3.0, X)
    HX = rng.standard_normal((P, M)) * 3
    val = HX.mean(axis=1) + rng.standard_normal(P)
    ts = []
    for _ in range(4):
        Yp = ctx.to_device(HX); ym = ctx.empty((P,))
        ctx.form_perts(P, M, Yp, ym, Yp)
        ctx.obs_phase(M, P, ym, Yp, val, np.ones(P), np.ones(P, bool))
        ctx.state_cycle(rows, M, X, post)
        ts.append(ctx.last_timing()["state_ms"])
    t = min(ts[1:])
    print("M %3d rows %9d  %.3f ms  HBM frac %.3f  fp64 %.1f TF" % (M, rows, t, 16.0 * rows * M / (t * 1e-3) / 8e12, 2.0 * rows * M * M / (t * 1e-3) / 1e12), flush=True)
    del X, post
ctx.set_option("path", 0)
ctx.set_option("timing", 0)
