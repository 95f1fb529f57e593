"""Diagnostic: single-batch (P=64) Phase A time vs ensemble size, both implementations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from efa_xray_amd import _lib
ctx = _lib.get_context(0)
ctx.set_option("timing", 1)
rng = np.random.default_rng(0)
def run(P, M, pipeline, debug=0, reps=7):
    ctx.set_option("path", 1); ctx.set_option("pipeline", pipeline); ctx.set_option("pipe_debug", debug)
    HX = rng.standard_normal((P, M)) * 3
    val = HX.mean(axis=1) + rng.standard_normal(P); err = np.ones(P); asm = np.ones(P, bool)
    best = 1e9
    for _ in range(reps):
        Yp = ctx.to_device(HX); ym = ctx.empty((P,))
        ctx.form_perts(P, M, Yp, ym, Yp)
        ctx.obs_phase(M, P, ym, Yp, val, err, asm)
        best = min(best, ctx.last_timing()["obs_ms"])
    return best
for M in (4, 16, 32, 64, 100, 128, 200, 256):
    a = run(64, M, 0); b = run(64, M, 1, 3); c = run(1, M, 0); d = run(32, M, 0)
    print("M=%3d  diag P=64: %7.1f us  P=32: %7.1f us  P=1: %7.1f us | pipe(1 WG, no global) P=64: %7.1f us" % (M, a*1e3, d*1e3, c*1e3, b*1e3))
ctx.set_option("pipe_debug", 0); ctx.set_option("path", 0); ctx.set_option("pipeline", 1)
