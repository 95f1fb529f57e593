"""Diagnostic: time Phase A alone for small P to separate leader-step cost from hand-off cost."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from efa_xray_amd import _lib
ctx = _lib.get_context(0)
ctx.set_option("timing", 1)
M = 100
rng = np.random.default_rng(0)
def run(P, path, pipeline, debug=0, reps=5):
    ctx.set_option("path", path); ctx.set_option("pipeline", pipeline); ctx.set_option("pipe_debug", debug)
    HX = rng.standard_normal((P, M)) * 3
    val = HX.mean(axis=1) + rng.standard_normal(P); err = np.ones(P); asm = np.ones(P, bool)
    best = 1e9
    for _ in range(reps):
        Yp = ctx.to_device(HX); ym = ctx.empty((P,))
        ctx.form_perts(P, M, Yp, ym, Yp)
        ctx.obs_phase(M, P, ym, Yp, val, err, asm)
        best = min(best, ctx.last_timing()["obs_ms"])
    return best
for P in (64, 128, 256, 1024, 4096):
    for name, path, pipe, dbg in (("pipeline sweep-path", 1, 1, 0), ("pipeline transform-path(+M rows)", 0, 1, 0), ("batch kernels", 1, 0, 0)):
        t = run(P, path, pipe, dbg)
        print("P=%5d %-34s %8.3f ms  %6.3f us/ob" % (P, name, t, 1e3 * t / P))
for name, dbg in (("single WG, full", 0), ("single WG, no global publish", 1), ("single WG, no prefetch loads", 2), ("single WG, neither", 3)):
    t = run(64, 1, 1, dbg)
    print("P=   64 %-34s %8.3f ms  %6.3f us/ob" % (name, t, 1e3 * t / 64))
ctx.set_option("pipe_debug", 0); ctx.set_option("path", 0)
