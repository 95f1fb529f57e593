"""Diagnostic: Phase-A time of the Gram leader, pipe_debug bit 1 off/on (A/B switch for experiments)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from efa_xray_amd import _lib
ctx = _lib.get_context(0)
M, P = 100, 10000
rng = np.random.default_rng(0)
HX = rng.standard_normal((P, M)) * 3
val = HX.mean(axis=1) + rng.standard_normal(P); err = np.ones(P); asm = np.ones(P, bool)
ctx.set_option("timing", 1); ctx.set_option("pipeline", 1); ctx.set_option("gram", 1)
for name, bits in (("A (bit 1 off)", 0), ("B (bit 1 on)", 1), ("A (bit 1 off)", 0), ("B (bit 1 on)", 1)):
    ctx.set_option("pipe_debug", bits)
    ts = []
    for _ in range(5):
        Yp = ctx.to_device(HX); ym = ctx.empty((P,))
        ctx.form_perts(P, M, Yp, ym, Yp)
        ctx.obs_phase(M, P, ym, Yp, val, err, asm)
        ts.append(ctx.last_timing()["obs_ms"])
    print("%-16s kind %d  obs_phase %.3f ms" % (name, ctx.get_option("phase_a_kind"), min(ts)))
ctx.set_option("pipe_debug", 0)
