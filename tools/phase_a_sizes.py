import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from efa_xray_amd import _lib
ctx = _lib.get_context(0)
M = 100
rng = np.random.default_rng(0)
ctx.set_option("path", 2); ctx.set_option("pipeline", 1); ctx.set_option("gram", 2); ctx.set_option("timing", 1)
for P in (16, 64, 100, 128, 640, 2000, 10000):
    HX = rng.standard_normal((P, M)) * 3
    val = HX.mean(axis=1) + rng.standard_normal(P); err = np.ones(P); asm = np.ones(P, bool)
    for _ in range(2):
        Yp = ctx.to_device(HX); ym = ctx.empty((P,))
        ctx.form_perts(P, M, Yp, ym, Yp)
        ctx.obs_phase(M, P, ym, Yp, val, err, asm)
    print(P, "kind", ctx.get_option("phase_a_kind"), "obs_ms %.3f" % ctx.last_timing()["obs_ms"], flush=True)
