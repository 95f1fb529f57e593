"""Diagnostic: Phase-A wall time at the headline obs count with parts of the leader's work switched off
(timing only; results are wrong with any bit set)."""
# Needs the diagnostic build of the library:  make -C efa_xray_amd/csrc clean all STAMPS=1
# (cycle stamps and timing switches are compiled out of the normal build).
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from efa_xray_amd import _lib
ctx = _lib.get_context(0)
M = 100
P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rng = np.random.default_rng(0)
HX = rng.standard_normal((P, M)) * 3
val = HX.mean(axis=1) + rng.standard_normal(P); err = np.ones(P); asm = np.ones(P, bool)
ctx.set_option("timing", 1); ctx.set_option("pipeline", 1)
for name, gram, bits in (("classic", 0, 0), ("gram full", 1, 0), ("gram, no helper FMAs", 1, 16), ("gram, no vector work", 1, 96),
                         ("gram, pivot only", 1, 112), ("gram, followers idle", 1, 1024), ("gram, pivot only + followers idle", 1, 112 | 1024),
                         ("gram, protocol only", 1, 112 | 2048), ("gram, protocol only + followers idle", 1, 112 | 1024 | 2048)):
    ctx.set_option("gram", gram); ctx.set_option("pipe_debug", bits)
    ts = []
    for _ in range(4):
        Yp = ctx.to_device(HX); ym = ctx.empty((P,))
        ctx.form_perts(P, M, Yp, ym, Yp)
        ctx.obs_phase(M, P, ym, Yp, val, err, asm)
        ts.append(ctx.last_timing()["obs_ms"])
    print("%-24s kind %d  obs_phase %.3f ms  (%.0f cycles/ob at 2.39 GHz)" % (name, ctx.get_option("phase_a_kind"), min(ts), min(ts) * 2.39e6 / P))
ctx.set_option("pipe_debug", 0); ctx.set_option("gram", 0)
