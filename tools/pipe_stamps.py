"""Diagnostic: cycle stamps of the Phase-A leader chain (pipe_debug bit 2)."""
# Needs the diagnostic build of the library:  make -C efa_xray_amd/csrc clean all STAMPS=1
# (cycle stamps and timing switches are compiled out of the normal build).
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from efa_xray_amd import _lib
ctx = _lib.get_context(0)
M, P = 100, 256
rng = np.random.default_rng(0)
ctx.set_option("path", 1); ctx.set_option("pipeline", 1); ctx.set_option("pipe_debug", 4)
HX = rng.standard_normal((P, M)) * 3
val = HX.mean(axis=1) + rng.standard_normal(P); err = np.ones(P); asm = np.ones(P, bool)
for _ in range(2):
    Yp = ctx.to_device(HX); ym = ctx.empty((P,))
    ctx.form_perts(P, M, Yp, ym, Yp)
    ctx.obs_phase(M, P, ym, Yp, val, err, asm)
addr = ctx.get_option("pipe_dbg_addr")
out = np.zeros((P, 8), dtype=np.uint64)
_lib._check(ctx.lib, ctx.lib.efa_memcpy_d2h(ctx.handle, out.ctypes.data, ctypes.c_void_p(addr), out.nbytes))
t = out.astype(np.int64)
names = ["top->ready_ye seen", "ye read+dot", "ready_sc seen+scalars", "update(+recurrence)", "publish", "vfresh/tail"]
good = [k for k in range(2, P - 2) if t[k, 0] > 0 and (k + 1) % 64 != 0]
d = np.array([[t[k, i + 1] - t[k, i] for i in range(6)] for k in good])
print("segment medians (s_memtime ticks) over %d steps:" % len(good))
for i, n in enumerate(names):
    print("  %-24s median %6.0f  p90 %6.0f" % (n, np.median(d[:, i]), np.percentile(d[:, i], 90)))
step = np.array([t[k + 1, 0] - t[k, 0] for k in good if t[k + 1, 0] > 0])
print("owner(k+1).top - owner(k).top (different lanes, same counter): median %.0f ticks" % np.median(step))
print("stamp 6 - stamp 0 (one owner's whole iteration): median %.0f" % np.median(t[good, 6] - t[good, 0]))
ctx.set_option("pipe_debug", 0); ctx.set_option("path", 0)
