"""Diagnostic: helper-wave detail stamps of the Gram-space leader (pipe_debug 4|8)."""
# Needs the diagnostic build of the library:  make -C efa_xray_amd/csrc clean all STAMPS=1
# (cycle stamps and timing switches are compiled out of the normal build).
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from efa_xray_amd import _lib
ctx = _lib.get_context(0)
M, P = 100, 512
rng = np.random.default_rng(0)
ctx.set_option("path", 1); ctx.set_option("pipeline", 1); ctx.set_option("gram", 1); ctx.set_option("pipe_debug", 12)
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
for lo, hi in ((4, 26), (32, 58)):
    ks = [k for k in range(P) if lo <= k % 64 <= hi and t[k, 3] > 0 and t[k, 6] > 0]
    print("steps %d..%d of a block (helper 0):" % (lo, hi))
    for a, b, n in ((0, 1, "pivot start->publish"), (1, 3, "publish->helper saw"), (3, 4, "own pair + t"), (4, 5, "half 0"), (5, 6, "half 1"), (0, 2, "pivot whole step")):
        d = np.array([t[k, b] - t[k, a] for k in ks])
        print("   %-22s median %6.0f p90 %6.0f" % (n, np.median(d), np.percentile(d, 90)))
    d = np.array([t[k + 1, 3] - t[k, 3] for k in ks if t[k + 1, 3] > 0])
    print("   helper period          median %6.0f" % np.median(d))
ctx.set_option("pipe_debug", 0); ctx.set_option("path", 0); ctx.set_option("gram", 0)
