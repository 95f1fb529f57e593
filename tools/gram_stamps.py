"""Diagnostic: cycle stamps of the Gram-space Phase-A leader (pipe_debug bit 2, option gram)."""
# Needs the diagnostic build of the library:  make -C efa_xray_amd/csrc clean all STAMPS=1
# (cycle stamps and timing switches are compiled out of the normal build).
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from efa_xray_amd import _lib
ctx = _lib.get_context(0)
M, P = 100, 512
rng = np.random.default_rng(0)
ctx.set_option("path", 1); ctx.set_option("pipeline", 1); ctx.set_option("gram", 1); ctx.set_option("pipe_debug", 4 | int(os.environ.get("EFA_EXP_BITS", "0")))
HX = rng.standard_normal((P, M)) * 3
val = HX.mean(axis=1) + rng.standard_normal(P); err = np.ones(P); asm = np.ones(P, bool)
for _ in range(2):
    Yp = ctx.to_device(HX); ym = ctx.empty((P,))
    ctx.form_perts(P, M, Yp, ym, Yp)
    ctx.obs_phase(M, P, ym, Yp, val, err, asm)
print("phase_a_kind", ctx.get_option("phase_a_kind"))
addr = ctx.get_option("pipe_dbg_addr")
out = np.zeros((P, 8), dtype=np.uint64)
_lib._check(ctx.lib, ctx.lib.efa_memcpy_d2h(ctx.handle, out.ctypes.data, ctypes.c_void_p(addr), out.nbytes))
t = out.astype(np.int64)
names = ["0 pivot step start", "1 pivot record published", "2 pivot next row ready", "3 chain wave saw step", "4 chain wave published ye",
         "5 helper got record", "6 helper handed row", "7 forwarded"]
inner = [k for k in range(P) if 4 <= k % 64 <= 59]
def med(x): return "median %6.0f p90 %6.0f" % (np.median(x), np.percentile(x, 90))
print("ticks of s_memtime (10 ns); step-to-step period of each stamp inside a block:")
for i, n in enumerate(names):
    d = np.array([t[k + 1, i] - t[k, i] for k in inner if t[k, i] > 0 and t[k + 1, i] > 0])
    if len(d): print("  %-28s %s" % (n, med(d)))
print("lags within a step (relative to pivot step start):")
for i in range(1, 8):
    d = np.array([t[k, i] - t[k, 0] for k in inner if t[k, i] > 0])
    if len(d): print("  %-28s %s" % (names[i], med(d)))
blk = [t[64 * b, 0] - t[64 * b - 1, 7] for b in range(1, P // 64)]
print("block hand-over (prev block's last forward -> next block's first pivot step):", blk)
print("block durations (first pivot step -> last forward):", [t[64 * b + 63, 7] - t[64 * b, 0] for b in range(P // 64)])
ctx.set_option("pipe_debug", 0); ctx.set_option("path", 0); ctx.set_option("gram", 0)
