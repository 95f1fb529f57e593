"""Diagnostic: LDS round-trip latency seen by a helper wave of the Gram leader under the real load."""
# Needs the diagnostic build of the library:  make -C efa_xray_amd/csrc clean all STAMPS=1
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from efa_xray_amd import _lib
ctx = _lib.get_context(0)
M, P = 100, 512
rng = np.random.default_rng(0)
ctx.set_option("path", 1); ctx.set_option("pipeline", 1); ctx.set_option("gram", 1); ctx.set_option("pipe_debug", 4 | 128 | 4096)
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
ks = [k for k in range(P) if 4 <= k % 64 <= 55 and t[k, 1] > 0]
a = t[ks, 1]; b = t[ks, 2]
print("kind", ctx.get_option("phase_a_kind"), "samples", len(ks))
print("ds_read_b32 round trip + s_memtime: median %.0f p10 %.0f p90 %.0f" % (np.median(a), np.percentile(a, 10), np.percentile(a, 90)))
print("s_memtime alone:                    median %.0f p10 %.0f p90 %.0f" % (np.median(b), np.percentile(b, 10), np.percentile(b, 90)))
print("=> LDS round trip ~ %.0f cycles" % (np.median(a) - np.median(b)))
ctx.set_option("pipe_debug", 0); ctx.set_option("path", 0)
