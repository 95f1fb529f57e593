import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np
from efa_xray_amd import _lib
ctx = _lib.get_context(0)
M = 100; P = 64
rng = np.random.default_rng(0)
ctx.set_option("path", 2); ctx.set_option("pipeline", 1); ctx.set_option("gram", 2); ctx.set_option("pipe_debug", 4); ctx.set_option("timing", 1)
HX = rng.standard_normal((P, M)) * 3
val = HX.mean(axis=1) + rng.standard_normal(P); err = np.ones(P); asm = np.ones(P, bool)
Yp = ctx.to_device(HX); ym = ctx.empty((P,))
ctx.form_perts(P, M, Yp, ym, Yp)
ctx.obs_phase(M, P, ym, Yp, val, err, asm)
print("kind", ctx.get_option("phase_a_kind"), ctx.last_timing()["obs_ms"])
addr = ctx.get_option("pipe_dbg_addr")
out = np.zeros((P, 8), dtype=np.uint64)
_lib._check(ctx.lib, ctx.lib.efa_memcpy_d2h(ctx.handle, out.ctypes.data, ctypes.c_void_p(addr), out.nbytes))
for w in range(8): print("wave", w, out[32 + w, :4])

taddr = ctx.get_option("traj_addr")
rec = np.zeros((4, 108), dtype=np.uint64)
_lib._check(ctx.lib, ctx.lib.efa_memcpy_d2h(ctx.handle, rec.ctypes.data, ctypes.c_void_p(taddr), rec.nbytes))
sent = rec == rec.max()
for k in range(4):
    print("record", k, "words still sentinel:", np.nonzero(rec[k] > np.uint64(0x7ff0000000000000))[0][:20], "last words", rec[k, 100:].view(np.float64))
