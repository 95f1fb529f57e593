"""Diagnostic: block duration of the Gram-space leader with parts of the work switched off (timing only)."""
# Needs the diagnostic build of the library:  make -C efa_xray_amd/csrc clean all STAMPS=1
# (cycle stamps and timing switches are compiled out of the normal build).
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from efa_xray_amd import _lib
ctx = _lib.get_context(0)
M, P = 100, 512
rng = np.random.default_rng(0)
HX = rng.standard_normal((P, M)) * 3
val = HX.mean(axis=1) + rng.standard_normal(P); err = np.ones(P); asm = np.ones(P, bool)
ctx.set_option("path", 1); ctx.set_option("pipeline", 1); ctx.set_option("gram", 1)
for name, bits in (("full", 0), ("no helper FMAs", 16), ("no rank-4 MFMA", 32), ("no window update", 64), ("no vector work", 96), ("pivot only", 112), ("pivot only, long dozes", 112 | 512), ("full, long dozes", 512)):
    ctx.set_option("pipe_debug", 4 | 128 | 256 | bits)
    for _ in range(2):
        Yp = ctx.to_device(HX); ym = ctx.empty((P,))
        ctx.form_perts(P, M, Yp, ym, Yp)
        try:
            ctx.obs_phase(M, P, ym, Yp, val, err, asm)
        except Exception as e:
            print(name, "error", e)
    kind = ctx.get_option("phase_a_kind")
    addr = ctx.get_option("pipe_dbg_addr")
    out = np.zeros((P, 8), dtype=np.uint64)
    _lib._check(ctx.lib, ctx.lib.efa_memcpy_d2h(ctx.handle, out.ctypes.data, ctypes.c_void_p(addr), out.nbytes))
    t = out.astype(np.int64)
    dur = [t[64 * b + 63, 7] - t[64 * b, 0] for b in range(P // 64)]
    per = np.median([t[k + 1, 0] - t[k, 0] for k in range(P) if 4 <= k % 64 <= 59])
    print("%-18s kind %d  block %6.0f cycles  pivot period %5.0f" % (name, kind, np.median(dur), per))
    segs = np.array([[t[64 * b + i, 5] for i in range(4)] for b in range(P // 64)]) / 56.0
    print("      pivot segments (cycles/step): top->kb %.0f  publish %.0f  tail %.0f  loop-back %.0f" % tuple(np.median(segs, axis=0)))
ctx.set_option("pipe_debug", 0); ctx.set_option("path", 0); ctx.set_option("gram", 0)
