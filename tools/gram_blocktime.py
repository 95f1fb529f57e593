"""Diagnostic: per-block times of the Gram leader with the production loop code.
Needs:  make -C efa_xray_amd/csrc clean all EXTRA=-DEFA_PIPE_BLOCKTIME   (stamps only outside the loops)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from efa_xray_amd import _lib
ctx = _lib.get_context(0)
M = 100
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(0)
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx.set_option("path", 1); ctx.set_option("pipeline", 1); ctx.set_option("gram", 1); ctx.set_option("pipe_debug", 4 | bits)
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
nb = P // 64
first = 0 if nb == 1 else 1
piv = [t[64 * b, 1] - t[64 * b, 0] for b in range(first, nb)]
blk = [t[64 * b, 2] - t[64 * b, 0] for b in range(first, nb)]
pre = [t[64 * b, 0] - t[64 * b, 3] for b in range(first, nb)]
print("bits", bits, "kind", ctx.get_option("phase_a_kind"))
print("pivot loop (64 steps):               median %7.0f cycles = %5.0f per step" % (np.median(piv), np.median(piv) / 64))
print("pivot start -> last record forwarded: median %7.0f cycles = %5.0f per step" % (np.median(blk), np.median(blk) / 64))
print("last foreign record seen -> pivot start (park, Gram, barriers): median %7.0f cycles" % np.median(pre))
print("steps per block whose hand-over row was late: median %d" % np.median([t[64 * b, 4] for b in range(first, nb)]))
rt = np.array([t[64 * b, 7] - t[64 * b, 6] for b in range(first, nb)], dtype=float)
if rt.min() > 0:
    print("s_memtime ticks per s_memrealtime tick over the pivot loop: median %.2f (x 100 MHz = clock of s_memtime)" % np.median(np.array(piv) / rt))
ctx.set_option("pipe_debug", 0); ctx.set_option("path", 0)
