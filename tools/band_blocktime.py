"""CAUTION (end of round 2): k_pipe_band<13> sits at the edge of the register file -- the product build uses 235 VGPRs
without spills, the diagnostic build (a few stamps and the dbg pointer more) spills 80 and its FOLLOWERS slow down by 2x, which
inflates every hand-over figure below.  Check `.vgpr_spill_count` of the diag library (tests/_codeobj.py) before trusting it.

Diagnostic: per-block times of the band leader (k_pipe_band).  Needs `make -C efa_xray_amd/csrc diag` and
EFA_HIP_LIB=efa_xray_amd/libefa_hip_diag.so.   usage: band_blocktime.py [P] [debug bits] [gram option]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from efa_xray_amd import _lib
ctx = _lib.get_context(0)
M = 100
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 0
gram = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rng = np.random.default_rng(0)
ctx.set_option("path", 1); ctx.set_option("pipeline", 1); ctx.set_option("gram", gram); ctx.set_option("pipe_debug", 4 | bits)
ctx.set_option("timing", 1)
HX = rng.standard_normal((P, M)) * 3
val = HX.mean(axis=1) + rng.standard_normal(P); err = np.ones(P); asm = np.ones(P, bool)
for _ in range(3):
    Yp = ctx.to_device(HX); ym = ctx.empty((P,))
    ctx.form_perts(P, M, Yp, ym, Yp)
    try:
        ctx.obs_phase(M, P, ym, Yp, val, err, asm)
    except Exception as e:
        print("error", e)
addr = ctx.get_option("pipe_dbg_addr")
out = np.zeros((P, 8), dtype=np.uint64)
_lib._check(ctx.lib, ctx.lib.efa_memcpy_d2h(ctx.handle, out.ctypes.data, ctypes.c_void_p(addr), out.nbytes))
t = out.astype(np.int64)
nb = P // 64
piv = [t[64 * b, 1] - t[64 * b, 0] for b in range(1, nb)]
blk = [t[64 * b, 2] - t[64 * b, 0] for b in range(1, nb)]
pre = [t[64 * b, 0] - t[64 * b, 3] for b in range(1, nb)]
print("bits %d gram %d kind %d obs_ms %.3f" % (bits, gram, ctx.get_option("phase_a_kind"), ctx.last_timing()["obs_ms"]))
print("  pivot loop %6.0f cycles/step | pivot start -> last record forwarded %6.0f /step | last foreign record -> pivot start %6.0f cycles"
      % (np.median(piv) / 64, np.median(blk) / 64, np.median(pre)))
if gram == 2:
    for w in range(8):
        print("    vector wave %d reaches its block %6.0f cycles after the last record is in the ring" % (w, np.median([t[64 * b + w, 4] - t[64 * b, 3] for b in range(1, nb)])))
    a = [t[64 * b, 4] - t[64 * b, 3] for b in range(1, nb)]
    b1 = [t[64 * b, 5] - t[64 * b, 4] for b in range(1, nb)]
    c1 = [t[64 * b, 0] - t[64 * b, 5] for b in range(1, nb)]
    print("  hand-over split: last record in ring -> followers done %6.0f | park + B1 %6.0f | Gram, tile load, B2, band load %6.0f"
          % (np.median(a), np.median(b1), np.median(c1)))
    med = lambda f: np.median([f(b) for b in range(1, nb)])
    print("  leader block: pivot start -> pivot end %6.0f | -> vector wave 0 through its bands %6.0f | -> last record forwarded %6.0f"
          % (med(lambda b: t[64 * b, 1] - t[64 * b, 0]), med(lambda b: t[64 * b, 6] - t[64 * b, 0]), med(lambda b: t[64 * b, 2] - t[64 * b, 0])))
    print("  records still to be applied by the slowest vector wave when the last foreign record is in the ring: median %.0f (min %d, max %d)"
          % (med(lambda b: t[64 * b + 3, 5]), min(t[64 * b + 3, 5] for b in range(1, nb)), max(t[64 * b + 3, 5] for b in range(1, nb))))
    # absolute times (s_memrealtime, 10 ns ticks, comparable across workgroups): block b's last record forwarded -> block b+1
    rt = lambda b, r, c: int(t[64 * b + r, c])
    ch = [(rt(b + 1, 4, 6) - rt(b, 4, 5), rt(b + 1, 4, 7) - rt(b + 1, 4, 6), rt(b + 1, 5, 5) - rt(b + 1, 4, 7), rt(b + 1, 4, 5) - rt(b + 1, 5, 5),
           rt(b + 1, 4, 5) - rt(b, 4, 5)) for b in range(1, nb - 1)]
    ch = np.array(ch) * 10.0
    print("  chain between blocks, ns (median): last record forwarded -> in the next leader's ring %5.0f | -> its vector waves at their block %5.0f"
          " | -> its pivot starts %5.0f | -> its last record forwarded %5.0f ; block period %5.0f"
          % tuple(np.median(ch, axis=0)))
ctx.set_option("pipe_debug", 0); ctx.set_option("path", 0); ctx.set_option("gram", 2)
