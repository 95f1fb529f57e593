"""Where a block's time goes inside the band leader's serial waves: `make -C efa_xray_amd/csrc pivstamp`, then
EFA_HIP_LIB=efa_xray_amd/libefa_hip_pivstamp.so python tools/band_pivstamp.py [P].
Per block of 64 obs (medians, shader cycles): the pivot wave's loop, the part of it spent waiting for the G waves'
rows; per G wave: waiting for the pivot's first two steps of a band, phase 1 (early rows), waiting for the band's end,
phase 2 (the rest of the band)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from efa_xray_amd import _lib
ctx = _lib.get_context(0)
M = 100
P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(0)
ctx.set_option("path", 2); ctx.set_option("pipeline", 1); ctx.set_option("gram", 2); ctx.set_option("pipe_debug", 4 | bits)
ctx.set_option("timing", 1)
HX = rng.standard_normal((P, M)) * 3
val = HX.mean(axis=1) + rng.standard_normal(P); err = np.ones(P); asm = np.ones(P, bool)
for _ in range(3):
    Yp = ctx.to_device(HX); ym = ctx.empty((P,))
    ctx.form_perts(P, M, Yp, ym, Yp)
    ctx.obs_phase(M, P, ym, Yp, val, err, asm)
addr = ctx.get_option("pipe_dbg_addr")
out = np.zeros((P, 8), dtype=np.uint64)
_lib._check(ctx.lib, ctx.lib.efa_memcpy_d2h(ctx.handle, out.ctypes.data, ctypes.c_void_p(addr), out.nbytes))
t = out.astype(np.int64)
nb = P // 64
med = lambda f: float(np.median([f(b) for b in range(1, nb)]))
print("bits %d" % bits, "kind %d obs_ms %.3f" % (ctx.get_option("phase_a_kind"), ctx.last_timing()["obs_ms"]))
print("pivot loop %.0f cycles per block (%.0f per step), of which waiting for the G waves' rows %.0f"
      % (med(lambda b: t[64 * b, 0]), med(lambda b: t[64 * b, 0]) / 64, med(lambda b: t[64 * b, 1])))
r = [med(lambda b, i=i: t[64 * b + 16, i]) for i in range(5)]
print("pivot loop per block: band head (ob constants, deferred-Gram wait) %.0f | LDS row loads after the wait %.0f | previous band applied (16 v_readlane pairs + FMA) %.0f | the four steps %.0f | L^-1 store + flags %.0f  (each interval ends in an s_memtime: +100-150 cycles per band each)"
      % (r[4], r[0], r[1], r[2], r[3]))
for h in (0, 1):
    r = [med(lambda b, i=i: t[64 * b + h, 2 + i]) for i in range(5)]
    print("G wave %d per block: wait first half %.0f | phase 1 %.0f (its operand loads %.0f) | wait band end %.0f | phase 2 %.0f  (sum %.0f)"
          % (h, r[0], r[1], r[4], r[2], r[3], sum(r[:4])))
# the hand-over chain between consecutive blocks, absolute s_memrealtime stamps (10 ns ticks) in row own0 + 8
rt = lambda b, c: int(t[64 * b + 8, c])
ch = []
ch2 = []
for b in range(2, nb - 2):
    T0, T1 = rt(b, 5), rt(b, 1)                                   # leader b: pivot done, last record forwarded
    T2, T3, T4, T5 = rt(b + 1, 0), rt(b + 1, 2), rt(b + 1, 3), rt(b + 1, 4)   # leader b+1
    if min(T0, T1, T2, T3, T4, T5) > 0:
        ch.append((T1 - T0, T2 - T1, T3 - T2, T4 - T3, T5 - T4, T5 - T0, rt(b + 1, 5) - T5))
        if rt(b, 6) > 0 and rt(b + 1, 7) > 0:
            ch2.append((rt(b, 6) - T0, rt(b + 1, 7) - rt(b, 6), T2 - rt(b + 1, 7)))
ch = np.array(ch) * 10.0 * 2.39   # -> shader cycles at 2.39 GHz
if len(ch):
    print("hand-over chain (median cycles): pivot done -> last record forwarded %.0f | -> in next leader's ring %.0f | -> its rows parked (B1) %.0f | -> Gram done (B2) %.0f | -> its pivot starts %.0f ;  total %.0f ; its pivot loop %.0f"
          % tuple(np.median(ch, axis=0)))
ow = []
for b in range(2, nb - 2):
    T0 = rt(b, 5)
    r9 = [int(t[64 * b + 9, c]) for c in range(3)]
    if T0 > 0 and min(r9) > 0 and rt(b, 6) > 0:
        ow.append((r9[0] - T0, r9[1] - T0, r9[2] - T0, rt(b, 6) - T0))
if len(ow):
    print("   owner wave of the last band, cycles after the pivot finished: ready for the band %.0f | saw the pivot's flag %.0f | has ring space %.0f | ye rows stored %.0f" % tuple(np.median(np.array(ow) * 10.0 * 2.39, axis=0)))
lag = []
for b in range(2, nb - 2):
    pv = [int(t[64 * b + 14, c]) for c in range(4)]
    vw = [[int(t[64 * b + 10 + w, c]) for c in range(4)] for w in range(4)]
    if min(pv) > 0 and min(min(v) for v in vw) > 0:
        lag.append([[vw[w][c] - pv[c] for c in range(4)] for w in range(4)])
if len(lag):
    L = np.median(np.array(lag) * 10.0 * 2.39, axis=0)
    for w in range(4):
        print("   vector wave %d is through with bands 3 / 7 / 11 / 15 this many cycles after the pivot: %s" % (w, np.round(L[w])))
if len(ch2):
    c2 = np.median(np.array(ch2) * 10.0 * 2.39, axis=0)
    print("   of which: pivot done -> last band's ye rows stored by its owner wave %.0f | -> seen complete by the next leader's loader %.0f | -> in its ring (ring space, LDS writes) %.0f" % tuple(c2))
# per band (rows own0+64+b of the NEXT block's stamp rows hold band b of block own0): absolute s_memtime stamps
ev = []
for blk in range(2, nb - 2):
    for b in range(1, 15):
        r = t[64 * blk + 64 + b]
        rp = t[64 * blk + 64 + b + 1]      # next band: its wait start / rows received
        # r[0] cHalf set (band b), r[3] G wave saw it, r[4] G wave set the rows flag; rp[1] pivot starts waiting for band b+1's rows, rp[2] got them
        if r[0] and r[3] and r[4] and rp[1] and rp[2]:
            ev.append((r[3] - r[0], r[4] - r[3], rp[1] - r[0], rp[2] - r[4], rp[2] - rp[1]))
ev = np.array(ev)
if len(ev):
  print("per band (median cycles): cHalf set -> G wave sees it %.0f | -> rows flag set %.0f | cHalf set -> pivot starts waiting %.0f | flag set -> pivot sees rows %.0f | pivot waits %.0f"
      % tuple(np.median(ev, axis=0)))
ctx.set_option("pipe_debug", 0); ctx.set_option("path", 0)
