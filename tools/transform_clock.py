"""Diagnostic: shader clock during k_transform (s_memtime ticks per 10 ns s_memrealtime tick).
Needs:  make -C efa_xray_amd/csrc clean all EXTRA=-DEFA_T_CLOCKSTAMP"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from efa_xray_amd import _lib
ctx = _lib.get_context(0)
rows, M, P = 10_000_000, 100, 10_000
rng = np.random.default_rng(1)
X = ctx.empty((rows, M)); ctx.fill_synthetic(rows, 0, M, 99, 3.0, X)
xm = ctx.empty((rows,)); ctx.form_perts(rows, M, X, xm, X)
pick = np.sort(rng.choice(rows, P, replace=False)).astype(np.int64)
HX = ctx.empty((P, M)); ctx.forward_stencil(rows, 0, M, X, pick[:, None], np.ones((P, 1)), HX)
hx = HX.download(); val = rng.standard_normal(P); err = np.ones(P)
ctx.set_option("path", 2); ctx.set_option("timing", 1)
ym = ctx.empty((P,)); Yp = ctx.to_device(hx); ctx.form_perts(P, M, Yp, ym, Yp)
ctx.obs_phase(M, P, ym, Yp, val, err, np.ones(P, dtype=bool))
xo = ctx.empty((rows,)); Xo = ctx.empty((rows, M))
for _ in range(3):
    ctx.state_phase(rows, M, xm, X, xo, Xo)
    t = ctx.last_timing()
    v = xo.download_rows(0, 2)
    print("state_ms %.3f  s_memtime ticks %.0f  realtime ticks %.0f  -> shader clock %.3f GHz, block 0 busy %.3f ms" % (t["state_ms"], v[0], v[1], v[0] / v[1] * 0.1, v[1] * 1e-5))
