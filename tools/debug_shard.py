import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from test_gpu_parity import _random_case, _run_hip, _ctx
from oracle import ensrf_oracle as orc
c = _random_case(33, 3 * 800, 50, 70, True, ncol=800)
a = _run_hip(c, path="sweep"); b = _run_hip(c, path="sweep")
print("repeat identical xam:", np.array_equal(a[0], b[0]), "Xap:", np.array_equal(a[1], b[1]))
for k in a[2]:
    print(" diag", k, np.array_equal(a[2][k], b[2][k], equal_nan=True))
ctx = _ctx()
N, M, P, ncol, L = c["N"], c["M"], c["P"], 800, 3
xbm, Xbp = orc.format_prior_state(c["X"], c["HX"])
lat, lon = c["lat"].reshape(-1), c["lon"].reshape(-1)
for lo, hi in ((0, 333), (333, 800), (0, 800), (16, 800), (0, 16), (0,17)):
    cols = np.arange(lo, hi)
    rows = (np.arange(L)[:, None] * ncol + cols[None, :]).reshape(-1)
    xs = np.ascontiguousarray(np.hstack((xbm[rows], xbm[N:])))
    Xs = np.ascontiguousarray(np.vstack((Xbp[rows], Xbp[N:])))
    ctx.set_option("path", 1)
    d = ctx.ensrf_update_host(xs, Xs, len(rows), c["val"], c["err"], c["asm"], loc_mode=1, ob_lat=c["ob_lat"],
                              ob_lon=c["ob_lon"], ob_halfwidth=c["hw"], grid_lat=lat[lo:hi], grid_lon=lon[lo:hi], n_lead=L)
    dx = xs[:len(rows)] != a[0][rows]
    dX = (Xs[:len(rows)] != a[1][rows]).any(axis=1)
    print("shard", lo, hi, "xm diff rows:", dx.sum(), "Xp diff rows:", dX.sum(), "obs rows equal:", np.array_equal(xs[len(rows):], a[0][N:]),
          "first bad:", np.nonzero(dx | dX)[0][:10], "maxrel", np.abs(xs[:len(rows)] - a[0][rows]).max())
    for k in d:
        if not np.array_equal(d[k], a[2][k], equal_nan=True): print("   diag differs:", k)
