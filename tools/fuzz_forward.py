"""Fuzz of the device forward operator (efa_interp_stencils + efa_forward_interp_dev) against the host restatement
of the reference's nearest_points / interpolate (EnsembleState.interp_stencil, pinned by fixtures G9/G10): random jittered regional
and global grids, random valid times, obs anywhere in and around the domain (dateline, high latitudes), random variables and times.
usage: python tools/fuzz_forward.py [cases] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
from efa_xray_amd import EnsembleState, Observation, _lib

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 9
rng = np.random.default_rng(seed)
ctx = _lib.get_context(0)
fails = []
worst = 0.0
nobs = 0
t0 = time.time()
for it in range(ncases):
    ny, nx, nt, nvar, M = int(rng.integers(2, 40)), int(rng.integers(2, 50)), int(rng.integers(1, 5)), int(rng.integers(1, 4)), int(rng.integers(2, 30))
    glob = rng.random() < 0.3
    if glob:
        la = np.linspace(-88, 88, ny)
        lo = np.linspace(0, 360 - 360.0 / nx, nx)
    else:
        c_la, c_lo = rng.uniform(-60, 60), rng.uniform(0, 360)
        la = c_la + np.linspace(-1, 1, ny) * rng.uniform(2, 20)
        lo = (c_lo + np.linspace(-1, 1, nx) * rng.uniform(2, 30)) % 360.0
    lat, lon = np.meshgrid(la, lo, indexing="ij")
    # jitter: no two grid points at the same pseudo-distance from an ob (ties are order-dependent in the reference's argsort)
    lat = lat + 1e-3 * rng.standard_normal(lat.shape)
    lon = (lon + 1e-3 * rng.standard_normal(lon.shape)) % 360.0
    times = np.cumsum(rng.uniform(600, 7200, nt))
    arr = rng.standard_normal((nvar, nt, ny, nx, 1)) + 2.0 * rng.standard_normal((nvar, nt, ny, nx, M))
    state = EnsembleState.from_array(arr, lat, lon, validtime=times)
    names = state.vars()
    P = int(rng.integers(1, 120))
    obs = []
    for k in range(P):
        if glob:
            olat, olon = rng.uniform(-89.5, 89.5), rng.uniform(0, 360)
        else:
            olat = float(np.clip(rng.uniform(lat.min() - 3, lat.max() + 3), -89.9, 89.9))
            olon = float(rng.uniform(0, 360)) if rng.random() < 0.1 else float((lon[ny // 2, int(rng.integers(nx))] + rng.uniform(-3, 3)) % 360.0)
        tsel = rng.random()
        tt = float(times[int(rng.integers(nt))]) if tsel < 0.4 else float(rng.uniform(times[0], times[-1]))
        obs.append(Observation(value=0.0, obtype=names[int(rng.integers(nvar))], time=tt, error=1.0, lat=float(olat), lon=float(olon),
                               assimilate_this=True, localize_radius=1000.0))
    tag = "case %2d %s ny=%d nx=%d nt=%d nvar=%d M=%d P=%d" % (it, "global" if glob else "regional", ny, nx, nt, nvar, M, P)
    try:
        idx, wts, st = ctx.interp_stencils(nvar, nt, ny, nx, state.coords["lat"], state.coords["lon"], state.ensemble_times(),
                                           [names.index(o.obtype) for o in obs], [o.time for o in obs],
                                           [o.lat for o in obs], [o.lon for o in obs])
        assert not st.any(), "status %r" % st
        e = 0.0
        for k, o in enumerate(obs):
            rows, w = o.stencil(state)
            got = {}
            for r, v in zip(idx[k], wts[k]):
                if r >= 0:
                    got[int(r)] = got.get(int(r), 0.0) + float(v)
            ref = {}
            for r, v in zip(rows, w):
                ref[int(r)] = ref.get(int(r), 0.0) + float(v)
            keys = sorted(set(got) | set(ref))
            ga = np.array([got.get(r, 0.0) for r in keys])
            ra = np.array([ref.get(r, 0.0) for r in keys])
            e = max(e, float(np.max(np.abs(ga - ra))) / max(float(np.max(np.abs(ra))), 1e-300))
        X = state.to_vect()
        Xd = ctx.to_device(X)
        HX = ctx.empty((P, M))
        ctx.forward_interp(ny * nx, 0, ny * nx, nvar * nt, M, Xd, HX)
        refhx = np.array([o.estimate(state) for o in obs])
        e2 = float(np.max(np.abs(HX.download() - refhx))) / max(float(np.max(np.abs(refhx))), 1e-300)
        worst = max(worst, e, e2)
        nobs += P
        okk = e < 1e-10 and e2 < 1e-10
        if not okk:
            fails.append(tag + " stencil err %.2e estimate err %.2e" % (e, e2))
        print("%s stencil %.2e estimate %.2e %s" % (tag, e, e2, "ok" if okk else "FAIL"), flush=True)
    except Exception as ex:  # noqa: BLE001
        fails.append(tag + " " + repr(ex)[:300])
        print(tag, "EXCEPTION", repr(ex)[:300], flush=True)
print("fuzz forward: %d cases (%d obs), %d failures, worst rel err %.2e, %.0f s" % (ncases, nobs, len(fails), worst, time.time() - t0))
for f in fails:
    print("  ", f)
sys.exit(1 if fails else 0)
