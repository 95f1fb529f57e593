"""Fuzz of efa_ensrf_cycle_dev over SEQUENCES of cycles on one context: from one cycle to the next the geometry (obs positions, radii,
assimilate flags, grid, sizes) is kept or changed at random while values, error variances and the state always change -- what the
library keeps across cycles (obs-obs taper table, active lists, grid and stencil mirrors, the speculated transform, deferred timing)
must never leak from one cycle into a different one.  Every cycle is checked against the oracle.
usage: python tools/fuzz_cycle.py [cycles] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import test_gpu_parity as T
from oracle import ensrf_oracle as orc

ncyc = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 17
rng = np.random.default_rng(seed)
ctx = T._ctx()
ctx.set_option("timing", 2)
fails, worst, kept = [], 0.0, 0


def new_case():
    loc = bool(rng.random() < 0.6)
    M = int(rng.choice([4, 10, 20, 40, 64, 100, 104]))
    P = int(rng.integers(1, 400))
    if loc:
        n_lead, ncol = int(rng.integers(1, 6)), int(rng.integers(17, 300))
        N = n_lead * ncol
    else:
        N, ncol = int(rng.integers(1, 2000)), None
    return T._random_case(int(rng.integers(1 << 30)), N, M, P, loc, frac_assim=float(rng.choice([0.5, 0.9, 1.0])), ncol=ncol)


c = new_case()
t0 = time.time()
try:
    for it in range(ncyc):
        what = rng.random()
        if what < 0.25:
            c = new_case()
            tag = "new case"
        else:
            c = dict(c)
            N, M, P = c["N"], c["M"], c["P"]
            c["X"] = rng.standard_normal((N, 1)) + 3.0 * rng.standard_normal((N, M))
            rows = rng.choice(N, P, replace=(P > N))
            c["HX"] = c["X"][rows]
            c["val"] = c["HX"].mean(axis=1) + rng.standard_normal(P)
            c["err"] = rng.uniform(0.5, 2.0, P)
            tag = "same geometry"
            kept += 1
            if c["loc"] and what < 0.55:
                kept -= 1
                k = int(rng.integers(P))
                kind = int(rng.integers(4))
                if kind == 0:
                    c["ob_lat"] = c["ob_lat"].copy(); c["ob_lat"][k] += rng.uniform(-5, 5); tag = "ob moved"
                elif kind == 1:
                    c["asm"] = c["asm"].copy(); c["asm"][k] = not c["asm"][k]; tag = "flag flipped"
                elif kind == 2:
                    c["hw"] = c["hw"].copy(); c["hw"][k] *= rng.uniform(0.2, 3.0); tag = "radius changed"
                else:
                    c["lon"] = (c["lon"] + rng.uniform(-2, 2)) % 360.0; tag = "grid shifted"
            elif (not c["loc"]) and what < 0.45:
                c["asm"] = c["asm"].copy(); k = int(rng.integers(P)); c["asm"][k] = not c["asm"][k]; tag = "flag flipped"; kept -= 1
        N, M, P = c["N"], c["M"], c["P"]
        xam, Xap, diag = T._run_oracle(c)
        ref = orc.format_posterior_state(xam, Xap, N)
        inplace = bool(rng.random() < 0.2)
        ctx.set_option("path", int(rng.choice([0, 0, 1, 2])) if not c["loc"] else 0)
        X = ctx.to_device(c["X"])
        post = X if inplace else ctx.empty((N, M))
        Yp = ctx.to_device(c["HX"])
        ym = ctx.empty((P,))
        ctx.form_perts(P, M, Yp, ym, Yp)
        kw = {}
        if c["loc"]:
            kw = dict(loc_mode=1, ob_lat=c["ob_lat"], ob_lon=c["ob_lon"], ob_halfwidth=c["hw"], grid_lat=c["lat"].reshape(-1),
                      grid_lon=c["lon"].reshape(-1), n_lead=c["n_lead"])
        d = ctx.ensrf_cycle(N, M, P, X, post, ym, Yp, c["val"], c["err"], c["asm"], **kw)
        got = post.download()
        e = float(np.max(np.abs(got - ref))) / max(float(np.max(np.abs(ref))), 1e-300)
        for key in ("prior_var", "post_var", "post_mean"):
            r = np.asarray(diag[key], float); g = np.asarray(d[key], float); ok = np.isfinite(r)
            if ok.any():
                e = max(e, float(np.max(np.abs(g[ok] - r[ok]))) / max(float(np.max(np.abs(r[ok]))), 1e-300))
        worst = max(worst, e)
        line = "cycle %3d %-15s loc=%d M=%3d P=%3d N=%4d inplace=%d kind=%d rel err %.2e" % (it, tag, c["loc"], M, P, N, inplace, ctx.get_option("phase_a_kind"), e)
        if not (e < 1e-10) or not np.array_equal(d["assimilated"], diag["assimilated"]):
            fails.append(line)
            line += " FAIL"
        print(line, flush=True)
finally:
    ctx.set_option("path", 0)
    ctx.set_option("timing", 0)
print("fuzz cycle: %d cycles (%d on the previous cycle's geometry), %d failures, worst rel err %.2e, %.0f s" % (ncyc, kept, len(fails), worst, time.time() - t0))
for f in fails:
    print("  ", f)
sys.exit(1 if fails else 0)
