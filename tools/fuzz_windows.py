"""Fuzz of Phase A IN WINDOWS (more observations than one persistent launch holds) against the per-batch kernels (which the test suite
checks against the oracle): random member counts, ob counts at and around the window limits (one window = 256 x 64 rows minus the
carried transform rows), with and without Gaspari-Cohn, random assimilate fractions; without localisation also the transform the
windows leave behind against the per-batch sweep on 300 state rows.
usage: python tools/fuzz_windows.py [cases] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import test_gpu_parity as T

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(seed)
ctx = T._ctx()
fails = []
worst = 0.0


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    ok = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), ok)
    if not ok.any():
        return 0.0
    return float(np.max(np.abs(a[ok] - b[ok]))) / max(float(np.max(np.abs(b[ok]))), 1e-300)


t0 = time.time()
try:
    for it in range(ncases):
        loc = bool(rng.random() < 0.4)
        M = int(rng.choice([8, 20, 40, 50, 64, 100, 104, 128]))
        extra = 0 if loc else M
        w_one = 256 * 64 - extra
        w_max = w_one - extra
        P = int(rng.choice([w_one, w_one + 1, w_one + 63, w_one + 64, w_one + 65, 2 * w_max, 2 * w_max + 1, 2 * w_max - 1,
                            int(rng.integers(w_one + 1, 45000))]))
        frac = float(rng.choice([0.5, 0.9, 1.0]))
        HX = 3.0 * rng.standard_normal((P, M)) + rng.standard_normal((P, 1))
        val = HX.mean(axis=1) + rng.standard_normal(P)
        err = rng.uniform(0.5, 2.0, P)
        asm = rng.random(P) < frac
        kw = {}
        if loc:
            kw = dict(loc_mode=1, ob_lat=rng.uniform(-60, 60, P), ob_lon=rng.uniform(0, 360, P), ob_halfwidth=rng.uniform(300, 900, P))
        X = rng.standard_normal((300, M))
        res = {}
        tag = "case %2d loc=%d M=%3d P=%5d (one window %d, later windows %d) assim=%.2f" % (it, loc, M, P, w_one, w_max, frac)
        try:
            for name, pipe in (("batch", 0), ("windows", 1)):
                ctx.set_option("pipeline", pipe)
                ctx.set_option("gram", T.GRAM_DEFAULT)
                ctx.set_option("path", 1 if (loc or pipe == 0) else 2)
                Yp = ctx.to_device(HX)
                ym = ctx.empty((P,))
                ctx.form_perts(P, M, Yp, ym, Yp)
                d = ctx.obs_phase(M, P, ym, Yp, val, err, asm, **kw)
                kind = ctx.get_option("phase_a_kind")
                out = [Yp.download(), ym.download(), d, kind]
                if not loc:
                    xm = ctx.to_device(X.mean(axis=1))
                    Xp = ctx.to_device(X - X.mean(axis=1, keepdims=True))
                    ctx.state_phase(300, M, xm, Xp, xm, Xp)
                    out += [xm.download(), Xp.download()]
                res[name] = out
            a, b = res["windows"], res["batch"]
            e = max(rel(a[0], b[0]), rel(a[1], b[1]))
            for key in ("prior_mean", "prior_var", "post_mean", "post_var"):
                e = max(e, rel(a[2][key], b[2][key]))
            assert np.array_equal(a[2]["assimilated"], b[2]["assimilated"])
            if not loc:
                e = max(e, rel(a[4], b[4]), rel(a[5], b[5]))
            worst = max(worst, e)
            okk = e < 1e-10
            if not okk:
                fails.append(tag + " err %.2e" % e)
            print("%s kinds %d/%d rel err %.2e %s" % (tag, a[3], b[3], e, "ok" if okk else "FAIL"), flush=True)
        except Exception as ex:  # noqa: BLE001
            fails.append(tag + " " + repr(ex)[:300])
            print(tag, "EXCEPTION", repr(ex)[:300], flush=True)
finally:
    ctx.set_option("pipeline", 1)
    ctx.set_option("gram", T.GRAM_DEFAULT)
    ctx.set_option("path", 0)
print("fuzz windows: %d cases, %d failures, worst rel err %.2e, %.0f s" % (ncases, len(fails), worst, time.time() - t0))
for f in fails:
    print("  ", f)
sys.exit(1 if fails else 0)
