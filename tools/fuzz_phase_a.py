"""Wider fuzz of Phase A + both Phase-B paths against the oracle than tests/test_gpu_parity.py keeps in the suite: random member and ob
counts, localised and not, random assimilate fractions (incl. 0 and 1), error variances over six decades, correlated / duplicated
obs rows (the Gram-space cancellation guard), every persistent leader and the per-batch kernels.
usage: python tools/fuzz_phase_a.py [cases] [seed]      (prints one line per case, a summary, exits non-zero on any failure)"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import test_gpu_parity as T

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 31337
rng = np.random.default_rng(seed)
ctx = T._ctx()
fails = []
worst = 0.0
t0 = time.time()
for it in range(ncases):
    loc = bool(rng.random() < 0.45)
    M = int(rng.choice([2, 3, 5, 8, 13, 20, 33, 50, 64, 80, 96, 99, 100, 101, 104, 105, 112, 128]))
    P = int(rng.choice([1, 2, 3, 4, 5, 63, 64, 65, 127, 128, 129, int(rng.integers(1, 1000))]))
    if loc:
        n_lead, ncol = int(rng.integers(1, 9)), int(rng.integers(17, 120))
        N = n_lead * ncol
    else:
        N, ncol = int(rng.integers(1, 400)), None
    frac = float(rng.choice([0.0, 0.3, 0.75, 0.95, 1.0]))
    c = T._random_case(int(rng.integers(1 << 30)), N, M, P, loc, frac_assim=frac, ncol=ncol)
    mode = int(rng.integers(0, 4))
    if mode == 1:      # error variances over six decades
        c["err"] = 10.0 ** rng.uniform(-3, 3, P)
    elif mode == 2:    # duplicated / nearly duplicated obs rows: strongly correlated obs block
        k = max(1, P // 3)
        src = rng.integers(0, P, k)
        dst = rng.integers(0, P, k)
        c["HX"][dst] = c["HX"][src] + 1e-3 * rng.standard_normal((k, M))
        c["val"][dst] = c["HX"][dst].mean(axis=1) + rng.standard_normal(k)
        if loc:
            c["ob_lat"][dst], c["ob_lon"][dst] = c["ob_lat"][src], c["ob_lon"][src]
    elif mode == 3 and loc:  # short and long radii mixed (zero tapers, whole-globe tapers)
        c["hw"] = 10.0 ** rng.uniform(1.5, 4.3, P)
    xam, Xap, diag = T._run_oracle(c)
    pipeline = int(rng.choice([0, 1, 2, 3, 3, 3]))
    path = str(rng.choice(["auto", "sweep"] if loc else ["auto", "sweep", "transform"]))
    batch = int(rng.choice([1, 7, 32, 64]))
    tag = "case %3d loc=%d M=%3d P=%4d N=%4d assim=%.2f mode=%d pipeline=%d path=%s batch=%d" % (it, loc, M, P, N, frac, mode, pipeline, path, batch)
    try:
        h_xam, h_Xap, h_diag = T._run_hip(c, path=path, batch=batch, pipeline=pipeline)
        kind = ctx.get_option("phase_a_kind")
        e = 0.0
        for got, ref in ((h_xam, xam), (h_Xap, Xap)) + tuple((h_diag[k], diag[k]) for k in ("prior_mean", "prior_var", "post_mean", "post_var")):
            ref = np.asarray(ref, dtype=float)
            got = np.asarray(got, dtype=float)
            ok = np.isfinite(ref)
            scale = max(float(np.max(np.abs(ref[ok]))) if ok.any() else 0.0, 1e-300)
            e = max(e, float(np.max(np.abs(got[ok] - ref[ok]))) / scale if ok.any() else 0.0)
            assert np.array_equal(np.isfinite(got), ok)
        assert np.array_equal(h_diag["assimilated"], diag["assimilated"])
        worst = max(worst, e)
        status = "ok" if e < 1e-10 else "FAIL"
        if e >= 1e-10:
            fails.append(tag + " err %.2e" % e)
        print("%s kind=%d rel err %.2e %s" % (tag, kind, e, status), flush=True)
    except Exception as ex:  # noqa: BLE001
        fails.append(tag + " " + repr(ex)[:300])
        print(tag, "EXCEPTION", repr(ex)[:300], flush=True)
print("fuzz: %d cases, %d failures, worst rel err %.2e, %.0f s" % (ncases, len(fails), worst, time.time() - t0))
for f in fails:
    print("  ", f)
sys.exit(1 if fails else 0)
