import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import test_gpu_parity as T
from oracle import ensrf_oracle as orc
rng = np.random.default_rng(2026)
worst = 0.0
for it in range(24):
    M = int(rng.choice([2, 4, 6, 10, 18, 34, 50, 66, 80, 98, 100, 102, 104]))
    n_lead = int(rng.integers(1, 41)); ncol = int(rng.integers(17, 140)); P = int(rng.integers(5, 70))
    N = n_lead * ncol
    c = T._random_case(5000 + it, N, M, P, True, ncol=ncol)
    c["hw"][:] = rng.uniform(300, 4000, P)
    xam, Xap, diag = T._run_oracle(c)
    ctx = T._ctx()
    X = ctx.to_device(c["X"]); Yp = ctx.to_device(c["HX"]); ym = ctx.empty((P,))
    ctx.form_perts(P, M, Yp, ym, Yp)
    ctx.obs_phase(M, P, ym, Yp, c["val"], c["err"], c["asm"], 1, c["ob_lat"], c["ob_lon"], c["hw"])
    post = ctx.empty((N, M))
    ctx.state_cycle(N, M, X, post, c["lat"].reshape(-1), c["lon"].reshape(-1), c["n_lead"])
    ref = orc.format_posterior_state(xam, Xap, N)
    got = post.download()
    err = np.max(np.abs(got - ref)) / max(np.max(np.abs(ref)), 1e-300)
    worst = max(worst, err)
    h_xam, h_Xap, _ = T._run_hip(c, path="sweep")     # perturbation form (FUSED = false)
    err2 = np.max(np.abs(h_Xap - Xap)) / max(np.max(np.abs(Xap)), 1e-300)
    worst = max(worst, err2)
    print(it, "M", M, "n_lead", n_lead, "ncol", ncol, "P", P, "rel err %.2e %.2e" % (err, err2), flush=True)
    assert err < 1e-10 and err2 < 1e-10
print("fuzz ok, worst", worst)
