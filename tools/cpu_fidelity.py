#!/usr/bin/env python3
"""BASELINE.md section 3 fidelity check (build container only: needs /root/reference):
wall time per observation of the REFERENCE's EnSRF.update loop against the oracle in faithful_cost mode (what
bench.py times as cpu_baseline) at configs[1] (512x512 x 50 members).  Must agree within +-15 %.

    python3 -B tools/cpu_fidelity.py [nobs]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_goldens as mg            # installs the placeholder modules and imports the reference
from oracle import ensrf_oracle as orc

nobs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nwarm = 2
rng = np.random.default_rng(0)
ny = nx = 512
M = 50
arr = rng.standard_normal((1, 1, ny, nx, 1)) + 3.0 * rng.standard_normal((1, 1, ny, nx, M))
lat, lon = np.meshgrid(np.linspace(20, 60, ny), np.linspace(200, 280, nx), indexing="ij")
N = ny * nx
X = arr.reshape(N, M)
rows = rng.choice(N, nobs, replace=False)


def reference_seconds(n):
    state = mg.DuckState(arr, lat, lon)
    obs = [mg.LinOb([rows[k]], [1.0], value=float(X[rows[k]].mean() + 0.5), error=1.0, lat=0.0, lon=0.0,
                    assimilate_this=True) for k in range(n)]
    flt = mg.EnSRF(state, obs, verbose=False, loc=False)
    xbm, Xbp = flt.format_prior_state()
    flt.format_prior_state = lambda: (xbm, Xbp)          # time the loop, not the setup
    flt.format_posterior_state = lambda xam, Xap: (None, obs)
    t0 = time.perf_counter()
    flt.update()
    return time.perf_counter() - t0


def oracle_seconds(n):
    xbm, Xbp = orc.format_prior_state(X, X[rows[:n]])
    val = X[rows[:n]].mean(axis=1) + 0.5
    t0 = time.perf_counter()
    orc.ensrf_update(xbm, Xbp, N, val, np.ones(n), np.ones(n, dtype=bool), faithful_cost=True)
    return time.perf_counter() - t0


for name, fn in (("reference EnSRF.update (ensrf.py:50-149)", reference_seconds), ("oracle, faithful_cost", oracle_seconds)):
    fn(nwarm)
    tw = fn(nwarm)
    ta = fn(nwarm + nobs - nwarm)
    per = (ta - tw) / (nobs - 2 * nwarm) if nobs > 2 * nwarm else float("nan")
    print("%-44s %.1f ms per observation (%d x %d x %d, %d obs after %d warm-up, %d threads)"
          % (name, 1e3 * per, ny, nx, M, nobs - 2 * nwarm, nwarm, os.cpu_count()))
