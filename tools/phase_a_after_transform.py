"""Diagnostic: why Phase A takes 4.4 ms inside a headline cycle and 4.0 ms on its own.  The same k_pipe_band launch (1e4 obs x 100
members) timed by the library's events with different work put on the stream between consecutive Phase A launches."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from efa_xray_amd import _lib

ctx = _lib.get_context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.set_option("timing", 1)
rng = np.random.default_rng(0)
M, P, N = 100, 10000, 10_000_000
HX = 3.0 * rng.standard_normal((P, M))
val = HX.mean(axis=1) + rng.standard_normal(P)
err, asm = np.ones(P), np.ones(P, bool)
Xt = torch.empty((N, M), dtype=torch.float64, device="cuda")
Pt = torch.empty_like(Xt)
ctx.fill_synthetic(N, 0, M, 7, 3.0, Xt.data_ptr())
A = torch.randn(8192, 8192, device="cuda", dtype=torch.float64)
B = torch.randn(8192, 8192, device="cuda", dtype=torch.float64)


def phase_a():
    Yp = ctx.to_device(HX)
    ym = ctx.empty((P,))
    ctx.form_perts(P, M, Yp, ym, Yp)
    ctx.obs_phase(M, P, ym, Yp, val, err, asm)
    return ctx.last_timing()["obs_ms"]


def run(label, fn, n=8):
    out = []
    for _ in range(n):
        o = phase_a()
        fn()
        out.append(o)
    print("%-58s Phase A ms %s" % (label, np.round(out[1:], 3)), flush=True)


T = lambda: ctx.state_cycle(N, M, Xt.data_ptr(), Pt.data_ptr())
run("nothing in between", lambda: None)
run("the transform (16 GB of HBM traffic + 200 GFLOP fp64 MFMA)", T)
run("an 8 GB device-to-device copy", lambda: Pt.copy_(Xt))
run("an fp64 GEMM 8192^3 (hipBLASLt, ~20 ms, no HBM stream)", lambda: torch.mm(A, B))
run("the transform, then read 2 GB of the prior", lambda: (T(), Xt[:2560000].sum()))
run("the transform, then 1 ms of idle", lambda: (T(), torch.cuda.synchronize(), time.sleep(0.001)))
run("the transform, then 20 ms of idle", lambda: (T(), torch.cuda.synchronize(), time.sleep(0.02)))
run("nothing in between", lambda: None)
ctx.set_option("timing", 0)
