"""Time configs[4]: 1e6 state x 128 members x 4096 obs fp32 contraction (device-resident)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from efa_xray_amd import _lib
N, M, P = 1_000_000, 128, 4096
ctx = _lib.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
X = torch.randn((N, M), dtype=torch.float32, device="cuda")
Ye = torch.randn((P, M), dtype=torch.float32, device="cuda")
C = torch.empty((N, P), dtype=torch.float32, device="cuda")
for _ in range(2):
    ctx.cov_contract_f32(N, M, P, X.data_ptr(), Ye.data_ptr(), C.data_ptr())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
reps = 5
for _ in range(reps):
    ctx.cov_contract_f32(N, M, P, X.data_ptr(), Ye.data_ptr(), C.data_ptr())
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
flops = 2.0 * N * M * P
byts = 4.0 * (N * M + P * M + N * P)
print("contract f32 %d x %d x %d: %.3f ms  %.1f TFLOP/s (%.0f%% of 157.3)  %.0f GB/s algorithmic bytes" % (N, M, P, ms, flops / ms / 1e9, 100 * flops / ms / 1e9 / 157.3, byts / ms / 1e6))
ref = (X[:64].double() @ Ye.double().T)
err = ((C[:64].double() - ref).abs() / (ref.abs() + 1e-3)).max().item()
print("max rel err vs float64 on 64 rows: %.2e" % err)
t0 = time.perf_counter(); torch.matmul(X, Ye.T, out=C); torch.cuda.synchronize(); t1 = time.perf_counter()
e0.record()
for _ in range(reps): torch.matmul(X, Ye.T, out=C)
e1.record(); torch.cuda.synchronize()
print("for scale only: torch.matmul (hipBLASLt) same shape: %.3f ms" % (e0.elapsed_time(e1) / reps))
