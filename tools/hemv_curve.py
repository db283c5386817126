#!/usr/bin/env python3
"""Time per launch of the panel mat-vec kernel (plain hemv mode) as a function of the trailing order n:
separates the fixed (latency) part from the streaming part.  Usage: python tools/hemv_curve.py [N] [real]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cplx = not (len(sys.argv) > 2 and sys.argv[2] == "real")
dt = torch.complex128 if cplx else torch.float64
s = 16 if cplx else 8
A = torch.randn((N, N), dtype=dt, device="cuda")
x = torch.randn(N, dtype=dt, device="cuda")
print("# n  us/launch  TB/s(algorithmic s*n(n+1)/2)  tiles")
ns = (64, 128, 256, 384, 512, 768, 1024, 1280, 1536, 1792, 2048, 2304, 2560, 3072, 3584, 4096)
if os.environ.get("HEMV_CURVE_FINE"):
    ns = tuple(range(1984, N + 1, 128))
for n in ns:
    if n > N:
        break
    ms = api.hemv_bench(A, x, reps=200, n=n)
    nt = (n + 63) // 64
    print("%5d  %8.2f  %6.2f  %5d" % (n, ms * 1e3, s * n * (n + 1) / 2 / (ms * 1e-3) * 1e-12, nt * (nt + 1) // 2), flush=True)
