#!/usr/bin/env python3
"""Single-launch rate of the mat-vec kernel (plain hemv mode) at a few orders + the sweep of one tridiagonalization.
EIGSOLVE_GPU_LIB selects a compile-time variant build for A/B runs (make OUTDIR=../lib/v_x EXTRA=-D...); EIGSOLVE_MV_DMA=0/1 the
data path of the kernel (registers / LDS-DMA ring).
Usage: python tools/hemv_rate.py [N ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

torch.cuda.set_device(0)
tag = os.path.basename(os.path.dirname(os.environ.get("EIGSOLVE_GPU_LIB", "product/x")))
for N in [int(a) for a in sys.argv[1:]] or [8192, 4096]:
    A = torch.randn((N, N), dtype=torch.complex128, device="cuda")
    x = torch.randn(N, dtype=torch.complex128, device="cuda")
    out = []
    for n in (N, N * 3 // 4, N // 2):
        ms = api.hemv_bench(A, x, reps=50, n=n)
        out.append("n=%d %.1f us %.2f TB/s" % (n, ms * 1e3, 16 * n * (n + 1) / 2 / (ms * 1e-3) * 1e-12))
    r = api.hetrd_mv_sweep(A.clone(), 0, reps=1)
    print("%-10s N=%d: %s | sweep %.1f ms %.2f TB/s" % (tag, N, "  ".join(out), r["ms_total"], r["algo_bytes"] / (r["ms_total"] * 1e-3) * 1e-12), flush=True)
