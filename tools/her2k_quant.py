#!/usr/bin/env python3
"""Upper-triangle rank-2k update with long K: tile-quantisation experiment (auto split-K on/off)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eigensolver_gpu_amd import api
dt = torch.complex128
for n, k in ((2048, 2048), (1984, 2048), (2112, 2048), (1024, 1024), (3000, 1024)):
    V = torch.randn((k, n), dtype=dt, device='cuda'); W = torch.randn((k, n), dtype=dt, device='cuda'); C = torch.randn((n, n), dtype=dt, device='cuda')
    ms = api.her2k_bench(V, W, C, n, k, reps=5)
    nt = (n + 63) // 64
    print("autosplit=%s her2k n=%d k=%d (%d upper tiles): %.1f us %.1f TF" % (os.environ.get("EIGSOLVE_GEMM_AUTOSPLIT", "on"), n, k, nt * (nt + 1) // 2, ms * 1e3, 4 * 2.0 * n * n * k / ms * 1e-9))
