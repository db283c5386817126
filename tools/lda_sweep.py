#!/usr/bin/env python3
"""The mat-vec launches of one whole tridiagonalisation, back to back, for different leading dimensions of the same order.
Usage: python tools/lda_sweep.py n:pad,pad,... [n:pad,...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

real = os.environ.get("LDA_SWEEP_REAL")
dt = torch.float64 if real else torch.complex128
for arg in sys.argv[1:] or ["8192:0,64,192"]:
    n, pads = arg.split(":")
    n = int(n)
    for pad in [int(p) for p in pads.split(",")]:
        lda = n + pad
        A = torch.randn((n, lda), dtype=dt, device="cuda")
        r = api.hetrd_mv_sweep(A, 0, 1)
        r = api.hetrd_mv_sweep(A, 0, 2)
        print("n %5d lda %5d (+%d)  sweep %8.2f ms  %5.2f TB/s  (%d launches)" % (n, lda, pad, r["ms_total"], r["algo_bytes"] / r["ms_total"] * 1e-9, r["launches"]), flush=True)
        del A
