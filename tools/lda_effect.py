#!/usr/bin/env python3
"""Does the mat-vec's streaming rate depend on the LEADING DIMENSION (power-of-two column strides: channel / TLB effects)?
Same order n, different lda.  Usage: python tools/lda_effect.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

dt = torch.complex128
for n, ldas in ((4096, (4096, 4160, 4224, 8192, 8256, 16384, 16448)), (8192, (8192, 8256, 8320, 16384, 16448)),
                (12288, (12288, 12352, 16384, 16448))):
    x = torch.randn(n, dtype=dt, device="cuda")
    for lda in ldas:
        A = torch.randn((n, lda), dtype=dt, device="cuda")
        ms = api.hemv_bench(A, x, reps=100, n=n)
        print("n %5d lda %5d  %8.2f us  %5.2f TB/s" % (n, lda, ms * 1e3, 16 * n * (n + 1) / 2 / (ms * 1e-3) * 1e-12), flush=True)
        del A
