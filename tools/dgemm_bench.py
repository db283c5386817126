#!/usr/bin/env python3
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eigensolver_gpu_amd import api
for dt, nm in ((torch.float64, "dgemm"), (torch.complex128, "zgemm")):
    for n in (2048, 4096):
        A = torch.randn((n, n), dtype=dt, device='cuda'); B = torch.randn((n, n), dtype=dt, device='cuda'); C = torch.zeros((n, n), dtype=dt, device='cuda')
        ms = api.gemm_bench('N', 'N', n, n, n, A, n, B, n, C, n, reps=5)
        fl = (8.0 if dt == torch.complex128 else 2.0) * n ** 3
        print("%s %d^3 (NO128=%s): %.3f ms %.1f TFLOP/s" % (nm, n, os.environ.get("EIGSOLVE_GEMM_NO128", "0"), ms, fl / ms * 1e-9))
