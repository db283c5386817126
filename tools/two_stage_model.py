#!/usr/bin/env python3
"""Go / no-go measurement for a two-stage tridiagonalization: the complete LAUNCH SKELETON of stage 1
(full -> band of width 64: CholeskyQR2 + Householder reconstruction per panel, two-sided rank-2b trailing update; true shapes,
masks and K on the MFMA engine, data meaningless) timed on the GPU -- a real implementation cannot be faster than its own launch
sequence.  Kill criterion of the review: stage 1 > 15 ms at N = 4096 or > 100 ms at N = 8192 (complex).
Usage: python tools/two_stage_model.py [N ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _toolslib  # noqa: E402,F401  (tools-side build of the library: EIG_TOOLS hooks)
import torch  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

torch.cuda.set_device(0)
h = api.lib()
f = h.eigsolve_debug_two_stage_model
f.restype = ctypes.c_int
for n in [int(a) for a in sys.argv[1:]] or [4096, 8192]:
    for cplx in (1, 0):
        row = []
        for what in (0, 1, 2):
            ms = ctypes.c_double(0)
            rc = f(ctypes.c_int(n), ctypes.c_int(cplx), ctypes.c_int(what), ctypes.c_int(3), ctypes.byref(ms))
            assert rc == 0
            row.append(ms.value)
        fl = (8.0 if cplx else 2.0) * (2.0 / 3.0) * n ** 3
        print("stage-1 skeleton  N=%5d %s   all %8.2f ms   panel chains alone %8.2f ms (%d panels, %.0f us each)   trailing products "
              "alone %8.2f ms (%.1f TFLOP/s on 2/3 N^3 multiply-adds)" % (n, "complex" if cplx else "real   ", row[0], row[1], n // 64 - 1,
                                                                           row[1] * 1e3 / (n // 64 - 1), row[2], fl / (row[2] * 1e-3) * 1e-12),
              flush=True)
