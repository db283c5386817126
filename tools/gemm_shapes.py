#!/usr/bin/env python3
"""Times the fp64 MFMA engine on the shapes the C3 solve contains (beta = 0 products through eigsolve_?gemm_bench).
Usage: python tools/gemm_shapes.py [real]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _toolslib  # noqa: E402,F401  (tools-side build of the library: EIG_TOOLS hooks)
import torch  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

cplx = not (len(sys.argv) > 1 and sys.argv[1] == "real")
dt = torch.complex128 if cplx else torch.float64
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
big = 4096
A = torch.randn((big, big), dtype=dt, device=dev)
B = torch.randn((big, big), dtype=dt, device=dev)
C = torch.empty((big, big), dtype=dt, device=dev)
cm = 8.0 if cplx else 2.0
shapes = [("N", "N", 4096, 4096, 4096), ("N", "N", 2048, 2048, 2048), ("N", "N", 1024, 2048, 1024),
          ("N", "C", 4096, 1024, 64), ("N", "C", 4096, 1024, 128), ("N", "C", 4096, 1024, 256), ("N", "C", 4096, 1024, 512),
          ("N", "C", 2048, 1024, 256), ("N", "C", 1024, 1024, 256),
          ("N", "C", 4096, 4096, 64), ("N", "C", 4096, 4096, 128), ("N", "C", 4096, 4096, 256),
          ("C", "N", 1024, 256, 4096), ("C", "N", 1024, 256, 2048), ("C", "N", 1024, 512, 4096),
          ("C", "N", 2048, 2048, 64), ("C", "N", 4032, 4032, 64), ("C", "N", 4032, 4032, 128),
          ("N", "N", 256, 2048, 256), ("N", "N", 512, 2048, 512), ("N", "N", 1024, 1024, 256), ("N", "N", 256, 1024, 256)]
for ta, tb, M, N, K in shapes:
    ms = api.gemm_bench(ta, tb, M, N, K, A, big, B, big, C, big, reps=10)
    print("%s%s M=%5d N=%5d K=%5d  %9.1f us  %6.1f TFLOP/s" % (ta, tb, M, N, K, ms * 1e3, cm * M * N * K / (ms * 1e-3) * 1e-12), flush=True)

# beta = 1 (C read-modify-write) and masked operands (the back-transformation's V is a unit trapezoid, mask 4)
import ctypes  # noqa: E402
L = api.lib()
fn = L.eigsolve_zgemm_probe if cplx else L.eigsolve_dgemm_probe
for ta, tb, M, N, K, b1, mA, oA, mB, oB in [("N", "C", 4096, 1024, 256, 0, 0, 0, 0, 0), ("N", "C", 4096, 1024, 256, 1, 0, 0, 0, 0),
                                            ("N", "C", 4096, 1024, 256, 0, 4, 3840, 0, 0), ("N", "C", 4096, 1024, 256, 1, 4, 3840, 0, 0),
                                            ("N", "C", 4096, 1024, 128, 1, 4, 3968, 0, 0), ("N", "C", 4096, 1024, 512, 1, 4, 3584, 0, 0),
                                            ("C", "N", 1024, 256, 4096, 0, 0, 0, 4, 3840), ("C", "N", 1024, 256, 4096, 0, 0, 0, 0, 0),
                                            ("N", "N", 2048, 2048, 2048, 1, 0, 0, 0, 0), ("N", "N", 2048, 2048, 2048, 0, 0, 0, 0, 0)]:
    ms = ctypes.c_double(0)
    rc = fn(ctypes.c_char(ta.encode()), ctypes.c_char(tb.encode()), M, N, K, api._p(A), big, api._p(B), big, api._p(C), big, 10, b1, mA, oA,
            mB, oB, ctypes.byref(ms))
    assert rc == 0
    print("%s%s M=%5d N=%5d K=%5d beta=%d maskA=%d maskB=%d  %9.1f us  %6.1f TFLOP/s" %
          (ta, tb, M, N, K, b1, mA, mB, ms.value * 1e3, cm * M * N * K / (ms.value * 1e-3) * 1e-12), flush=True)
