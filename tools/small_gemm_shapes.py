#!/usr/bin/env python3
"""Times the MFMA engine on the <= 256-workgroup (32 x 32 tile) products of hegst / trsm / the T factors (two passes).  A/B a
compile-time variant with EIGSOLVE_GPU_LIB=<variant .so> (make OUTDIR=../lib/v_x EXTRA=-D...), both commands in one gpurun call.
Usage: python tools/small_gemm_shapes.py [real]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
shapes = [("N", "N", 256, 1024, 256), ("C", "N", 256, 1024, 256), ("N", "C", 256, 1024, 256), ("N", "N", 256, 1024, 512),
          ("N", "N", 256, 1024, 1024), ("N", "N", 256, 512, 256), ("N", "N", 128, 1024, 256), ("C", "N", 256, 256, 4096),
          ("N", "N", 512, 512, 512), ("N", "N", 256, 1024, 128), ("N", "N", 256, 2048, 256), ("N", "N", 64, 2048, 2048)]
for ta, tb, M, N, K in shapes:
    out = []
    for rep in range(2):
        ms = api.gemm_bench(ta, tb, M, N, K, A, big, B, big, C, big, reps=20)
        out.append("%7.1f us %5.1f TFLOP/s" % (ms * 1e3, cm * M * N * K / (ms * 1e-3) * 1e-12))
    print("%s%s M=%5d N=%5d K=%5d   %s | %s" % (ta, tb, M, N, K, out[0], out[1]), flush=True)
