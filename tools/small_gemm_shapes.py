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
          ("N", "N", 512, 512, 512), ("N", "N", 256, 1024, 128), ("N", "N", 256, 2048, 256), ("N", "N", 64, 2048, 2048),
          ("N", "N", 512, 1024, 512), ("N", "C", 512, 512, 64), ("C", "N", 1024, 256, 4096), ("C", "N", 1024, 256, 1024), ("N", "C", 1024, 256, 256)]
modes = [0, 1, 2] if cplx else [0]        # option "gemm_wide": four-wave workgroups | whole-CU workgroups (16 waves) | + 8-wave form
print("option gemm_wide = " + " | ".join(str(m) for m in modes))
for ta, tb, M, N, K in shapes:
    out = []
    for md in modes:
        api.set_option("gemm_wide", md)
        ms = min(api.gemm_bench(ta, tb, M, N, K, A, big, B, big, C, big, reps=20) for rep in range(2))
        out.append("%7.1f us %5.1f TFLOP/s" % (ms * 1e3, cm * M * N * K / (ms * 1e-3) * 1e-12))
    print("%s%s M=%5d N=%5d K=%5d   %s" % (ta, tb, M, N, K, " | ".join(out)), flush=True)
api.set_option("gemm_wide", -1)
