#!/usr/bin/env python3
"""Counter targets for the workgroup -> tile map of the MFMA engine (option tile_map / EIGSOLVE_TILE_MAP): products whose
HBM-side traffic (FETCH_SIZE) is compared with the map on and off.  Run under rocprofv3 --kernel-trace --pmc FETCH_SIZE."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
n = 4096
dt = torch.complex128
A = torch.randn((n, n), dtype=dt, device=dev)
B = torch.randn((n, n), dtype=dt, device=dev)
C = torch.empty((n, n), dtype=dt, device=dev)
api.gemm_bench("N", "N", n, n, n, A, n, B, n, C, n, reps=1)            # zgemm 4096^3: 805 MB algorithmic
api.gemm_bench("N", "N", 2048, 2048, 2048, A, n, B, n, C, n, reps=1)   # hegst's hemm at C3
api.gemm_bench("N", "C", 4096, 1024, 256, A, n, B, n, C, n, reps=1)    # back-transformation update
V = torch.randn((64, n), dtype=dt, device=dev)
W = torch.randn((64, n), dtype=dt, device=dev)
api.her2k_bench(V, W, A.clone(), n, 64, reps=1)                        # trd trailing update (folded triangle)
V2 = torch.randn((1024, n), dtype=dt, device=dev)
W2 = torch.randn((1024, n), dtype=dt, device=dev)
api.her2k_bench(V2, W2, A.clone(), 2048, 1024, reps=1)                  # hegst's her2k (split-K)
torch.cuda.synchronize()
print("done")
