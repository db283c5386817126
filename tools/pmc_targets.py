#!/usr/bin/env python3
"""The kernels whose hardware counters profiles/ reports (run under `rocprofv3 --kernel-trace --pmc ...`, one counter
group per pass, see tools/pmc_collect.sh): plain hemv launches at n = 4096, zgemm 4096^3, the tridiagonalization's
rank-2k update (n = 4096, k = 64), the Cholesky factorization and one reduction to standard form."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import gen_pair  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
n = 4096
dt = torch.complex128
A0, B0 = gen_pair(n, True, 1002, dev)
x = torch.randn(n, dtype=dt, device=dev)
api.hemv_bench(A0, x, reps=3, n=4096)     # (one order only: every n >= 2048 launches the same 512-workgroup grid)
Bm = torch.randn((n, n), dtype=dt, device=dev)
Cm = torch.empty((n, n), dtype=dt, device=dev)
api.gemm_bench("N", "N", n, n, n, A0, n, Bm, n, Cm, n, reps=1)
V = torch.randn((64, n), dtype=dt, device=dev)
W = torch.randn((64, n), dtype=dt, device=dev)
C = A0.clone()
api.her2k_bench(V, W, C, n, 64, reps=2)
B = B0.clone()
assert api.potrf(B) == 0
A = A0.clone()
api.hegst(A, B)
torch.cuda.synchronize()
print("pmc targets done")
