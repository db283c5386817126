#!/usr/bin/env python3
"""The kernels whose hardware counters profiles/ reports (run under `rocprofv3 --kernel-trace --pmc ...`, one counter
group per pass, see tools/pmc_collect.sh).  Complex (C3 shapes): plain hemv launches at n = 4096, zgemm 4096^3, the
tridiagonalization's rank-2k update (n = 4096, k = 64), the Cholesky factorization, one reduction to standard form, the
back-transformation (N = 4096, m = 1024, 256-reflector blocks) and the final trsm.  Real (C2, dsygvdx N = 2048 m = 512):
dsymv, dsyr2k k = 64, dpotrf, dsygst, the back-transformation and the trsm."""
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
# back-transformation and trsm.  (Counter collection serialises every dispatch: the ~8000 launches of a tridiagonalization would
# take minutes per pass, so the reflectors are simply the upper triangle of A as it is, with small taus -- the launches, shapes
# and masks of the back-transformation are the same.)
tau = (0.01 * torch.randn(n - 1, dtype=torch.float64, device=dev)).to(dt)
Z = torch.zeros((n, n), dtype=dt, device=dev)
Z[:1024, :] = torch.randn((1024, n), dtype=dt, device=dev)
api.unmtr(A, tau, Z, 1024, 256)
api.trsm_lun(B, Z, 1024)
# real path: C2
nr = 2048
Ar, Br = gen_pair(nr, False, 1001, dev)
xr = torch.randn(nr, dtype=torch.float64, device=dev)
api.hemv_bench(Ar, xr, reps=3, n=nr)
Vr = torch.randn((64, nr), dtype=torch.float64, device=dev)
Wr = torch.randn((64, nr), dtype=torch.float64, device=dev)
api.her2k_bench(Vr, Wr, Ar.clone(), nr, 64, reps=2)
Bq = Br.clone()
assert api.potrf(Bq) == 0
Aq = Ar.clone()
api.hegst(Aq, Bq)
taur = 0.01 * torch.randn(nr - 1, dtype=torch.float64, device=dev)
Zr = torch.zeros((nr, nr), dtype=torch.float64, device=dev)
Zr[:512, :] = torch.randn((512, nr), dtype=torch.float64, device=dev)
api.unmtr(Aq, taur, Zr, 512, 256)
api.trsm_lun(Bq, Zr, 512)
torch.cuda.synchronize()
print("pmc targets done")
