#!/usr/bin/env python3
"""One solve at an order beyond BASELINE's configs (sizing check for 288 GB of HBM: every index is in elements, byte offsets
exceed 2^32 from N = 16384 on).  Well-conditioned pair (A random Hermitian, B = T T^H / N + I), m lowest eigenpairs, residual and
B-orthonormality evaluated on the device.  Usage: python tools/large_order.py [n] [m] [real]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
m = int(sys.argv[2]) if len(sys.argv) > 2 else n // 4
cplx = not (len(sys.argv) > 3 and sys.argv[3] == "real")
dt = torch.complex128 if cplx else torch.float64
torch.cuda.set_device(0)
g = torch.Generator(device="cuda").manual_seed(4242)
R = torch.randn((n, n), dtype=dt, device="cuda", generator=g)
A0 = (R + R.conj().T) * 0.5
T = torch.randn((n, n), dtype=dt, device="cuda", generator=g)
B0 = T @ T.conj().T / n
del R, T
B0 += torch.eye(n, dtype=dt, device="cuda")
torch.cuda.synchronize()
ws = api.Workspace(n, cplx, pinned=False)
out = {"n": n, "m": m, "dtype": "c128" if cplx else "f64"}
for rep in range(2):
    A, B = A0.T.contiguous(), B0.T.contiguous()   # column-major device images (a torch tensor is row-major)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info, _ = api.hegvdx(A, B, 1, m, ws)
    wall = (time.perf_counter() - t0) * 1e3
    out["info"] = info
    out["wall_ms_run%d" % rep] = wall
    out["phase_ms"] = api.phase_times()
    del A, B
# column-major (N,N) device tensor: Z[j] is eigenvector j
Z = ws.Z[:m].T.contiguous() if True else None      # n x m, math orientation
lam = ws.w[:m]
AZ = A0 @ Z
BZ = B0 @ Z
res = AZ - BZ * lam.to(dt)[None, :]
out["residual_rel_fro"] = float(torch.linalg.norm(res) / (torch.linalg.norm(A0) * torch.linalg.norm(Z)))
G = Z.conj().T @ BZ
G -= torch.eye(m, dtype=dt, device="cuda")
out["b_orthonormality_max"] = float(G.abs().max())
out["eps_n"] = n * 2.220446049250313e-16
out["host_copy_matches_device"] = bool(torch.equal(ws.Z_h[:m].to("cuda"), ws.Z[:m]))
out["eigenvalues_sorted"] = bool((lam[1:] >= lam[:-1]).all())
out["hbm_allocated_GB"] = torch.cuda.max_memory_allocated() / 1e9
print(json.dumps(out))
