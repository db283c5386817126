#!/usr/bin/env python3
"""Cholesky factorization alone: ms per factorization and backward error for the "potrf" option settings
(2 = block rows in pairs, elimination blocked by 16 on MFMA; 1 = the round-2 block-row kernel; 0 = recursive).
Usage: python tools/potrf_bench.py [n] [real] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import gen_pair  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cplx = not (len(sys.argv) > 2 and sys.argv[2] == "real")
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
modes = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [2, 1, 0]
lax = os.environ.get("POTRF_BENCH_LAX") == "1"      # timing of deliberately broken build variants
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
_, B0 = gen_pair(n, cplx, 1002, dev)
ref = None
for mode in modes:
    api.set_option("potrf", mode)
    ts = []
    for r in range(1 + reps):
        B = B0.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info = api.potrf(B)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        assert lax or info == 0, info
    ts = sorted(ts[1:])
    # B is stored column-major in a (n, n) tensor whose [j, i] entry is element (i, j): U = triu in matrix terms
    M = B.T                           # matrix view: M[i, j] = element (i, j)
    Uf = torch.triu(M)
    Bm = torch.triu(B0.T)
    Bfull = Bm + Bm.conj().T - torch.diag(torch.diagonal(Bm))
    err = (torch.linalg.norm(Uf.conj().T @ Uf - Bfull) / torch.linalg.norm(Bfull)).item()
    if ref is None:
        ref = Uf
    dif = (torch.linalg.norm(Uf - ref) / torch.linalg.norm(ref)).item()
    print("potrf=%d  n=%d %s  min %.3f  median %.3f ms   ||U^H U - B||/||B|| = %.2e   vs mode 2: %.2e" %
          (mode, n, "z" if cplx else "d", ts[0], ts[len(ts) // 2], err, dif), flush=True)
api.set_option("potrf", -1)
