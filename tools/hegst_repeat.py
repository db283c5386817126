#!/usr/bin/env python3
"""Repeated hegst calls, wall-clock each (looking for sporadic host-side stalls)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import eigensolver_gpu_amd.api as api
N = 4096
rng = np.random.default_rng(0)
def spd(n):
    T = rng.random((n, n)) + 1j * rng.random((n, n))
    return T @ T.conj().T + n * np.eye(n)
A = spd(N); B = spd(N)
Ad = torch.from_numpy(np.ascontiguousarray(A.T)).cuda(); Bd = torch.from_numpy(np.ascontiguousarray(B.T)).cuda()
api.potrf(Bd)
bufs = [Ad.clone() for _ in range(8)]
torch.cuda.synchronize()
for i in range(8):
    t0 = time.perf_counter(); api.hegst(bufs[i], Bd); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("call %d: %.2f ms" % (i, (t1 - t0) * 1e3))
