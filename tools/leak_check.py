#!/usr/bin/env python3
"""Repeated solves of varying order: device memory held by the library's grow-only scratch must plateau."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import eigensolver_gpu_amd.api as api
rng = np.random.default_rng(0)
def spd(n, cplx, shift):
    T = rng.random((n, n)) + (1j * rng.random((n, n)) if cplx else 0.0)
    return T @ T.conj().T + shift * np.eye(n)
free0 = None
for it in range(120):
    n = int(rng.choice([64, 200, 333, 512, 700]))
    cplx = bool(it & 1)
    A = spd(n, cplx, 0.0); B = spd(n, cplx, float(n))
    info, ws = api.hegvdx(api.to_device(np.triu(A)), api.to_device(np.triu(B)), 1, max(1, n // 4))
    assert info == 0
    del ws
    torch.cuda.synchronize()
    free, tot = torch.cuda.mem_get_info()
    if it == 40: free0 = free
    if it % 20 == 0: print("iter %3d n=%4d free %.1f MB" % (it, n, free / 2**20))
print("free after warm-up %.1f MB, at the end %.1f MB" % (free0 / 2**20, free / 2**20))
assert free0 - free < 64 * 2**20, "device memory keeps growing"
print("OK")
