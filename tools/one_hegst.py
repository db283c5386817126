#!/usr/bin/env python3
"""potrf + hegst once at N (complex), for rocprofv3 kernel traces of the reduction to standard form."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import eigensolver_gpu_amd.api as api
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
def spd(n):
    T = rng.random((n, n)) + 1j * rng.random((n, n))
    return T @ T.conj().T + n * np.eye(n)
A = spd(N); B = spd(N)
Ad = torch.from_numpy(np.ascontiguousarray(A.T)).cuda(); Bd = torch.from_numpy(np.ascontiguousarray(B.T)).cuda()
api.potrf(Bd)
for _ in range(2):
    A2 = Ad.clone()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); api.hegst(A2, Bd); e1.record(); torch.cuda.synchronize()
    print("hegst N=%d: %.2f ms" % (N, e0.elapsed_time(e1)))
