#!/usr/bin/env python3
"""Device divide & conquer vs LAPACK on assorted tridiagonals (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.linalg import eigh_tridiagonal
from eigensolver_gpu_amd import api


def check(d, e, name):
    n = len(d)
    rc, w, Q, ms = api.stedc_device(d, e)
    t0 = time.time()
    wr = eigh_tridiagonal(d, e, eigvals_only=True) if n > 1 else d.copy()
    T = np.diag(d) + (np.diag(e, 1) + np.diag(e, -1) if n > 1 else 0)
    nrm = max(np.abs(wr).max(), 1e-300)
    print("%-22s n=%5d rc=%d  |w-wref|/|w|=%.1e  orth=%.1e  resid=%.1e  %.1f ms" % (
        name, n, rc, np.abs(w - wr).max() / nrm, np.abs(Q.T @ Q - np.eye(n)).max(), np.abs(T @ Q - Q * w).max() / nrm, ms), flush=True)


rng = np.random.default_rng(0)
for n in (1, 2, 5, 31, 32, 33, 64, 65, 100, 257, 1000):
    check(rng.standard_normal(n), rng.standard_normal(max(n - 1, 0)), "random")
n = 600
check(np.ones(n) * 2, -np.ones(n - 1), "1-2-1 laplacian")
check(np.abs(np.arange(n) - n // 2).astype(float), np.ones(n - 1), "wilkinson")
dg = np.concatenate([np.abs(np.arange(21) - 10).astype(float)] * 20)
eg = np.ones(len(dg) - 1); eg[20::21] = 1e-8
check(dg, eg, "glued wilkinson")
check(np.ones(n), 1e-9 * rng.standard_normal(n - 1), "clustered (tiny e)")
e0 = rng.standard_normal(n - 1); e0[::7] = 0.0
check(rng.standard_normal(n), e0, "zeros in e")
check(10.0 ** (-np.arange(n) / 30.0), 10.0 ** (-np.arange(n - 1) / 30.0) * 0.5, "graded")
check(rng.standard_normal(n) * 1e150, rng.standard_normal(n - 1) * 1e150, "huge scale")
check(np.zeros(n), np.zeros(n - 1), "zero matrix")
for n in (2048, 4096):
    d = rng.standard_normal(n) * 50 + 100; e = rng.standard_normal(n - 1) * 30
    check(d, e, "random big")
    check(d, e, "random big (warm)")
