#!/usr/bin/env python3
"""Large odd orders (tail loops of the panel kernels, remainder panels, non-power-of-two recursion splits) on the
well-conditioned family: residual must meet N*eps.  Usage: check_large_odd.py N [real|cplx] [m]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import eigensolver_gpu_amd.api as api


def gen_spd(n, seed, cplx, shift=0.0):
    """Reference recipe (test_zhegvdx.F90:41-59): Hermitian T with uniform[0,1) entries, A = T T^H (+ shift I)."""
    r = np.random.default_rng(seed)
    T = r.random((n, n)) + (1j * r.random((n, n)) if cplx else 0.0)
    T = np.tril(T, -1) + np.tril(T, -1).conj().T + np.diag(r.random(n))
    M = T @ T.conj().T
    return M + shift * np.eye(n)

N = int(sys.argv[1]); cplx = (len(sys.argv) < 3 or sys.argv[2] != "real"); m = int(sys.argv[3]) if len(sys.argv) > 3 else 64
A = gen_spd(N, 11, cplx)
B = gen_spd(N, 12, cplx, shift=float(N))
Ad, Bd = api.to_device(np.triu(A)), api.to_device(np.triu(B))
t0 = time.perf_counter()
info, ws = api.hegvdx(Ad, Bd, 1, m)
t1 = time.perf_counter()
assert info == 0
w = ws.w_h.numpy()[:N].copy()
Z = np.asfortranarray(api.to_host(ws.Z_h, N, m)).copy()
R = A @ Z - (B @ Z) * w[:m]
res = np.linalg.norm(R) / np.linalg.norm(A)
orth = np.abs(Z.conj().T @ (B @ Z) - np.eye(m)).max()
print("N=%d %s m=%d: %.1f ms  residual %.3e (N*eps %.3e)  B-orth %.3e  w[0..2]=%s" % (N, "z" if cplx else "d", m, (t1 - t0) * 1e3, res, N * 2.2e-16, orth, w[:3]))
assert res <= N * 2.2e-16 and orth <= 1e-11
print("OK")
