#!/usr/bin/env python3
"""Stage-by-stage GPU-vs-oracle diagnostic (prints errors instead of asserting). GPU box only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from eigensolver_gpu_amd import api

torch.cuda.set_device(0)
print("device:", torch.cuda.get_device_name(0), api.lib().eigsolve_version().decode())


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def run(n, cplx):
    tag = ("z" if cplx else "d") + str(n)
    rng = np.random.default_rng(n)
    dt = np.complex128 if cplx else np.float64

    def rnd(*s):
        x = rng.standard_normal(s)
        if cplx:
            x = x + 1j * rng.standard_normal(s)
        return np.asfortranarray(x.astype(dt))

    # gemm variants
    for ta, tb in (("N", "N"), ("C", "N"), ("N", "C"), ("T", "T")):
        M, N, K = n, max(n // 2, 1) + 3, n + 5
        A = rnd(M, K) if ta == "N" else rnd(K, M)
        B = rnd(K, N) if tb == "N" else rnd(N, K)
        C = rnd(M, N)
        opa = {"N": A, "T": A.T, "C": A.conj().T}[ta]
        opb = {"N": B, "T": B.T, "C": B.conj().T}[tb]
        al, be = (0.7 - 0.2j, 0.3 + 0.1j) if cplx else (0.7, 0.3)
        ref = al * (opa @ opb) + be * C
        Cd = api.to_device(C)
        api.gemm(ta, tb, M, N, K, al, api.to_device(A), A.shape[0], api.to_device(B), B.shape[0], be, Cd, M)
        print(tag, "gemm", ta, tb, "err %.2e" % rel(api.to_host(Cd), ref))
    # hemv
    A = oracle.gen_spd(n, 10 + n, cplx)
    x = rnd(n)
    y = api.hemv(api.to_device(np.triu(A)), torch.from_numpy(x).cuda())
    print(tag, "hemv err %.2e" % rel(y.cpu().numpy(), oracle.herm_from_upper(A) @ x))
    # her2k
    k = min(64, n)
    V, W, C = rnd(n, k), rnd(n, k), oracle.gen_spd(n, 3, cplx)
    Cd = api.to_device(np.triu(C))
    api.her2k(api.to_device(V), api.to_device(W), Cd, n, k)
    ref = C - V @ W.conj().T - W @ V.conj().T
    got = api.to_host(Cd)
    print(tag, "her2k err %.2e lower-untouched %s" % (rel(np.triu(got), np.triu(ref)), bool(np.all(np.tril(got, -1) == 0))))
    # potrf
    B = oracle.gen_spd(n, 2000 + n, cplx, shift=float(n))
    Bd = api.to_device(np.triu(B))
    info = api.potrf(Bd)
    U = np.triu(api.to_host(Bd))
    Uo, _ = oracle.potrf_upper(B)
    print(tag, "potrf info", info, "err %.2e" % rel(U, np.triu(Uo)))
    # trsm
    Zr = rnd(n, max(n // 3, 1))
    Zd = api.to_device(Zr)
    api.trsm_lun(Bd, Zd, Zr.shape[1])
    print(tag, "trsm_lun err %.2e" % rel(api.to_host(Zd), np.linalg.solve(np.triu(Uo), Zr)))
    # hegst
    A = oracle.gen_spd(n, 1000 + n, cplx)
    Ad = api.to_device(np.triu(A))
    api.hegst(Ad, Bd)
    Co = oracle.hegst(np.triu(A), np.triu(Uo))
    Cg = api.to_host(Ad)
    print(tag, "hegst err %.2e lower-untouched %s" % (rel(np.triu(Cg), np.triu(Co)), bool(np.all(np.tril(Cg, -1) == 0))))
    # hetrd
    Cin = np.triu(Co)
    Ad = api.to_device(Cin)
    d, e, tau = api.hetrd(Ad)
    Ao, do, eo, tauo = oracle.hetrd(Cin, nb=32)
    s = np.abs(Cin).max()
    print(tag, "hetrd d %.2e e %.2e tau %.2e V %.2e" % (np.abs(d.cpu().numpy() - do).max() / s, np.abs(e.cpu().numpy() - eo).max() / s if n > 1 else 0,
          np.abs(tau.cpu().numpy() - tauo).max() if n > 1 else 0, rel(np.triu(api.to_host(Ad), 1), np.triu(Ao, 1))))
    # full driver
    m = max(1, n // 4)
    B2 = oracle.gen_spd(n, 2000 + n, cplx, shift=float(n))
    t0 = time.time()
    info, ws = api.hegvdx(api.to_device(np.triu(A)), api.to_device(np.triu(B2)), 1, m)
    dt_ = time.time() - t0
    w = ws.w_h.numpy().copy()
    Z = np.asfortranarray(api.to_host(ws.Z_h, n, m))
    wo, Zo, _, _, io = oracle.hegvdx(A, B2, 1, m)
    print(tag, "hegvdx info", info, "l2w %.2e l2|Z| %.2e resid %.2e (n*eps %.2e) Bortho %.2e  %.1f ms" % (
        oracle.compare_1d(wo, w)[0], oracle.compare_abs2d(Zo, Z)[0], oracle.residual(A, B2, w, Z), n * 2.2e-16,
        oracle.b_orthonormality(B2, Z), dt_ * 1e3), api.phase_times())


sizes = [int(s) for s in sys.argv[1:]] or [33, 64, 100, 200, 333]
for n in sizes:
    for cplx in (False, True):
        try:
            run(n, cplx)
        except Exception as ex:  # noqa
            import traceback
            traceback.print_exc()
