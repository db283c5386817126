#!/usr/bin/env python3
"""LAPACK fixtures for the BASELINE.json full-size configurations the oracle is too slow for
(run once in the build container, minutes to an hour on 8 cores; outputs are eigenvalues and scalars only).

The inputs are NOT stored: they are regenerated from the seeds with oracle.gen_spd_fast (counter-based RNG), so a
fixture is a few hundred KB at most.

  c4_z8192wc.npz   C4 (zhegvdx N=8192, il=1, iu=N), well-conditioned family B += N*I:
                   w = all eigenvalues from LAPACK zhegvd (eigenvalues only)
  c4_z8192ref.npz  C4 on the reference recipe (cond(B) ~ 1e10): w, and LAPACK zhegvd's OWN residual
                   ||A Z - B Z diag(w)||_F / ||A||_F, max backward error and B-orthonormality on the same input
                   (the reference's driver judges against LAPACK the same way, test_zhegvdx.F90:172-179,298-299)
  c3f_z4096ref.npz same at N=4096 full spectrum
  c5_z2048.npz     C5 problem 0 (zhegvdx N=2048 m=512, both families): w[:512] from LAPACK zhegvx and LAPACK's own
                   residual / backward error / B-orthonormality
  c2_d2048.npz     C2 (dsygvdx N=2048 m=512), same content, real arithmetic (LAPACK dsygvx / dsygvd)
  c3_z4096.npz     C3 (zhegvdx N=4096 m=1024), same content (round 6: the reference recipe at the headline configuration)
"""
import os
import sys
import time

import numpy as np
import scipy.linalg as sl

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402


def lapack_metrics(A, B, w, Z):
    nA, nB = np.linalg.norm(A), np.linalg.norm(B)
    BZ = B @ Z
    R = A @ Z - BZ * w[None, :]
    res = np.linalg.norm(R) / nA
    berr = (np.linalg.norm(R, axis=0) / ((nA + np.abs(w) * nB) * np.linalg.norm(Z, axis=0))).max()
    G = Z.conj().T @ BZ
    bortho = np.linalg.norm(G - np.eye(Z.shape[1]))
    return res, berr, bortho


def full_with_vectors(name, n, seedA, seedB):
    t = time.time()
    A = oracle.gen_spd_fast(n, seedA, True)
    B = oracle.gen_spd_fast(n, seedB, True)
    w, Z = sl.eigh(A, B, driver="gvd")
    res, berr, bortho = lapack_metrics(A, B, w, Z)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), n=n, seedA=seedA, seedB=seedB, shift=0.0, w=w,
                        lapack_residual=res, lapack_backward_error=berr, lapack_b_orthonormality=bortho)
    print(name, "ok res=%.3e berr=%.3e bortho=%.3e  %.0f s" % (res, berr, bortho, time.time() - t), flush=True)


def values_only(name, n, seedA, seedB, shift):
    t = time.time()
    A = oracle.gen_spd_fast(n, seedA, True)
    B = oracle.gen_spd_fast(n, seedB, True, shift=shift)
    w = sl.eigh(A, B, eigvals_only=True, driver="gvd")
    np.savez_compressed(os.path.join(HERE, name + ".npz"), n=n, seedA=seedA, seedB=seedB, shift=shift, w=w)
    print(name, "ok  %.0f s" % (time.time() - t), flush=True)


def c5_case(name, n, m, seedA, seedB, cplx=True):
    t = time.time()
    out = {}
    for fam, shift in (("wc", float(n)), ("ref", 0.0)):
        A = oracle.gen_spd_fast(n, seedA, cplx)
        B = oracle.gen_spd_fast(n, seedB, cplx, shift=shift)
        w, Z = sl.eigh(A, B, subset_by_index=[0, m - 1], driver="gvx")
        res, berr, bortho = lapack_metrics(A, B, w, Z)
        out["w_" + fam] = w
        out["lapack_residual_" + fam], out["lapack_backward_error_" + fam], out["lapack_b_orthonormality_" + fam] = res, berr, bortho
        print(fam, "LAPACK zhegvx residual %.3e berr %.3e bortho %.3e" % (res, berr, bortho), flush=True)
        # zhegvd (divide & conquer, all eigenpairs) is what the reference's driver compares with (test_zhegvdx.F90:172-179)
        # and what its own algorithm is (host zstedc): on the ill-conditioned recipe its low eigenpairs are far less
        # accurate than zhegvx's (bisection + inverse iteration) -- the gate for a D&C-based solver is zhegvd
        wd, Zd = sl.eigh(A, B, driver="gvd")
        res, berr, bortho = lapack_metrics(A, B, wd[:m], Zd[:, :m])
        out["gvd_residual_" + fam], out["gvd_backward_error_" + fam], out["gvd_b_orthonormality_" + fam] = res, berr, bortho
        print(fam, "LAPACK zhegvd (first m) residual %.3e berr %.3e bortho %.3e" % (res, berr, bortho), flush=True)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), n=n, m=m, seedA=seedA, seedB=seedB, **out)
    print(name, "ok  %.0f s" % (time.time() - t), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["c5", "c3f", "c4wc", "c4ref"]
    if "c5" in which:
        c5_case("c5_z2048", 2048, 512, 1004, 2004)
    if "c2" in which:
        c5_case("c2_d2048", 2048, 512, 1002, 2002, cplx=False)
    if "c3" in which:
        c5_case("c3_z4096", 4096, 1024, 1003, 2003)
    if "c3f" in which:
        full_with_vectors("c3f_z4096ref", 4096, 1003, 2003)
    if "c4wc" in which:
        values_only("c4_z8192wc", 8192, 1003, 2003, 8192.0)
    if "c4ref" in which:
        full_with_vectors("c4_z8192ref", 8192, 1003, 2003)
