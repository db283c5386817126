#!/usr/bin/env python3
"""Generate the LAPACK golden fixtures under tests/golden/ (run in the build container).

The reference has no golden vectors of its own; its test programs compare against CPU
LAPACK ?hegvd (test_driver/test_zhegvdx.F90:172-179).  These fixtures pin the oracle and
the HIP path the same way, stage by stage, with scipy's LAPACK (OpenBLAS 0.3.29):

  A, B      inputs (reference recipe T*T^H, see oracle.gen_spd; `wc` family adds N*I to B)
  U         zpotrf/dpotrf(B, upper)
  C         zhegst/dsygst(itype=1, 'U', A, U)         (upper triangle significant)
  d, e, tau zhetrd/dsytrd('U', C)
  w, Zabs   zhegvd/dsygvd(itype=1,'V','U', A, B): all eigenvalues, |eigenvectors|

Usage:  python tests/golden/make_golden.py     (writes tests/golden/*.npz)
"""
import os
import sys

import numpy as np
from scipy.linalg import lapack

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402

CASES = [  # (name, N, complex, family)
    ("d33", 33, False, "ref"), ("z33", 33, True, "ref"),
    ("d64wc", 64, False, "wc"), ("z70wc", 70, True, "wc"),
    ("d96", 96, False, "ref"), ("z96", 96, True, "ref"),
    ("d256", 256, False, "ref"),   # BASELINE.json configs[0] (C1): dsygvdx N=256, eigenpairs 1..64
]


def make(name, n, cplx, fam):
    p = "z" if cplx else "d"
    A = oracle.gen_spd(n, 1000 + n, cplx)
    B = oracle.gen_spd(n, 2000 + n, cplx, shift=float(n) if fam == "wc" else 0.0)
    U, info = getattr(lapack, p + "potrf")(B, lower=0, clean=1)
    assert info == 0
    C, info = getattr(lapack, ("zhegst" if cplx else "dsygst"))(A, U, itype=1, lower=0)
    assert info == 0
    trd = getattr(lapack, "zhetrd" if cplx else "dsytrd")
    Ct, d, e, tau, info = trd(C, lower=0)
    assert info == 0
    gvd = getattr(lapack, "zhegvd" if cplx else "dsygvd")
    res = gvd(A, B, itype=1, jobz="V", uplo="U")
    Zv, w, info = res[0], res[1], res[-1]
    if Zv.ndim == 1:
        Zv, w = w, Zv
    assert info == 0
    np.savez_compressed(os.path.join(HERE, name + ".npz"), A=np.triu(A), B=np.triu(B), U=np.triu(U), C=np.triu(C),
                        d=d, e=e, tau=tau, w=w, Zabs=np.abs(Zv))
    print(name, "ok", "cond(B)=%.2e" % np.linalg.cond(oracle.herm_from_upper(B)))


if __name__ == "__main__":
    only = sys.argv[1:]
    for c in CASES:
        if not only or c[0] in only:
            make(*c)
