#!/usr/bin/env python3
"""Do an MFMA-bound product and the HBM-bound mat-vec of two different launch chains share the CUs?
Two host threads, each with its own library context / stream: one loops over a product, the other over the mat-vec;
times alone and together.  Usage: python tools/corun.py"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

N = 4096
dt = torch.complex128
A = torch.randn((N, N), dtype=dt, device="cuda")
x = torch.randn(N, dtype=dt, device="cuda")
V = torch.randn((64, N), dtype=dt, device="cuda")   # column-major N x 64 (shape[1] = leading dimension)
W = torch.randn((64, N), dtype=dt, device="cuda")
C = torch.randn((N, N), dtype=dt, device="cuda")
Bm = torch.randn((N, N), dtype=dt, device="cuda")
C2 = torch.randn((N, N), dtype=dt, device="cuda")
torch.cuda.synchronize()


import ctypes  # noqa: E402
from ctypes import c_int  # noqa: E402

L = api.lib()
y1 = torch.zeros(N, dtype=dt, device="cuda")
y2 = torch.zeros(N, dtype=dt, device="cuda")
A2 = torch.randn((N, N), dtype=dt, device="cuda")    # a second matrix: two mat-vec chains must not share their operand in L2 / MALL
torch.cuda.synchronize()
P = api._p


# direct library calls (api.*_bench would start with a device-wide synchronize, which serialises the two threads)
def mv(reps):
    ms = ctypes.c_double(0)
    assert L.eigsolve_zhemv_bench(c_int(N), P(A), c_int(N), P(x), P(y1), c_int(reps), ctypes.byref(ms)) == 0
    return ms.value


def mv2(reps):
    ms = ctypes.c_double(0)
    assert L.eigsolve_zhemv_bench(c_int(N), P(A2), c_int(N), P(x), P(y2), c_int(reps), ctypes.byref(ms)) == 0
    return ms.value


def her2k(reps):
    ms = ctypes.c_double(0)
    assert L.eigsolve_zher2k_bench(c_int(N), c_int(64), P(V), c_int(N), P(W), c_int(N), P(C), c_int(N), c_int(reps), ctypes.byref(ms)) == 0
    return ms.value


def _gemm(ta, tb, M, Nn, K, Ad, Bd, Cd, reps):
    ms = ctypes.c_double(0)
    assert L.eigsolve_zgemm_bench(ctypes.c_char(ta), ctypes.c_char(tb), c_int(M), c_int(Nn), c_int(K), P(Ad), c_int(N), P(Bd), c_int(N),
                                  P(Cd), c_int(N), c_int(reps), ctypes.byref(ms)) == 0
    return ms.value


def gemm_cn(reps):   # potrf's rank-64 update shape: C -= B12^H B12, K = 64 ('C','N')
    return _gemm(b"C", b"N", N, N, 64, Bm, Bm, C2, reps)


def gemm_nc(reps):   # hegst / back-transformation shape 'N','C', K = 256
    return _gemm(b"N", b"C", N, 1024, 256, A, Bm, C2, reps)


def gemm_nn(reps):   # long-K plain product
    return _gemm(b"N", b"N", 2048, 2048, 2048, A, Bm, C2, reps)


def run(fa, ra, fb, rb):
    out = {}

    def t(name, f, r):
        out[name] = f(r)

    for f, r in ((fa, 3), (fb, 3)):
        f(r)
    t0 = time.perf_counter(); a = fa(ra); ta = time.perf_counter() - t0
    t0 = time.perf_counter(); b = fb(rb); tb = time.perf_counter() - t0
    torch.cuda.synchronize()
    th = [threading.Thread(target=t, args=("a", fa, ra)), threading.Thread(target=t, args=("b", fb, rb))]
    t0 = time.perf_counter()
    for q in th:
        q.start()
    for q in th:
        q.join()
    tt = time.perf_counter() - t0
    print("  alone: %-8s %7.1f us x %d = %6.1f ms | %-8s %7.1f us x %d = %6.1f ms | sum %6.1f ms" %
          (fa.__name__, a * 1e3, ra, ta * 1e3, fb.__name__, b * 1e3, rb, tb * 1e3, (ta + tb) * 1e3))
    print("  together: %6.1f ms wall  (%s %7.1f us, %s %7.1f us per launch)  -> %.2f of the sum" %
          (tt * 1e3, fa.__name__, out["a"] * 1e3, fb.__name__, out["b"] * 1e3, tt / (ta + tb)), flush=True)


print("her2k n=4096 k=64 ('N','N' instantiation) beside hemv n=4096")
run(her2k, 1000, mv, 4000)
print("rank-64 update 'C','N' beside hemv")
run(gemm_cn, 600, mv, 4000)
print("'N','C' 4096 x 1024 x 256 beside hemv")
run(gemm_nc, 600, mv, 4000)
print("zgemm 2048^3 'N','N' beside hemv")
run(gemm_nn, 200, mv, 4000)
print("hemv beside hemv (different matrices)")
run(mv, 4000, mv2, 4000)
