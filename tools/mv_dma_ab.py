#!/usr/bin/env python3
"""A/B of the two data paths of the panel mat-vec kernel (option "mv_dma": 0 = tiles staged through registers, 1 = LDS-DMA ring
for every order): results must be bit-identical; single-launch rate at a few orders, the mat-vec sweep of one tridiagonalization,
and the whole tridiagonalization.  Usage: python tools/mv_dma_ab.py [check] [rate] [trd]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

torch.cuda.set_device(0)
what = sys.argv[1:] or ["check", "rate", "trd"]


def herm(n, cplx, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn((n, n), generator=g, dtype=torch.float64, device="cuda")
    if cplx:
        a = torch.complex(a, torch.randn((n, n), generator=g, dtype=torch.float64, device="cuda"))
    return a + a.conj().T + 2.0 * n * torch.eye(n, dtype=a.dtype, device="cuda")


if "check" in what:
    bad = 0
    for cplx in (True, False):
        for n in (1, 5, 63, 64, 65, 129, 200, 777, 1500, 2049, 4096):
            A = herm(n, cplx, n)
            x = torch.randn(n, dtype=A.dtype, device="cuda")
            ys = []
            for mode in (0, 1):
                api.set_option("mv_dma", mode)
                ys.append(api.hemv(A, x).cpu().numpy())
            ref = (A.conj() @ x).cpu().numpy()        # (the library reads the row-major tensor as column-major: A^T = conj(A))
            err = np.abs(ys[1] - ref).max() / np.abs(ref).max()
            same = np.array_equal(ys[0], ys[1])
            bad += (not same) or not (err < 1e-12)
            print("hemv %s n=%5d  dma == reg: %s  rel err %.1e" % ("z" if cplx else "d", n, same, err), flush=True)
        for n in (97, 411, 1300, 2100):
            A0 = herm(n, cplx, 7 * n)
            outs = []
            for mode in (0, 1):
                api.set_option("mv_dma", mode)
                Ad = A0.clone()
                d, e, tau = api.hetrd(Ad)
                outs.append((d.cpu().numpy(), e.cpu().numpy(), tau.cpu().numpy(), Ad.cpu().numpy()))
            same = all(np.array_equal(a, b) for a, b in zip(*outs))
            bad += not same
            print("hetrd %s n=%5d  dma == reg: %s  finite: %s" % ("z" if cplx else "d", n, same, bool(np.isfinite(outs[1][0]).all())), flush=True)
    print("CHECK", "FAILED" if bad else "ok", flush=True)

if "rate" in what:
    for cplx in (True, False):
        s = 16 if cplx else 8
        N = 8192
        A = herm(N, cplx, 1)
        x = torch.randn(N, dtype=A.dtype, device="cuda")
        for n in (1024, 1536, 2048, 3072, 4096, 6144, 8192):
            row = []
            for rnd in range(2):
                for mode in (0, 1):
                    api.set_option("mv_dma", mode)
                    ms = api.hemv_bench(A, x, reps=40, n=n)
                    row.append("%s %6.1f us %5.2f TB/s" % ("dma" if mode else "reg", ms * 1e3, s * n * (n + 1) / 2 / (ms * 1e-3) * 1e-12))
            print("hemv %s n=%5d: %s" % ("z" if cplx else "d", n, " | ".join(row)), flush=True)

if "trd" in what:
    for cplx, N in ((True, 4096), (True, 8192), (False, 2048), (True, 2048), (False, 8192)):
        A0 = herm(N, cplx, 3)
        for thr in (0, 1, 1024, 2048, 3072):
            api.set_option("mv_dma", thr)
            r = api.hetrd_mv_sweep(A0.clone(), 0, reps=2)
            ts = []
            for rep in range(3):
                Ad = A0.clone()
                torch.cuda.synchronize()
                import time
                t0 = time.perf_counter()
                api.hetrd(Ad)
                ts.append((time.perf_counter() - t0) * 1e3)
            print("%s N=%d mv_dma=%4d: sweep %7.2f ms = %5.2f TB/s (%d launches) | hetrd %.2f ms (min of 3)" %
                  ("z" if cplx else "d", N, thr, r["ms_total"], r["algo_bytes"] / (r["ms_total"] * 1e-3) * 1e-12, r["launches"], min(ts)), flush=True)
api.set_option("mv_dma", -1)
