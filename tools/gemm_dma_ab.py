#!/usr/bin/env python3
"""A/B of the MFMA engine's two staging paths for the complex 64 x 64 tiles (option "gemm_dma": 0 = global -> registers -> ds_write,
1 = LDS-DMA): results must be bit-identical (same k order, same lane mapping); rates on the shapes of the C3 solve.
Usage: python tools/gemm_dma_ab.py [check] [rate] [solve]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

torch.cuda.set_device(0)
what = sys.argv[1:] or ["check", "rate", "solve"]
dt = torch.complex128


def rnd(*shape):
    return torch.complex(torch.randn(shape, dtype=torch.float64, device="cuda"), torch.randn(shape, dtype=torch.float64, device="cuda"))


if "check" in what:
    bad = 0
    big = 1100
    A = rnd(big, big); B = rnd(big, big)
    for ta, tb, M, N, K in [("N", "N", 1024, 1024, 1024), ("C", "N", 1024, 512, 777), ("N", "C", 1000, 1030, 130), ("T", "T", 999, 513, 64),
                            ("C", "C", 1100, 1100, 17), ("N", "N", 1027, 515, 1100), ("C", "N", 640, 640, 16)]:
        outs = []
        for mode in (0, 1):
            api.set_option("gemm_dma", mode)
            C = torch.full((big, big), 0.5, dtype=dt, device="cuda")
            api.gemm(ta, tb, M, N, K, 0.7 - 0.2j, A, big, B, big, 0.3 + 0.1j, C, big)
            outs.append(C.cpu().numpy())
        f = {"N": lambda x: x, "T": lambda x: x.T, "C": lambda x: x.conj().T}
        # column-major views of the row-major tensors
        Ac = A.cpu().numpy().T; Bc = B.cpu().numpy().T
        opA = f[ta](Ac[:M, :K] if ta == "N" else Ac[:K, :M]); opB = f[tb](Bc[:K, :N] if tb == "N" else Bc[:N, :K])
        ref = (0.7 - 0.2j) * (opA @ opB) + (0.3 + 0.1j) * 0.5
        got = outs[1].T[:M, :N]
        err = np.abs(got - ref).max() / np.abs(ref).max()
        same = np.array_equal(outs[0], outs[1])
        bad += (not same) or not (err < 1e-12)
        print("gemm %s%s %5d %5d %5d: dma == reg: %s  rel err %.1e" % (ta, tb, M, N, K, same, err), flush=True)
    for n, k in ((2048, 64), (1999, 32), (4032, 17)):
        V = rnd(k, n); W = rnd(k, n); C0 = rnd(n, n)
        outs = []
        for mode in (0, 1):
            api.set_option("gemm_dma", mode)
            C = C0.clone()
            api.her2k(V, W, C, n, k)
            outs.append(C.cpu().numpy())
        same = np.array_equal(outs[0], outs[1])
        bad += not same
        print("her2k n=%d k=%d: dma == reg: %s" % (n, k, same), flush=True)
    print("CHECK", "FAILED" if bad else "ok", flush=True)

if "phases" in what:
    import oracle
    n, m = (8192, 8192) if "c4" in what else (4096, 1024)
    A = oracle.gen_spd_fast(n, 11, True); B = oracle.gen_spd_fast(n, 12, True) + n * np.eye(n)
    api.set_option("overlap", 0)
    for rep in range(2):
        for mode in (0, 1, 2):
            api.set_option("gemm_dma", mode)
            best = None
            for it in range(3):
                info, ws = api.hegvdx(api.to_device(A), api.to_device(B), 1, m)
                ph = api.phase_times()
                if best is None or ph["total"] < best["total"]:
                    best = ph
            print("N=%d one stream gemm_dma=%d: potrf %.2f gst %.2f trd %.2f tridiag %.2f bt %.2f trsm %.2f total %.2f ms" %
                  (n, mode, best["potrf"], best["gst"], best["trd"], best["stedc_host"], best["backtransform"], best["trsm"], best["total"]), flush=True)
    api.set_option("overlap", -1)

if "solve" in what:
    import oracle
    for n, m in ((1500, 400), (2048, 512)):
        res = []
        for mode in (0, 2):
            api.set_option("gemm_dma", mode)
            A = oracle.gen_spd_fast(n, 11, True); B = oracle.gen_spd_fast(n, 12, True) + n * np.eye(n)
            info, ws = api.hegvdx(api.to_device(A), api.to_device(B), 1, m)
            res.append((info, ws.w_h.numpy()[:n].copy(), api.to_host(ws.Z_h, n, m)))
        same = res[0][0] == res[1][0] == 0 and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
        r = oracle.residual(A, B, res[1][1], res[1][2])
        print("zhegvdx N=%d m=%d: dma == reg: %s  residual %.2e (N eps %.2e)" % (n, m, same, r, n * 2.2e-16), flush=True)

if "rate" in what:
    big = 4096
    A = rnd(big, big); B = rnd(big, big); C = torch.empty((big, big), dtype=dt, device="cuda")
    shapes = [("N", "N", 4096, 4096, 4096), ("N", "N", 2048, 2048, 2048), ("N", "C", 4096, 1024, 256), ("N", "C", 4096, 4096, 64),
              ("N", "C", 4096, 4096, 128), ("C", "N", 1024, 256, 4096), ("C", "N", 4032, 4032, 128), ("C", "N", 2048, 2048, 64),
              ("N", "N", 1024, 2048, 1024), ("C", "C", 2048, 2048, 2048), ("N", "C", 2048, 2048, 2048)]
    for ta, tb, M, N, K in shapes:
        row = []
        for rep in range(2):
            for mode in (0, 1, 2):
                api.set_option("gemm_dma", mode)
                ms = api.gemm_bench(ta, tb, M, N, K, A, big, B, big, C, big, reps=10)
                row.append("%s %8.1f us %5.1f TF" % (("reg", "dma", "dmaP")[mode], ms * 1e3, 8.0 * M * N * K / (ms * 1e-3) * 1e-12))
        print("%s%s %5d %5d %5d: %s" % (ta, tb, M, N, K, " | ".join(row)), flush=True)
    for n, k in ((4096, 64), (4096, 32), (3000, 32), (2048, 32)):
        V = rnd(k, n); W = rnd(k, n); Cn = rnd(n, n)
        row = []
        for rep in range(2):
            for mode in (0, 1, 2):
                api.set_option("gemm_dma", mode)
                ms = api.her2k_bench(V, W, Cn, n, k, reps=10)
                row.append("%s %7.1f us %5.1f TF" % (("reg", "dma", "dmaP")[mode], ms * 1e3, 8.0 * n * n * k / (ms * 1e-3) * 1e-12))
        print("her2k n=%d k=%d: %s" % (n, k, " | ".join(row)), flush=True)

if "lean" in what:
    # option "gemm_lean": K-slabs of 8, C fetched in the epilogue, three to four workgroups per CU (gemm_dma_kernel<., 8, 3>) against the
    # default choice (registers below 96 of K, LDS-DMA with slabs of 16 above)
    big = 4096
    A = rnd(big, big); B = rnd(big, big); C = torch.empty((big, big), dtype=dt, device="cuda")
    api.set_option("gemm_dma", -1)
    print("default | lean", flush=True)
    for ta, tb, M, N, K in [("N", "C", 4096, 4096, 64), ("N", "C", 4096, 4096, 128), ("C", "N", 4032, 4032, 128), ("N", "C", 4096, 1024, 256),
                            ("N", "C", 2048, 2048, 64), ("N", "C", 2048, 2048, 128), ("N", "N", 2048, 2048, 512), ("N", "N", 4096, 4096, 1024)]:
        row = []
        for lean in (0, 1 << 20, 0, 1 << 20):
            api.set_option("gemm_lean", lean)
            ms = api.gemm_bench(ta, tb, M, N, K, A, big, B, big, C, big, reps=10)
            row.append("%8.1f us %5.1f TF" % (ms * 1e3, 8.0 * M * N * K / (ms * 1e-3) * 1e-12))
        print("%s%s %5d %5d %5d: %s" % (ta, tb, M, N, K, " | ".join(row)), flush=True)
    for n, k in ((4096, 64), (4096, 32), (3000, 32), (2048, 32), (1024, 32)):
        V = rnd(k, n); W = rnd(k, n); Cn = rnd(n, n)
        row = []
        for lean in (0, 1 << 20, 0, 1 << 20):
            api.set_option("gemm_lean", lean)
            ms = api.her2k_bench(V, W, Cn, n, k, reps=10)
            row.append("%7.1f us %5.1f TF" % (ms * 1e3, 8.0 * n * n * k / (ms * 1e-3) * 1e-12))
        print("her2k n=%d k=%d: %s" % (n, k, " | ".join(row)), flush=True)
    api.set_option("gemm_lean", -1)
api.set_option("gemm_dma", -1)
