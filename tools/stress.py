#!/usr/bin/env python3
"""Randomised stress run through the C ABI: sizes, types, eigenpair ranges, leading dimensions and algorithm options;
every case must meet the residual / orthonormality gates of the parity tests.  Usage: stress.py [cases] [seed] [nmax]"""
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

EPS = 2.220446049250313e-16
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 900
rng = np.random.default_rng(seed)
bad = 0
t0 = time.time()
for case in range(cases):
    n = int(rng.choice([1, 2, 3, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 640, 700, 769, 1000, 1025, 1500][: 25])) if rng.random() < 0.6 else int(rng.integers(1, nmax + 1))
    n = min(n, nmax)
    cplx = bool(rng.integers(0, 2))
    il = int(rng.integers(1, n + 1)); iu = int(rng.integers(il, n + 1))
    if rng.random() < 0.5: il = 1
    m = iu - il + 1
    opts = {"tridiag": int(rng.integers(0, 2)), "bt_nb": int(rng.choice([64, 128])), "gst": int(rng.integers(0, 4)),
            "gst_thr": int(rng.choice([256, 512, 1024])), "trsm_base": int(rng.choice([64, 256, 256, 512, 1024])),
            "trd_nb": int(rng.choice([64, 32, 17])), "potrf": int(rng.choice([2, 2, 1, 0])), "zs_cap_mb": int(rng.choice([0, 0, 0, 1])), "trd_finish": int(rng.choice([-1, -1, 32, 64, 100])),
            "tile_map": int(rng.integers(0, 2)), "hemv_blocks": int(rng.choice([0, 0, 0, 3, 40, 512])),
            "batch_workers": int(rng.choice([-1, -1, 0, 2])), "batch_fuse": int(rng.choice([-1, -1, 1, 3])),
            # round 6: staging paths of the MFMA engine, the lean / whole-CU forms, the mat-vec's LDS-DMA ring, zipped group launches,
            # the look-ahead factorization
            "gemm_dma": int(rng.choice([3, 3, 0, 1, 2])), "gemm_lean": int(rng.choice([128, 128, 0, 1 << 20])), "gemm_wide": int(rng.choice([0, 0, 1, 2])),
            "mv_dma": int(rng.choice([0, 0, 1, 600])), "batch_zip": int(rng.choice([3, 3, 0, 1, 2])), "overlap": int(rng.choice([3, 3, 0, 7, 4]))}
    # one case in five goes through the batch entry point: 2-5 distinct problems of this order in one call, each checked
    # (the batch interface carries no host workspaces: it needs the device tridiagonal solver and says so otherwise)
    nprob = int(rng.integers(2, 6)) if (rng.random() < 0.2 and n <= 700) else 1
    if nprob > 1: opts["tridiag"] = 1
    for k, v in opts.items(): assert api.set_option(k, v) == 0
    probs = [(gen_spd(n, 100 + case + 1000 * q, cplx), gen_spd(n, 200 + case + 1000 * q, cplx, shift=float(n))) for q in range(nprob)]
    if nprob == 1:
        info, ws = api.hegvdx(api.to_device(np.triu(probs[0][0])), api.to_device(np.triu(probs[0][1])), il, iu)
        infos, wss = [info], [ws]
    else:
        wss = [api.Workspace(n, cplx) for _ in range(nprob)]
        infos = api.hegvdx_batch([(api.to_device(np.triu(a)), api.to_device(np.triu(b))) for a, b in probs], il, iu, wss)
    ok, res, orth = True, 0.0, 0.0
    for (A, B), info, ws in zip(probs, infos, wss):
        w = ws.w_h.numpy()[:n].copy(); Z = np.asfortranarray(api.to_host(ws.Z_h, n, m)).copy()
        R = A @ Z - (B @ Z) * w[il - 1:iu]
        res = max(res, np.linalg.norm(R) / np.linalg.norm(A))
        orth = max(orth, np.abs(Z.conj().T @ (B @ Z) - np.eye(m)).max())
        srt = bool(np.all(np.diff(w) >= 0))
        ok = ok and info == 0 and res <= max(n, 4) * EPS and orth <= 1e-11 and srt and bool(np.all(np.isfinite(Z)))
    if not ok:
        bad += 1
    print("%s case %3d n=%4d %s il=%4d iu=%4d batch=%d %s res=%.2e orth=%.2e" % ("ok " if ok else "BAD", case, n, "z" if cplx else "d", il, iu, nprob, opts, res, orth), flush=True)
for k in ("tridiag", "gst"): api.set_option(k, -1)
for k in ("bt_nb", "gst_thr", "trsm_base", "trd_nb"): api.set_option(k, 0)
api.set_option("potrf", -1); api.set_option("trd_finish", -1); api.set_option("tile_map", 1); api.set_option("zs_cap_mb", 0)
api.set_option("hemv_blocks", 0); api.set_option("batch_workers", -1); api.set_option("batch_fuse", -1)
for k in ("gemm_dma", "gemm_lean", "gemm_wide", "mv_dma", "overlap"): api.set_option(k, -1)
api.set_option("batch_zip", 3)
print("%d cases, %d bad, %.1f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
