#!/usr/bin/env python3
"""BASELINE configs[4] on 8 GPUs leaves 8 problems (zhegvdx N=2048 m=512) per GPU: rate of ONE batch call of `nprob` problems on one GPU
for the library's chain / lockstep-group settings.  Usage: python tools/c5_small_batch.py [nprob]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import gen_pair, problem_seed  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, m = 2048, 512
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
base = [gen_pair(n, True, problem_seed(4, p, 0), dev) for p in range(nprob)]
wss = [api.Workspace(n, True) for _ in range(nprob)]
for workers, fuse in ((-1, -1), (4, 1), (4, 2), (2, 4), (3, 3), (4, 4), (2, 2), (1, 4)):
    api.set_option("batch_workers", workers)
    api.set_option("batch_fuse", fuse)
    best = 1e9
    for rep in range(4):
        pairs = [(a.clone(), b.clone()) for a, b in base]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        infos = api.hegvdx_batch(pairs, 1, m, wss)
        dt = time.perf_counter() - t0
        assert not any(infos)
        if rep > 0:
            best = min(best, dt)
    print("nprob %2d  workers %2d fuse %2d : %7.2f ms  %6.1f problems/s" % (nprob, workers, fuse, best * 1e3, nprob / best), flush=True)
api.set_option("batch_workers", -1)
api.set_option("batch_fuse", -1)
