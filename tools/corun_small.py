#!/usr/bin/env python3
"""Do the 20 us, 256-workgroup products of different launch chains overlap?  T host threads (own library context / stream each), each
looping over the same small product; us per launch per thread for T = 1, 2, 4 (perfect overlap: constant; none: x T).
Usage: python tools/corun_small.py [real]"""
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

cplx = not (len(sys.argv) > 1 and sys.argv[1] == "real")
dt = torch.complex128 if cplx else torch.float64
torch.cuda.set_device(0)
big = 2048
bufs = [[torch.randn((big, big), dtype=dt, device="cuda") for _ in range(3)] for _ in range(4)]
torch.cuda.synchronize()
shapes = [("N", "N", 256, 1024, 256), ("N", "C", 512, 512, 256), ("N", "N", 1024, 1024, 64), ("N", "N", 1024, 1024, 1024), ("C", "N", 256, 512, 2048)]
for ta, tb, M, N, K in shapes:
    line = []
    for T in (1, 2, 4):
        res = [0.0] * T

        def work(t):
            torch.cuda.set_device(0)
            A, B, C = bufs[t]
            api.gemm_bench(ta, tb, M, N, K, A, big, B, big, C, big, reps=20)
            res[t] = api.gemm_bench(ta, tb, M, N, K, A, big, B, big, C, big, reps=1500) * 1e3

        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        t0 = time.time()
        for x in th: x.start()
        for x in th: x.join()
        line.append("T=%d: %s us" % (T, " ".join("%.1f" % r for r in res)))
    print("%s%s %dx%dx%d %s:  %s" % (ta, tb, M, N, K, "z" if cplx else "d", "   ".join(line)), flush=True)
