#!/usr/bin/env python3
"""Time of the tridiagonalization with the one-workgroup finish at order 128 / 192 (option trd_finish = -1, default) against
the reference's cut-over at 32 (trd_finish = 32): whole hetrd calls through the C ABI (host wall clock around a synchronous
call, min of `reps`), for orders that fit the finish kernel entirely and for the benchmark orders.
Usage: python tools/trd_finish_bench.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)


def herm(n, cplx):
    x = torch.randn((n, n), dtype=torch.float64, device=dev)
    if cplx:
        x = torch.complex(x, torch.randn((n, n), dtype=torch.float64, device=dev))
    return (x + x.conj().T).contiguous()


for cplx, orders in ((True, (64, 128, 256, 512, 768, 1024, 2048, 4096)), (False, (96, 192, 384, 768, 1024, 2048))):
    for n in orders:
        A0 = herm(n, cplx)
        res = {}
        for fin in (32, 64, -1):
            api.set_option("trd_finish", fin)
            best = 1e9
            for r in range(4):
                A = A0.clone()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                api.hetrd(A)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) * 1e3)
            res[fin] = best
        api.set_option("trd_finish", -1)
        print("%s n=%5d  hetrd: finish at 32: %8.3f ms   one workgroup from 64: %8.3f ms   one workgroup from 128/192: %8.3f ms   saved %7.3f ms" %
              ("z" if cplx else "d", n, res[32], res[64], res[-1], res[32] - res[-1]), flush=True)
