#!/usr/bin/env python3
"""Median per-phase ms of isolated solves (options through EIGSOLVE_<NAME> in the environment).
Usage: python tools/iso_phases.py [n] [m] [real] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import gen_pair  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = int(sys.argv[2]) if len(sys.argv) > 2 else n // 4
cplx = not (len(sys.argv) > 3 and sys.argv[3] == "real")
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
A0, B0 = gen_pair(n, cplx, 1002, dev)
ws = api.Workspace(n, cplx)
rows = []
for r in range(1 + reps):
    A, B = A0.clone(), B0.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info, _ = api.hegvdx(A, B, 1, m, ws)
    wall = (time.perf_counter() - t0) * 1e3
    assert info == 0
    if r > 0:
        ph = api.phase_times()
        ph["wall"] = wall
        rows.append(ph)
keys = ["potrf", "gst", "trd", "stedc_host", "backtransform", "trsm", "d2h", "wall"]
med = {k: sorted(x[k] for x in rows)[len(rows) // 2] for k in keys}
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("EIGSOLVE_") and k != "EIGSOLVE_GPU_LIB")
print("%-40s %s" % (tag or "(defaults)", "  ".join("%s %7.3f" % (k, med[k]) for k in keys)), flush=True)
