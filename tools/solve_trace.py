#!/usr/bin/env python3
"""One warm-up + `reps` isolated solves (default C3) -- run under `rocprofv3 --kernel-trace` to get an in-situ kernel
trace of a single solve; tools/trace_phases.py segments the last solve by phase.
Usage: rocprofv3 --kernel-trace -d OUT -o NAME -- python tools/solve_trace.py [n] [m] [reps] [real]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import gen_pair  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = int(sys.argv[2]) if len(sys.argv) > 2 else n // 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cplx = not (len(sys.argv) > 4 and sys.argv[4] == "real")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
A0, B0 = gen_pair(n, cplx, 1002, dev)
ws = api.Workspace(n, cplx)
for r in range(1 + reps):
    A, B = A0.clone(), B0.clone()
    torch.cuda.synchronize()
    info, _ = api.hegvdx(A, B, 1, m, ws)
    assert info == 0
    print("solve %d: %s" % (r, {k: round(v, 3) for k, v in api.phase_times().items()}), flush=True)
