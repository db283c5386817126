#!/usr/bin/env python3
"""Shader-clock stamps inside hetd2_multi_kernel (a -DEIG_TRD_TIMING=1 build: make -C eigensolver_gpu_amd/csrc
OUTDIR=../lib/v_timing EXTRA=-DEIG_TRD_TIMING=1), workgroup 0 / thread 0, averaged over the steps of one launch.
Usage: EIGSOLVE_GPU_LIB=.../lib/v_timing/libeigsolve_gpu.so python tools/multi_finish_timing.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
names = ["scalars+v", "y rows", "publish", "poll", "read", "alpha+w", "update"]
for cplx, n in ((True, 768), (True, 512), (True, 256), (False, 1024), (False, 512)):
    x = torch.randn((n, n), dtype=torch.float64, device=dev)
    if cplx:
        x = torch.complex(x, torch.randn((n, n), dtype=torch.float64, device=dev))
    A = (x + x.conj().T).contiguous()
    lib = api.lib()
    o0 = (ctypes.c_ulonglong * 36)()
    lib.eigsolve_debug_trd_timing(o0)
    api.hetrd(A)
    torch.cuda.synchronize()
    o1 = (ctypes.c_ulonglong * 36)()
    lib.eigsolve_debug_trd_timing(o1)
    dlt = [o1[i] - o0[i] for i in range(36)]
    cnt = max(dlt[18], 1)
    cum = [dlt[19 + p] / cnt for p in range(7)]
    seg = [cum[0]] + [cum[i] - cum[i - 1] for i in range(1, 7)]
    print("%s n=%4d steps=%4d  cycles per step: %s   total %.0f" % ("z" if cplx else "d", n, cnt,
          "  ".join("%s %5.0f" % (nm, v) for nm, v in zip(names, seg)), cum[6]), flush=True)
