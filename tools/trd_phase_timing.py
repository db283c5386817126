#!/usr/bin/env python3
"""Phase stamps (shader clock) inside panel_row_kernel / panel_mv_kernel, from a -DEIG_TRD_TIMING=1 build
of trd.hip linked as gpurun_out/libeigsolve_timing.so.  Debug tool only."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eigensolver_gpu_amd.api as api
# (build: make -C eigensolver_gpu_amd/csrc OUTDIR=../lib/v_timing EXTRA=-DEIG_TRD_TIMING=1; run with EIGSOLVE_GPU_LIB=.../lib/v_timing/libeigsolve_gpu.so)
import torch

def run(n, cx=True):
    rng = np.random.default_rng(1)
    A = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cx else 0)
    A = A + A.conj().T
    lib = api.lib()
    out0 = (ctypes.c_ulonglong * 36)()
    lib.eigsolve_debug_trd_timing(out0)
    Ad = torch.from_numpy(np.ascontiguousarray(A.T)).cuda()
    r = api.hetrd(Ad)
    torch.cuda.synchronize()
    out1 = (ctypes.c_ulonglong * 36)()
    lib.eigsolve_debug_trd_timing(out1)
    d = [out1[i] - out0[i] for i in range(36)]
    for k, name in ((0, "mv "), (2, "mv+"), (1, "row")):
        cnt = d[k * 9]
        ph = [d[k * 9 + 1 + p] / max(cnt, 1) for p in range(8)]
        print("n=%5d %s launches=%5d  cumulative cycles at stamps: %s" % (n, name, cnt, " ".join("%7.0f" % x for x in ph)))

print("EIGSOLVE_MV_DMA =", os.environ.get("EIGSOLVE_MV_DMA"))
for n in (1024, 1400, 2048, 4096):
    run(n)
run(1024, cx=False)
run(2048, cx=False)
