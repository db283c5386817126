import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from eigensolver_gpu_amd import api
torch.cuda.set_device(0)
def rnd(*shape):
    return torch.complex(torch.randn(shape, dtype=torch.float64, device="cuda"), torch.randn(shape, dtype=torch.float64, device="cuda"))
big = 1100
A = rnd(big, big); B = rnd(big, big)
for ta, tb, M, N, K, al, be in [("N", "C", 1000, 1030, 130, 0.7-0.2j, 0.3+0.1j), ("N", "C", 1000, 1030, 130, 1.0, 0.0), ("N", "C", 1000, 1030, 130, 0.7-0.2j, 0.0), ("N", "C", 1000, 1030, 130, 1.0, 0.3+0.1j),
                                   ("N", "C", 1024, 1024, 130, 0.7-0.2j, 0.3+0.1j), ("N", "N", 1000, 1030, 130, 0.7-0.2j, 0.3+0.1j), ("C", "C", 1100, 1100, 17, 0.7-0.2j, 0.3+0.1j), ("C", "C", 1088, 1088, 17, 0.7-0.2j, 0.3+0.1j)]:
    outs = []
    for mode in (0, 1):
        api.set_option("gemm_dma", mode)
        C = torch.full((big, big), 0.5, dtype=torch.complex128, device="cuda")
        api.gemm(ta, tb, M, N, K, al, A, big, B, big, be, C, big)
        outs.append(C.cpu().numpy())
    d = np.abs(outs[0] - outs[1])
    idx = np.argwhere(d > 0)
    print(ta, tb, M, N, K, al, be, "ndiff", len(idx), "max", d.max(), "rows(colmajor j) range", (idx[:,0].min(), idx[:,0].max()) if len(idx) else None, "cols(i)", (idx[:,1].min(), idx[:,1].max()) if len(idx) else None, flush=True)
