import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from eigensolver_gpu_amd import api
torch.cuda.set_device(0)
for N in (8192, 4096):
    A = torch.randn((N, N), dtype=torch.complex128, device="cuda")
    x = torch.randn(N, dtype=torch.complex128, device="cuda")
    for hb in (0, 256, 384, 512, 768):
        api.set_option("hemv_blocks", hb)
        out = []
        for n in (N, N * 3 // 4, N // 2):
            ms = api.hemv_bench(A, x, reps=50, n=n)
            out.append("n=%d %.1f us %.2f TB/s" % (n, ms * 1e3, 16 * n * (n + 1) / 2 / (ms * 1e-3) * 1e-12))
        r = api.hetrd_mv_sweep(A.clone(), 0, reps=1)
        print("N=%d hemv_blocks=%d: %s | sweep %.1f ms %.2f TB/s" % (N, hb, "  ".join(out), r["ms_total"], r["algo_bytes"] / (r["ms_total"] * 1e-3) * 1e-12), flush=True)
    api.set_option("hemv_blocks", 0)
