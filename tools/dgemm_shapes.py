import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from eigensolver_gpu_amd import api
torch.cuda.set_device(0)
tag = os.path.basename(os.path.dirname(os.environ.get("EIGSOLVE_GPU_LIB", "product/x")))
for (M, N, K) in ((4096, 4096, 4096), (2048, 2048, 2048), (4096, 4096, 64), (2048, 2048, 128), (8192, 8192, 256), (1024, 512, 1024)):
    A = torch.randn((K, M), dtype=torch.float64, device="cuda"); B = torch.randn((N, K), dtype=torch.float64, device="cuda"); C = torch.empty((N, M), dtype=torch.float64, device="cuda")
    ms = api.gemm_bench("N", "N", M, N, K, A, M, B, K, C, M, reps=5)
    ref = (B @ A)          # column-major: C = A_cm * B_cm  <=> row-major C^T = B^T ... (tensor [N,M] = B[N,K] @ A[K,M])
    err = float((C - ref).abs().max() / ref.abs().max())
    print("%-8s dgemm %5d x %5d x %5d: %8.3f ms  %6.1f TFLOP/s  max rel err %.1e" % (tag, M, N, K, ms, 2.0 * M * N * K / (ms * 1e-3) * 1e-12, err), flush=True)
