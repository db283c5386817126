import sys, os; sys.path.insert(0,'/root/repo')
import torch
from eigensolver_gpu_amd import api
torch.cuda.set_device(0)
dt=torch.complex128
for n in (4096,3000,2048,1024):
    nb=64
    V=torch.randn((nb,n),dtype=dt,device='cuda'); W=torch.randn((nb,n),dtype=dt,device='cuda'); C=torch.randn((n,n),dtype=dt,device='cuda')
    ms=api.her2k_bench(V,W,C,n,nb,reps=20)
    print("tile=%s her2k n=%d k=64: %.1f us %.1f TF"%(os.environ.get("EIGSOLVE_GEMM_TILE","auto"),n,ms*1e3,4*2.0*n*n*nb/ms*1e-9))
