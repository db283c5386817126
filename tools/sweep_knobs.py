import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
from eigensolver_gpu_amd import api
torch.cuda.set_device(0)
n=4096
A=torch.randn((n,n),dtype=torch.complex128,device='cuda')
x=torch.randn(n,dtype=torch.complex128,device='cuda')
for hb in (0,256,512,768,1024,1536,2048):
    api.set_option("hemv_blocks",hb)
    r=api.hetrd_mv_sweep(A.clone(),0,reps=2)
    ms1=api.hemv_bench(A,x,reps=20)
    print("hemv_blocks=%5d sweep %.2f ms  %.0f GB/s | single n=4096 plain: %.1f us %.0f GB/s"%(hb,r["ms_total"],r["algo_bytes"]/r["ms_total"]*1e-6, ms1*1e3, 16*n*(n+1)/2/ms1*1e-6))
