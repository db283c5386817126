import sys; sys.path.insert(0,'/root/repo')
import torch
from eigensolver_gpu_amd import api
torch.cuda.set_device(0)
dt=torch.complex128
M=N=K=int(sys.argv[1]) if len(sys.argv)>1 else 448
A=torch.randn((K,M),dtype=dt,device='cuda'); B=torch.randn((N,K),dtype=dt,device='cuda'); C=torch.zeros((N,M),dtype=dt,device='cuda')
ms=api.gemm_bench('N','N',M,N,K,A,M,B,K,C,M,reps=20)
print("gemm %d^3: %.2f us"%(M,ms*1e3))
