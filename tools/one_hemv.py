import sys; sys.path.insert(0,'/root/repo')
import torch
from eigensolver_gpu_amd import api
torch.cuda.set_device(0)
n=int(sys.argv[1]) if len(sys.argv)>1 else 4096
A=torch.randn((n,n),dtype=torch.complex128,device='cuda'); x=torch.randn(n,dtype=torch.complex128,device='cuda')
ms=api.hemv_bench(A,x,reps=3)
print("hemv n=%d %.2f us"%(n,ms*1e3))
