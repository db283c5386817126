import sys; sys.path.insert(0,'/root/repo')
import torch
from eigensolver_gpu_amd import api
torch.cuda.set_device(0)
n=64; dt=torch.complex128
T=torch.randn((n,n),dtype=dt,device='cuda'); B0=T@T.conj().T+n*torch.eye(n,dtype=dt,device='cuda')
for r in range(3):
    B=B0.clone(); info=api.potrf(B)
print("ok",info)
