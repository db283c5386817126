import sys, time; sys.path.insert(0,'/root/repo')
import torch, numpy as np
from eigensolver_gpu_amd import api
torch.cuda.set_device(0)
for cplx in (True, False):
    n=4096
    dt=torch.complex128 if cplx else torch.float64
    T=torch.randn((n,n),dtype=dt,device='cuda'); B0=T@T.conj().T+n*torch.eye(n,dtype=dt,device='cuda')
    for rep in range(3):
        B=B0.clone(); torch.cuda.synchronize(); t0=time.perf_counter(); info=api.potrf(B); torch.cuda.synchronize(); t=time.perf_counter()-t0
    U=torch.triu(B.T)  # math orientation
    err=float(torch.linalg.norm(U.conj().T@U-B0.T)/torch.linalg.norm(B0))
    print("potrf %s N=%d: %.2f ms info=%d err=%.1e  (%.1f TF)"%("z" if cplx else "d",n,t*1e3,info,err,(4 if cplx else 1)*n**3/3/t*1e-12))
