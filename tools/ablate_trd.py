import sys, os, time; sys.path.insert(0,'/root/repo')
import torch
from eigensolver_gpu_amd import api
torch.cuda.set_device(0)
n=4096
A=torch.randn((n,n),dtype=torch.complex128,device='cuda'); A=A+A.conj().T.contiguous()
for rep in range(2):
    B=A.clone(); torch.cuda.synchronize(); t0=time.time(); api.hetrd(B); torch.cuda.synchronize(); t=time.time()-t0
print("EIGSOLVE_ABLATE=%s hetrd N=4096: %.2f ms"%(os.environ.get("EIGSOLVE_ABLATE","0"), t*1e3))
