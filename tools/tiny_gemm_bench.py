import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
from eigensolver_gpu_amd import api
torch.cuda.set_device(0)
dt=torch.complex128
for (ta,tb,M,N,K) in [('N','N',64,64,64),('C','N',64,64,64),('N','N',64,1024,64),('C','N',1024,64,1024),('N','N',512,512,64),('N','N',2048,2048,64),('N','N',2048,2048,2048),('C','N',2048,2048,2048)]:
    A=torch.randn((max(M,K),max(M,K)),dtype=dt,device='cuda'); B=torch.randn((max(N,K),max(N,K)),dtype=dt,device='cuda'); C=torch.zeros((N,max(M,1)),dtype=dt,device='cuda')
    ms=api.gemm_bench(ta,tb,M,N,K,A,A.shape[1],B,B.shape[1],C,M,reps=200 if M*N*K<1e8 else 10)
    print("%s%s M=%5d N=%5d K=%5d: %8.2f us  %.2f TF"%(ta,tb,M,N,K,ms*1e3, 8.0*M*N*K/ms*1e-9))
