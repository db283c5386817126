import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
from eigensolver_gpu_amd import api
torch.cuda.set_device(0)
dt=torch.complex128
for (ta,tb,M,N,K) in [('N','N',64,64,64),('C','N',64,64,64),('N','N',64,1024,64),('C','N',1024,64,1024),('N','N',512,512,64),('N','N',2048,2048,64),('N','N',2048,2048,2048),('C','N',2048,2048,2048)]:
    A=torch.randn((max(M,K),max(M,K)),dtype=dt,device='cuda'); B=torch.randn((max(N,K),max(N,K)),dtype=dt,device='cuda'); C=torch.zeros((N,max(M,1)),dtype=dt,device='cuda')
    ms=api.gemm_bench(ta,tb,M,N,K,A,A.shape[1],B,B.shape[1],C,M,reps=200 if M*N*K<1e8 else 10)
    print("%s%s M=%5d N=%5d K=%5d: %8.2f us  %.2f TF"%(ta,tb,M,N,K,ms*1e3, 8.0*M*N*K/ms*1e-9))
# alternating kernels (different code objects) back-to-back: I-cache / dependency effects
import time
M=N=K=64
A=torch.randn((64,64),dtype=dt,device='cuda'); B=torch.randn((64,64),dtype=dt,device='cuda'); C=torch.zeros((64,64),dtype=dt,device='cuda')
def seq(variants, reps=300):
    torch.cuda.synchronize(); t0=time.perf_counter()
    lib=api.lib()
    import ctypes
    one=(ctypes.c_double*2)(1.0,0.0); zero=(ctypes.c_double*2)(0.0,0.0)
    for r in range(reps):
        for (ta,tb) in variants:
            pass
    return 0
# use gemm_bench on chains is not possible; time python-level api.gemm calls (sync each) is too slow -> use C entry with reps on one variant only.
for (ta,tb) in [('N','N'),('C','N'),('N','C'),('C','C')]:
    ms=api.gemm_bench(ta,tb,64,64,64,A,64,B,64,C,64,reps=300)
    print("single-variant %s%s 64^3: %.2f us"%(ta,tb,ms*1e3))
for (M,N,K) in [(128,128,128),(256,256,64),(64,4096,64),(448,448,448)]:
    A=torch.randn((max(M,K),max(M,K)),dtype=dt,device='cuda'); B=torch.randn((max(N,K),max(N,K)),dtype=dt,device='cuda'); C=torch.zeros((N,M),dtype=dt,device='cuda')
    ms=api.gemm_bench('N','N',M,N,K,A,A.shape[1],B,B.shape[1],C,M,reps=200)
    print("NN M=%d N=%d K=%d: %.2f us %.2f TF"%(M,N,K,ms*1e3,8.0*M*N*K/ms*1e-9))
