import sys, time; sys.path.insert(0,'/root/repo')
import torch, numpy as np
import bench
from eigensolver_gpu_amd import api
torch.cuda.set_device(0); dev=torch.device('cuda',0)
def run(n,m,cplx,shift):
    A0,B0=bench.gen_pair(n,cplx,1000,dev,shift_b=shift)
    A=A0.clone(); B=B0.clone()
    info,ws=api.hegvdx(A,B,1,m)
    Zc=ws.Z[:m,:].T; wv=ws.w[:m]; Ah,Bh=A0.T,B0.T
    R=Ah@Zc-(Bh@Zc)*wv.to(Zc.dtype)[None,:]
    nA=torch.linalg.norm(Ah); nB=torch.linalg.norm(Bh)
    res=float(torch.linalg.norm(R)/nA)
    colres=torch.linalg.norm(R,dim=0)/((nA+wv.abs()*nB)*torch.linalg.norm(Zc,dim=0))
    G=Zc.conj().T@(Bh@Zc); bo=float(torch.linalg.norm(G-torch.eye(m,device=dev,dtype=G.dtype)))
    print("n=%d m=%d shift=%g: info=%d resid=%.2e  max backward err=%.2e  Bortho=%.2e  w[0]=%.3e w[m-1]=%.3e"%(n,m,shift,info,res,float(colres.max()),bo,float(wv[0]),float(wv[-1])),flush=True)
run(8192,8192,True,0.0)
run(8192,8192,True,8192.0)
run(4096,4096,True,0.0)
