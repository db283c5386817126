import sys, time; sys.path.insert(0,'/root/repo')
import torch, numpy as np
import bench
from eigensolver_gpu_amd import api
torch.cuda.set_device(0); dev=torch.device('cuda',0)
def run(n,m,cplx,tri):
    api.set_option("tridiag",tri)
    A0,B0=bench.gen_pair(n,cplx,1000,dev)
    A=A0.clone(); B=B0.clone()
    t0=time.time(); info,ws=api.hegvdx(A,B,1,m); dt=time.time()-t0
    Zc=ws.Z[:m,:].T; wv=ws.w[:m]; Ah,Bh=A0.T,B0.T
    R=Ah@Zc-(Bh@Zc)*wv.to(Zc.dtype)[None,:]
    res=float(torch.linalg.norm(R)/torch.linalg.norm(Ah))
    G=Zc.conj().T@(Bh@Zc); bo=float(torch.linalg.norm(G-torch.eye(m,device=dev,dtype=G.dtype)))
    print("n=%d m=%d cplx=%d tridiag=%s: info=%d resid=%.2e Bortho=%.2e %.0f ms"%(n,m,cplx,"dev" if tri else "host",info,res,bo,dt*1e3),flush=True)
    del A,B,A0,B0,R,G
for n,m in ((6144,1536),(8192,2048),(8192,8192)):
    for tri in (1,0):
        run(n,m,True,tri)
# standalone D&C at 8192
from scipy.linalg import eigh_tridiagonal
rng=np.random.default_rng(0)
n=8192; d=rng.standard_normal(n)*50+100; e=rng.standard_normal(n-1)*30
rc,w,Q,ms=api.stedc_device(d,e)
wr=eigh_tridiagonal(d,e,eigvals_only=True)
Qt=torch.from_numpy(np.ascontiguousarray(Q)).cuda()
orth=float((Qt.T@Qt-torch.eye(n,device='cuda',dtype=torch.float64)).abs().max())
print("stedc_device n=8192 rc=%d |w-wr|/|w|=%.1e orth=%.1e %.1f ms"%(rc,np.abs(w-wr).max()/np.abs(wr).max(),orth,ms))
