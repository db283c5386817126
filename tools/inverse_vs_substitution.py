"""Blocked triangular solves through EXPLICIT INVERSE diagonal blocks (any order, up to the whole factor) vs substitution,
on the reference recipe (cond(B) ~ 1e10): residual and B-orthonormality of the generalized eigenpairs are the same.
Usage: python tools/inverse_vs_substitution.py [N]   (numpy/scipy only, CPU)"""
import os, sys, numpy as np, scipy.linalg as sl
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from scipy.linalg import lapack
n=int(sys.argv[1]) if len(sys.argv)>1 else 2048
A=oracle.gen_spd_fast(n,1003,True); B=oracle.gen_spd_fast(n,2003,True)
print("cond(B)=%.2e"%np.linalg.cond(B))
U,info=lapack.zpotrf(B,lower=0,clean=1)
def blk_inv_solve_left(U, X, base, herm=False):
    # X <- U^-1 X (or U^-H X) by recursion down to explicit inverses of base x base diagonal blocks
    n=U.shape[0]
    if n<=base:
        inv=np.linalg.inv(np.triu(U)) if False else sl.solve_triangular(U, np.eye(n), lower=False)
        return (inv.conj().T if herm else inv) @ X
    n1=((n//base+1)//2)*base
    if not herm:
        X2=blk_inv_solve_left(U[n1:,n1:],X[n1:],base)
        X1=X[:n1]-U[:n1,n1:]@X2
        X1=blk_inv_solve_left(U[:n1,:n1],X1,base)
    else:
        X1=blk_inv_solve_left(U[:n1,:n1],X[:n1],base,True)
        X2=X[n1:]-U[:n1,n1:].conj().T@X1
        X2=blk_inv_solve_left(U[n1:,n1:],X2,base,True)
    return np.vstack([X1,X2])
# C = U^-H A U^-1 two ways
def hegst_subst():
    F=sl.solve_triangular(U,A,trans='C',lower=False)
    return sl.solve_triangular(U,F.conj().T,trans='C',lower=False).conj().T
def hegst_inv(base):
    F=blk_inv_solve_left(U,A,base,True)
    return blk_inv_solve_left(U,F.conj().T,base,True).conj().T
def metrics(Z,w):
    BZ=B@Z; R=A@Z-BZ*w
    return np.linalg.norm(R)/np.linalg.norm(A), np.linalg.norm(Z.conj().T@BZ-np.eye(n))
for name,C in (("subst",hegst_subst()),("inv64",hegst_inv(64)),("inv256",hegst_inv(256))):
    C=0.5*(C+C.conj().T)
    w,Q=np.linalg.eigh(C)
    for tn,Z in (("trsm-subst",sl.solve_triangular(U,Q,lower=False)),("trsm-inv64",blk_inv_solve_left(U,Q,64)),("trsm-inv256",blk_inv_solve_left(U,Q,256))):
        r,b=metrics(Z,w)
        print("hegst %-7s %-11s residual %.3e bortho %.3e"%(name,tn,r,b),flush=True)
wl,Zl=sl.eigh(A,B,driver='gvd'); print("LAPACK gvd  residual %.3e bortho %.3e"%metrics(Zl,wl))
