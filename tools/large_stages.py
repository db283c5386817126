#!/usr/bin/env python3
"""Stage-by-stage check at a LARGE order through size-independent identities evaluated on the device (no oracle: too slow there).
Usage: python tools/large_stages.py [n] [real]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eigensolver_gpu_amd import api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
cplx = not (len(sys.argv) > 2 and sys.argv[2] == "real")
dt = torch.complex128 if cplx else torch.float64
torch.cuda.set_device(0)
g = torch.Generator(device="cuda").manual_seed(7)
nrm = torch.linalg.norm


def cm(M):   # math matrix -> column-major device tensor (shape[1] = leading dimension)
    return M.T.contiguous()


def herm(n):
    R = torch.randn((n, n), dtype=dt, device="cuda", generator=g)
    return (R + R.conj().T) * 0.5


tag = ("z" if cplx else "d") + str(n)
# ---- products
m = int(os.environ.get("STAGE_M", "1024"))
A = torch.randn((n, n), dtype=dt, device="cuda", generator=g)
Bm = torch.randn((n, m), dtype=dt, device="cuda", generator=g)
for ta in ("N", "C"):
    C = torch.zeros((m, n), dtype=dt, device="cuda")
    api.gemm(ta, "N", n, m, n, 1.0, cm(A), n, cm(Bm), n, 0.0, C, n)
    ref = (A if ta == "N" else A.conj().T) @ Bm
    print(tag, "gemm", ta, "N  rel err %.2e" % float(nrm(C.T - ref) / nrm(ref)), flush=True)
    del C, ref
x = torch.randn(n, dtype=dt, device="cuda", generator=g)
H = herm(n)
y = api.hemv(cm(torch.triu(H)), x)
print(tag, "hemv rel err %.2e" % float(nrm(y - H @ x) / nrm(H @ x)), flush=True)
del A
# ---- potrf
T = torch.randn((n, n), dtype=dt, device="cuda", generator=g)
B = T @ T.conj().T / n + torch.eye(n, dtype=dt, device="cuda")
del T
Bd = cm(torch.triu(B))
info = api.potrf(Bd)
U = torch.triu(Bd.T)
print(tag, "potrf info", info, "||U^H U - B|| / ||B|| %.2e" % float(nrm(U.conj().T @ U - B) / nrm(B)), flush=True)
# ---- trsm
Zd = cm(Bm.clone())
api.trsm_lun(Bd, Zd, m)
print(tag, "trsm_lun ||U X - Z|| / ||Z|| %.2e" % float(nrm(U @ Zd.T - Bm) / nrm(Bm)), flush=True)
del Zd
# ---- hegst
Ad = cm(torch.triu(H))
api.hegst(Ad, Bd)
Cu = torch.triu(Ad.T)
Cf = Cu + torch.triu(Cu, 1).conj().T
print(tag, "hegst ||U^H C U - A|| / ||A|| %.2e" % float(nrm(U.conj().T @ Cf @ U - H) / nrm(H)), flush=True)
del Cu, Ad, U, Bd, B
# ---- standard eigenproblem (hetrd + tridiagonal solver + back-transformation)
ws = api.Workspace(n, cplx, pinned=False)
info, _ = api.heevd(cm(torch.triu(Cf)), 1, m, ws)
Z = ws.Z[:m].T
lam = ws.w[:m].to(dt)
print(tag, "heevd info", info, "||C Z - Z L|| / (||C|| ||Z||) %.2e   ||Z^H Z - I||max %.2e" % (
    float(nrm(Cf @ Z - Z * lam[None, :]) / (nrm(Cf) * nrm(Z))),
    float((Z.conj().T @ Z - torch.eye(m, dtype=dt, device="cuda")).abs().max())), flush=True)
# ---- tridiagonalisation alone: the tridiagonal matrix must have the eigenvalues of C (first m against heevd's)
Ad = cm(torch.triu(Cf))
d, e, tau = api.hetrd(Ad)
Tm = torch.diag(d.cpu()) + torch.diag(e.cpu(), 1) + torch.diag(e.cpu(), -1) if n <= 4096 else None
print(tag, "hetrd trace(T) - trace(C) rel %.2e   ||T||_F vs ||C||_F rel %.2e" % (
    float(abs(d.sum() - torch.diagonal(Cf).real.sum()) / nrm(torch.diagonal(Cf).real)),
    float(abs(torch.sqrt((d * d).sum() + 2 * (e * e).sum()) - nrm(Cf)) / nrm(Cf))), flush=True)
