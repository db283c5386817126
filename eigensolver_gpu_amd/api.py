"""Host-side mirror of the reference's Fortran interface, over the C ABI (include/eigsolve_gpu.h).

The reference's public surface is four Fortran modules (``zhegvdx_gpu``, ``dsygvdx_gpu``,
``eigsolve_vars``, ``nvtx_inters``); the Fortran drop-in shims live in
``eigensolver_gpu_amd/fortran/``.  This module exposes the *same procedures with the same
argument order and meaning* to Python through ``ctypes`` so that the parity tests read like
the reference's own test programs (test_driver/test_zhegvdx.F90:266-303).

PyTorch is used only as plumbing (device allocation / pinned host memory); every number is
computed by the hand-written HIP kernels in ``libeigsolve_gpu.so``.  There is NO CPU or
PyTorch fallback: if the shared library is missing, importing the entry points raises.

Matrix convention: column-major, exactly as the Fortran reference.  A device matrix with
leading dimension ``ld`` and ``ncol`` columns is a torch tensor of shape ``(ncol, ld)``
(row-major storage of the transpose == column-major storage of the matrix).
"""
import ctypes
import glob
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# EIGSOLVE_GPU_LIB: another build of the SAME library (A/B measurements of compile-time variants, tools/); never a fallback
LIB_PATH = os.environ.get("EIGSOLVE_GPU_LIB") or os.path.join(_HERE, "lib", "libeigsolve_gpu.so")
_lib = None

c_int = ctypes.c_int
c_void_p = ctypes.c_void_p
c_double_p = ctypes.POINTER(ctypes.c_double)

# every symbol include/eigsolve_gpu.h declares (tests check that the library exports them all)
EXPORTS = [
    "eigsolve_init", "eigsolve_finalize", "eigsolve_set_lapack", "eigsolve_set_host_threads", "eigsolve_set_option",
    "eigsolve_range_push", "eigsolve_range_pop", "eigsolve_get_phase_times", "eigsolve_zhegvdx", "eigsolve_dsygvdx",
    "eigsolve_zheevd", "eigsolve_dsyevd", "eigsolve_zhegst", "eigsolve_dsygst", "eigsolve_zhetrd", "eigsolve_dsytrd",
    "eigsolve_zpotrf", "eigsolve_dpotrf", "eigsolve_zhemv", "eigsolve_dsymv", "eigsolve_zhemv_bench",
    "eigsolve_dsymv_bench", "eigsolve_zgemm", "eigsolve_dgemm", "eigsolve_zgemm_bench", "eigsolve_dgemm_bench",
    "eigsolve_zher2k", "eigsolve_dsyr2k", "eigsolve_zher2k_bench", "eigsolve_dsyr2k_bench", "eigsolve_ztrsm_lun",
    "eigsolve_dtrsm_lun", "eigsolve_version", "eigsolve_zhetrd_mv_sweep", "eigsolve_dsytrd_mv_sweep", "eigsolve_zhetrd_her2k_sweep", "eigsolve_dsytrd_her2k_sweep",
    "eigsolve_dstedc_device", "eigsolve_zlarft", "eigsolve_dlarft", "eigsolve_zunmtr", "eigsolve_dormtr",
    "eigsolve_zhegvdx_batch", "eigsolve_dsygvdx_batch",
]


class EigsolveLibraryMissing(RuntimeError):
    pass


def find_host_lapack():
    """Path of a shared library exporting dstedc (the reference links MKL/LAPACK,
    test_driver/Makefile:20-23; this image has only scipy's bundled OpenBLAS)."""
    env = os.environ.get("EIGSOLVE_LAPACK_LIB")
    if env:
        return env
    try:
        import scipy
        cands = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so"))
        cands = [c for c in cands if "64_" not in os.path.basename(c)]
        if cands:
            return os.path.realpath(cands[0])
    except Exception:
        pass
    return None


def lib():
    """Loads libeigsolve_gpu.so (built by __graft_entry__.build()).  Fails loudly if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EigsolveLibraryMissing(
                "%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        _lib.eigsolve_version.restype = ctypes.c_char_p
        lp = find_host_lapack()
        if lp:
            _lib.eigsolve_set_lapack(lp.encode())
    return _lib


def _p(t):
    """Raw address of a torch tensor / numpy array / None."""
    if t is None:
        return c_void_p(0)
    if isinstance(t, np.ndarray):
        return c_void_p(t.ctypes.data)
    return c_void_p(t.data_ptr())


def _sync():
    """Inputs produced by torch kernels must be complete before the library's own stream touches them.
    Only the calling thread's current torch stream is synchronised (NOT the device): several host threads
    may drive independent solves on the same GPU concurrently (one context per (thread, device))."""
    import torch
    torch.cuda.current_stream().synchronize()


# ---- module eigsolve_vars -------------------------------------------------------------------------
def init_eigsolve_gpu():
    """eigsolve_vars.F90:39-59."""
    return lib().eigsolve_init()


def finalize():
    """eigsolve_finalize: releases the calling thread's context for the current device, the contexts of the library's
    worker threads and of finished threads, and the idle streams of the pool.  The next call re-creates what it needs."""
    return lib().eigsolve_finalize()


# ---- module nvtx_inters ---------------------------------------------------------------------------
def nvtxStartRange(name, color_id=0):
    """lib_eigsolve/toolbox.F90:71-89 (roctx range on this platform)."""
    lib().eigsolve_range_push(name.encode(), c_int(color_id))


def nvtxEndRange():
    """lib_eigsolve/toolbox.F90:91-97."""
    lib().eigsolve_range_pop()


# ---- module zhegvdx_gpu / dsygvdx_gpu: same argument order as the Fortran subroutines -------------
def zhegvdx_gpu(N, A, lda, B, ldb, Z, ldz, il, iu, w, work, lwork, rwork, lrwork, work_h, lwork_h, rwork_h, lrwork_h,
                iwork_h, liwork_h, Z_h, ldz_h, w_h, _skip_host_copy=False):
    """zhegvdx_gpu.F90:75-76.  Device arrays: A,B,Z,w,work,rwork (torch cuda tensors);
    host arrays: work_h,rwork_h,iwork_h,Z_h,w_h (pinned torch tensors or numpy).  Returns info."""
    _sync()
    info = c_int(0)
    lib().eigsolve_zhegvdx(c_int(N), _p(A), c_int(lda), _p(B), c_int(ldb), _p(Z), c_int(ldz), c_int(il), c_int(iu), _p(w),
                           _p(work), c_int(lwork), _p(rwork), c_int(lrwork), _p(work_h), c_int(lwork_h), _p(rwork_h),
                           c_int(lrwork_h), _p(iwork_h), c_int(liwork_h), _p(Z_h), c_int(ldz_h), _p(w_h),
                           ctypes.byref(info), c_int(1 if _skip_host_copy else 0))
    return info.value


def dsygvdx_gpu(N, A, lda, B, ldb, Z, ldz, il, iu, w, work, lwork, work_h, lwork_h, iwork_h, liwork_h, Z_h, ldz_h, w_h,
                _skip_host_copy=False):
    """dsygvdx_gpu.F90:71-72.  Returns info."""
    _sync()
    info = c_int(0)
    lib().eigsolve_dsygvdx(c_int(N), _p(A), c_int(lda), _p(B), c_int(ldb), _p(Z), c_int(ldz), c_int(il), c_int(iu), _p(w),
                           _p(work), c_int(lwork), _p(work_h), c_int(lwork_h), _p(iwork_h), c_int(liwork_h), _p(Z_h),
                           c_int(ldz_h), _p(w_h), ctypes.byref(info), c_int(1 if _skip_host_copy else 0))
    return info.value


# ---- convenience layer (what test_zhegvdx.F90's main program does around the call) ----------------
def to_device(a):
    """numpy (n_rows, n_cols) any order -> torch cuda tensor (n_cols, n_rows) == column-major on device."""
    import torch
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a).T)).cuda()


def to_host(t, nrows=None, ncols=None):
    """inverse of to_device: returns a Fortran-ordered numpy view (n_rows, n_cols)."""
    a = t.detach().cpu().numpy().T
    if nrows is not None:
        a = a[:nrows]
    if ncols is not None:
        a = a[:, :ncols]
    return a


class Workspace:
    """Workspaces sized exactly to the reference's minima (SURVEY.md 8(b) size contract),
    allocated once and reused like the reference's test driver does (test_zhegvdx.F90:276-285)."""

    def __init__(self, N, is_complex, pinned=True):
        import torch
        self.N, self.cx = N, is_complex
        dt = torch.complex128 if is_complex else torch.float64
        dev = "cuda"
        if is_complex:
            self.lwork = 2 * 64 * 64 + 65 * N
            self.lrwork = N
            self.lwork_h = N
            self.lrwork_h = 1 + 5 * N + 2 * N * N
        else:
            self.lwork = 2 * 64 * 64 + 66 * N
            self.lwork_h = 1 + 6 * N + 2 * N * N
        self.liwork_h = 3 + 5 * N
        self.work = torch.empty(self.lwork, dtype=dt, device=dev)
        self.rwork = torch.empty(self.lrwork, dtype=torch.float64, device=dev) if is_complex else None
        self.w = torch.zeros(N, dtype=torch.float64, device=dev)
        self.Z = torch.zeros((N, N), dtype=dt, device=dev)

        def host(n, d):
            t = torch.empty(n, dtype=d)
            return t.pin_memory() if pinned else t

        self.work_h = host(self.lwork_h, dt)
        self.rwork_h = host(self.lrwork_h, torch.float64) if is_complex else None
        self.iwork_h = host(self.liwork_h, torch.int32)
        self.Z_h = host(N * N, dt).view(N, N)
        self.w_h = host(N, torch.float64)


def hegvdx(A_d, B_d, il, iu, ws=None, skip_host_copy=False):
    """Solve on device tensors (column-major, shape (N,N), lda=ldb=N).  A_d/B_d are overwritten
    like in the reference.  Returns (info, ws): eigenvalues ws.w / ws.w_h, vectors ws.Z / ws.Z_h."""
    import torch
    N = A_d.shape[0]
    cx = A_d.dtype == torch.complex128
    if ws is None:
        ws = Workspace(N, cx)
    if cx:
        info = zhegvdx_gpu(N, A_d, N, B_d, N, ws.Z, N, il, iu, ws.w, ws.work, ws.lwork, ws.rwork, ws.lrwork, ws.work_h,
                           ws.lwork_h, ws.rwork_h, ws.lrwork_h, ws.iwork_h, ws.liwork_h, ws.Z_h, N, ws.w_h, skip_host_copy)
    else:
        info = dsygvdx_gpu(N, A_d, N, B_d, N, ws.Z, N, il, iu, ws.w, ws.work, ws.lwork, ws.work_h, ws.lwork_h, ws.iwork_h,
                           ws.liwork_h, ws.Z_h, N, ws.w_h, skip_host_copy)
    return info, ws


def hegvdx_batch(pairs, il, iu, wss, skip_host_copy=False, null_entry=None):
    """nprob problems of ONE order and type in one call (eigsolve_zhegvdx_batch / eigsolve_dsygvdx_batch): the library keeps
    `batch_workers` of them in flight on its own worker threads (option 0: lockstep tridiagonalizations on the caller's
    context); per-problem results are bit-identical to `hegvdx`.  pairs = [(A_d, B_d), ...] (overwritten like in the
    reference), wss = one Workspace per problem.  Returns the list of per-problem info values.  null_entry (tests): name of
    a pointer array whose second entry is passed as NULL."""
    import torch
    _sync()
    nprob = len(pairs)
    assert nprob >= 1 and len(wss) >= nprob
    N = pairs[0][0].shape[0]
    cx = pairs[0][0].dtype == torch.complex128

    def arr(ts):
        return (c_void_p * nprob)(*[t.data_ptr() for t in ts])

    A, B = arr([p[0] for p in pairs]), arr([p[1] for p in pairs])
    Z, w, work = arr([ws.Z for ws in wss[:nprob]]), arr([ws.w for ws in wss[:nprob]]), arr([ws.work for ws in wss[:nprob]])
    Zh, wh = arr([ws.Z_h for ws in wss[:nprob]]), arr([ws.w_h for ws in wss[:nprob]])
    if null_entry == "Z_h":
        Zh[1] = None
    elif null_entry == "w_h":
        wh[1] = None
    info = (c_int * nprob)()
    ws0 = wss[0]
    if cx:
        rwork = arr([ws.rwork for ws in wss[:nprob]])
        lib().eigsolve_zhegvdx_batch(c_int(nprob), c_int(N), A, c_int(N), B, c_int(N), Z, c_int(N), c_int(il), c_int(iu), w, work,
                                     c_int(ws0.lwork), rwork, c_int(ws0.lrwork), Zh, c_int(N), wh, info,
                                     c_int(1 if skip_host_copy else 0))
    else:
        lib().eigsolve_dsygvdx_batch(c_int(nprob), c_int(N), A, c_int(N), B, c_int(N), Z, c_int(N), c_int(il), c_int(iu), w, work,
                                     c_int(ws0.lwork), Zh, c_int(N), wh, info, c_int(1 if skip_host_copy else 0))
    return [info[q] for q in range(nprob)]


def diaghg(H_d, S_d, m, ws=None):
    """The QE / LAXlib call pattern (cdiaghg_gpu / rdiaghg_gpu, external to the reference repo; README.md:2,11):
    H v = e S v with H, S resident on the device and left INTACT, lowest m eigenpairs, results consumed on the
    device (`_skip_host_copy=.true.`, zhegvdx_gpu.F90:171-180).  Returns (info, e_d[:m], v_d[:, :m] as an (m, N)
    column-major view, ws).  Mirrors eigensolver_gpu_amd/fortran/laxlib_glue.F90."""
    N = H_d.shape[0]
    info, ws = hegvdx(H_d.clone(), S_d.clone(), 1, m, ws=ws, skip_host_copy=True)
    _sync()
    return info, ws.w[:m], ws.Z[:m], ws


def phase_times():
    """Per-phase ms of the last driver call: potrf, gst, trd, stedc(host), back-transform, trsm, d2h, total."""
    buf = (ctypes.c_double * 8)()
    n = lib().eigsolve_get_phase_times(buf, c_int(8))
    names = ["potrf", "gst", "trd", "stedc_host", "backtransform", "trsm", "d2h", "total"]
    return {names[i]: buf[i] for i in range(n)}


def set_option(name, value):
    return lib().eigsolve_set_option(name.encode(), c_int(value))


def set_host_threads(n):
    return lib().eigsolve_set_host_threads(c_int(n))


# ---- stage / kernel level (public module procedures of the reference + vendor call sites) ---------
def _pre(t):
    import torch
    return "z" if t.dtype == torch.complex128 else "d"


def potrf(B_d):
    """Upper Cholesky in place; returns LAPACK-style info."""
    _sync()
    N = B_d.shape[0]
    info = c_int(0)
    rc = getattr(lib(), "eigsolve_%spotrf" % _pre(B_d))(c_int(N), _p(B_d), c_int(B_d.shape[1]), ctypes.byref(info))
    assert rc == 0
    return info.value


def hegst(A_d, U_d):
    """zhegst_gpu / dsygst_gpu (itype=1, 'U') in place on A_d."""
    _sync()
    N = A_d.shape[0]
    name = "eigsolve_zhegst" if _pre(A_d) == "z" else "eigsolve_dsygst"
    rc = getattr(lib(), name)(c_int(N), _p(A_d), c_int(A_d.shape[1]), _p(U_d), c_int(U_d.shape[1]), c_int(448))
    assert rc == 0


def hetrd(A_d, nb=0):
    """zhetrd_gpu / dsytrd_gpu ('U') in place; returns (d, e, tau) device tensors."""
    import torch
    _sync()
    N = A_d.shape[0]
    d = torch.zeros(N, dtype=torch.float64, device="cuda")
    e = torch.zeros(max(N - 1, 1), dtype=torch.float64, device="cuda")
    tau = torch.zeros(max(N - 1, 1), dtype=A_d.dtype, device="cuda")
    name = "eigsolve_zhetrd" if _pre(A_d) == "z" else "eigsolve_dsytrd"
    rc = getattr(lib(), name)(c_int(N), _p(A_d), c_int(A_d.shape[1]), _p(d), _p(e), _p(tau), c_void_p(0), c_int(0), c_int(nb))
    assert rc == 0
    return d, e[: N - 1], tau[: N - 1]


def hemv(A_d, x_d, n=None):
    """y = A x with A Hermitian, upper triangle stored (zhemv_gpu / dsymv_gpu)."""
    import torch
    _sync()
    n = A_d.shape[0] if n is None else n
    y = torch.zeros(n, dtype=A_d.dtype, device="cuda")
    name = "eigsolve_zhemv" if _pre(A_d) == "z" else "eigsolve_dsymv"
    rc = getattr(lib(), name)(c_int(n), _p(A_d), c_int(A_d.shape[1]), _p(x_d), _p(y))
    assert rc == 0
    return y


def hemv_bench(A_d, x_d, reps=20, n=None):
    """Average ms per launch of the HBM-bound hemv/symv kernel (HIP events on the library stream)."""
    import torch
    _sync()
    n = A_d.shape[0] if n is None else n
    y = torch.zeros(n, dtype=A_d.dtype, device="cuda")
    ms = ctypes.c_double(0)
    name = "eigsolve_zhemv_bench" if _pre(A_d) == "z" else "eigsolve_dsymv_bench"
    rc = getattr(lib(), name)(c_int(n), _p(A_d), c_int(A_d.shape[1]), _p(x_d), _p(y), c_int(reps), ctypes.byref(ms))
    assert rc == 0
    return ms.value


def _scal(v, cx):
    v = complex(v)
    return (ctypes.c_double * 2)(v.real, v.imag) if cx else (ctypes.c_double * 1)(v.real)


def gemm(ta, tb, M, N, K, alpha, A_d, lda, B_d, ldb, beta, C_d, ldc):
    """C = alpha op(A) op(B) + beta C on the fp64 MFMA engine (cublas?gemm_v2 call sites)."""
    _sync()
    cx = _pre(C_d) == "z"
    name = "eigsolve_zgemm" if cx else "eigsolve_dgemm"
    rc = getattr(lib(), name)(ctypes.c_char(ta.encode()), ctypes.c_char(tb.encode()), c_int(M), c_int(N), c_int(K),
                              _scal(alpha, cx), _p(A_d), c_int(lda), _p(B_d), c_int(ldb), _scal(beta, cx), _p(C_d), c_int(ldc))
    assert rc == 0


def gemm_bench(ta, tb, M, N, K, A_d, lda, B_d, ldb, C_d, ldc, reps=10):
    _sync()
    cx = _pre(C_d) == "z"
    ms = ctypes.c_double(0)
    name = "eigsolve_zgemm_bench" if cx else "eigsolve_dgemm_bench"
    rc = getattr(lib(), name)(ctypes.c_char(ta.encode()), ctypes.c_char(tb.encode()), c_int(M), c_int(N), c_int(K), _p(A_d),
                              c_int(lda), _p(B_d), c_int(ldb), _p(C_d), c_int(ldc), c_int(reps), ctypes.byref(ms))
    assert rc == 0
    return ms.value


def her2k(V_d, W_d, C_d, n, k):
    """C(upper) -= V W^H + W V^H  (cublaszher2k / cublasdsyr2k, zhetrd_gpu.F90:67)."""
    _sync()
    name = "eigsolve_zher2k" if _pre(C_d) == "z" else "eigsolve_dsyr2k"
    rc = getattr(lib(), name)(c_int(n), c_int(k), _p(V_d), c_int(V_d.shape[1]), _p(W_d), c_int(W_d.shape[1]), _p(C_d),
                              c_int(C_d.shape[1]))
    assert rc == 0


def her2k_bench(V_d, W_d, C_d, n, k, reps=10):
    _sync()
    ms = ctypes.c_double(0)
    name = "eigsolve_zher2k_bench" if _pre(C_d) == "z" else "eigsolve_dsyr2k_bench"
    rc = getattr(lib(), name)(c_int(n), c_int(k), _p(V_d), c_int(V_d.shape[1]), _p(W_d), c_int(W_d.shape[1]), _p(C_d),
                              c_int(C_d.shape[1]), c_int(reps), ctypes.byref(ms))
    assert rc == 0
    return ms.value


def trsm_lun(U_d, Z_d, m):
    """Z(:, :m) <- U^-1 Z (cublasZtrsm L,U,N,N, zhegvdx_gpu.F90:169)."""
    _sync()
    N = U_d.shape[0]
    name = "eigsolve_ztrsm_lun" if _pre(Z_d) == "z" else "eigsolve_dtrsm_lun"
    rc = getattr(lib(), name)(c_int(N), c_int(m), _p(U_d), c_int(U_d.shape[1]), _p(Z_d), c_int(Z_d.shape[1]))
    assert rc == 0


def hetrd_mv_sweep(A_d, nb=0, reps=1):
    """Roofline leg: the panel mat-vec kernels of one full ?hetrd, back to back.
    Returns dict(ms_total, launches, algo_bytes).  Destroys the numerical content of A_d."""
    _sync()
    N = A_d.shape[0]
    ms = ctypes.c_double(0)
    nl = ctypes.c_long(0)
    by = ctypes.c_double(0)
    name = "eigsolve_zhetrd_mv_sweep" if _pre(A_d) == "z" else "eigsolve_dsytrd_mv_sweep"
    rc = getattr(lib(), name)(c_int(N), _p(A_d), c_int(A_d.shape[1]), c_int(nb), c_int(reps), ctypes.byref(ms),
                              ctypes.byref(nl), ctypes.byref(by))
    assert rc == 0
    return {"ms_total": ms.value, "launches": nl.value, "algo_bytes": by.value}


def hetrd_her2k_sweep(A_d, W_d, nb=0, reps=1):
    """Roofline leg: the trailing rank-2nb updates of one full ?hetrd, back to back (W_d: nb x N row-major = N x nb panel workspace).
    Returns dict(ms_total, launches, flops).  Overwrites A_d."""
    _sync()
    N = A_d.shape[0]
    ms = ctypes.c_double(0)
    nl = ctypes.c_long(0)
    fl = ctypes.c_double(0)
    name = "eigsolve_zhetrd_her2k_sweep" if _pre(A_d) == "z" else "eigsolve_dsytrd_her2k_sweep"
    rc = getattr(lib(), name)(c_int(N), _p(A_d), c_int(A_d.shape[1]), _p(W_d), c_int(nb), c_int(reps), ctypes.byref(ms),
                              ctypes.byref(nl), ctypes.byref(fl))
    assert rc == 0
    return {"ms_total": ms.value, "launches": nl.value, "flops": fl.value}


def bt_block(nb, N):
    """Reflectors per block the library uses for a requested width nb (bt_norm_nb in evd.hip): 64 / 128 / 256 / 512, no
    wider than the problem needs."""
    nb = 512 if nb >= 512 else 256 if nb >= 256 else 128 if nb >= 128 else 64
    while nb > 64 and nb // 2 >= N:
        nb //= 2
    return nb


def larft(A_d, tau_d, nb=256):
    """All T factors (zlarft_gpu, zheevd_gpu.F90:136-176) of a tridiagonalized A_d.  Returns a numpy array
    (nblk, ldt, ldt) of lower-triangular blocks (math orientation)."""
    import torch
    _sync()
    N = A_d.shape[0]
    k = N - 1
    nbe = bt_block(nb, N)
    ldt = nbe
    nblk = max(1, (k + nbe - 1) // nbe)
    T = torch.zeros((nblk, ldt, ldt), dtype=A_d.dtype, device="cuda")
    name = "eigsolve_zlarft" if _pre(A_d) == "z" else "eigsolve_dlarft"
    rc = getattr(lib(), name)(c_int(N), _p(A_d), c_int(A_d.shape[1]), _p(tau_d), c_int(nb), _p(T), c_int(ldt))
    assert rc == 0
    return T.cpu().numpy().transpose(0, 2, 1)


def unmtr(A_d, tau_d, Z_d, m, nb=256):
    """Z(:, :m) <- Q Z with Q from the reflectors in upper(A_d) (the zlarft_gpu / zlarfb_gpu loop, zheevd_gpu.F90:113-131)."""
    _sync()
    N = A_d.shape[0]
    name = "eigsolve_zunmtr" if _pre(A_d) == "z" else "eigsolve_dormtr"
    rc = getattr(lib(), name)(c_int(N), c_int(m), _p(A_d), c_int(A_d.shape[1]), _p(tau_d), _p(Z_d), c_int(Z_d.shape[1]),
                              c_int(nb))
    assert rc == 0


def stedc_device(d, e):
    """Device divide & conquer on a symmetric tridiagonal (numpy d[N], e[N-1]) -> (w, Q, ms)."""
    import torch
    _sync()
    N = len(d)
    dd = torch.from_numpy(np.ascontiguousarray(d, dtype=np.float64)).cuda()
    ed = torch.from_numpy(np.ascontiguousarray(np.r_[e, 0.0], dtype=np.float64)).cuda()
    w = torch.zeros(N, dtype=torch.float64, device="cuda")
    Q = torch.zeros((N, N), dtype=torch.float64, device="cuda")
    ms = ctypes.c_double(0)
    rc = lib().eigsolve_dstedc_device(c_int(N), _p(dd), _p(ed), _p(w), _p(Q), c_int(N), ctypes.byref(ms))
    return rc, w.cpu().numpy(), to_host(Q), ms.value


def heevd(A_d, il, iu, ws=None):
    """zheevd_gpu / dsyevd_gpu (zheevd_gpu.F90:32-134): standard problem A z = w z, jobz='V', uplo='U',
    eigenpairs il..iu.  A_d (upper triangle) is overwritten with the reflectors.  Returns (info, ws)."""
    import torch
    _sync()
    N = A_d.shape[0]
    cx = A_d.dtype == torch.complex128
    if ws is None:
        ws = Workspace(N, cx)
    info = c_int(0)
    if cx:
        lib().eigsolve_zheevd(c_int(il), c_int(iu), c_int(N), _p(A_d), c_int(A_d.shape[1]), _p(ws.Z), c_int(N), _p(ws.w),
                              _p(ws.work), c_int(ws.lwork), _p(ws.rwork), c_int(ws.lrwork), _p(ws.work_h), c_int(ws.lwork_h),
                              _p(ws.rwork_h), c_int(ws.lrwork_h), _p(ws.iwork_h), c_int(ws.liwork_h), _p(ws.Z_h), c_int(N),
                              _p(ws.w_h), ctypes.byref(info))
    else:
        lib().eigsolve_dsyevd(c_int(il), c_int(iu), c_int(N), _p(A_d), c_int(A_d.shape[1]), _p(ws.Z), c_int(N), _p(ws.w),
                              _p(ws.work), c_int(ws.lwork), _p(ws.work_h), c_int(ws.lwork_h), _p(ws.iwork_h),
                              c_int(ws.liwork_h), _p(ws.Z_h), c_int(N), _p(ws.w_h), ctypes.byref(info))
    return info.value, ws
