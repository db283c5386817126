"""ctypes/numpy front-end of the C oracle (oracle.c / oracle_impl.h).

TEST INFRASTRUCTURE ONLY.  Every function mirrors one stage of the reference path; the
reference file:line each stage follows is cited in oracle_impl.h.  All matrices are
column-major (numpy ``order='F'``) float64 / complex128.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    """Compile oracle.c with gcc (seconds).  Building the checker is not using it."""
    src = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle_impl.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_u01.restype = ctypes.c_double
        _lib.oracle_u01.argtypes = [ctypes.c_uint64] * 4
    return _lib


def _pfx(dtype):
    return "z_" if np.dtype(dtype) == np.complex128 else "d_"


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f(a, dtype=None):
    a = np.asarray(a, dtype=dtype)
    return np.asfortranarray(a)


def _dt(is_complex):
    return np.complex128 if is_complex else np.float64


def u01(seed, i, j, part):
    return lib().oracle_u01(seed, i, j, part)


def gen_spd(n, seed, is_complex, shift=0.0):
    """Reference input recipe (test_zhegvdx.F90:28-66): T*T^H from a counter-based RNG."""
    A = np.zeros((n, n), dtype=_dt(is_complex), order="F")
    fn = getattr(lib(), _pfx(A.dtype) + "gen_spd")
    fn(ctypes.c_int(n), ctypes.c_uint64(seed), ctypes.c_double(shift), _ptr(A), ctypes.c_int(n))
    return A


def gen_spd_fast(n, seed, is_complex, shift=0.0):
    """Same recipe and the same uniform draws as gen_spd, vectorised in numpy
    (the T*T^H product goes through BLAS, so the result agrees with gen_spd to rounding,
    not bit-for-bit).  Used for the large configurations."""
    ii, jj = np.meshgrid(np.arange(n, dtype=np.uint64), np.arange(n, dtype=np.uint64), indexing="ij")

    def sm(x):
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))

    def draw(part):
        with np.errstate(over="ignore"):
            h = sm(np.full((n, n), seed, dtype=np.uint64))
            h = sm(h ^ (ii * np.uint64(0x100000001B3) + np.uint64(0x51)))
            h = sm(h ^ (jj * np.uint64(0xC2B2AE3D27D4EB4F) + np.uint64(0x2F)))
            h = sm(h ^ np.uint64(part + 0x9))
        return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    re = draw(0)
    L = np.tril(re, -1)
    if is_complex:
        L = L + 1j * np.tril(draw(1), -1)
    T = L + L.conj().T + np.diag(np.diag(re))
    A = T @ T.conj().T
    A = 0.5 * (A + A.conj().T)
    if is_complex:
        A[np.diag_indices(n)] = A.diagonal().real
    A[np.diag_indices(n)] += shift
    return np.asfortranarray(A)


def potrf_upper(B):
    B = _f(B).copy(order="F")
    n = B.shape[0]
    fn = getattr(lib(), _pfx(B.dtype) + "potrf_upper")
    fn.restype = ctypes.c_int
    info = fn(ctypes.c_int(n), _ptr(B), ctypes.c_int(n))
    return B, info


def hegst(A, U, nb=448):
    A = _f(A).copy(order="F")
    U = _f(U)
    n = A.shape[0]
    getattr(lib(), _pfx(A.dtype) + "hegst")(ctypes.c_int(n), _ptr(A), ctypes.c_int(n), _ptr(U),
                                             ctypes.c_int(n), ctypes.c_int(nb))
    return A


def hetrd(A, nb=32):
    """Returns (A_out, d, e, tau): reflectors in upper(A_out) as the reference leaves them."""
    A = _f(A).copy(order="F")
    n = A.shape[0]
    d = np.zeros(n)
    e = np.zeros(max(n, 1))
    tau = np.zeros(max(n, 1), dtype=A.dtype)
    W = np.zeros((max(n, 1), max(nb, 1)), dtype=A.dtype, order="F")
    getattr(lib(), _pfx(A.dtype) + "hetrd")(ctypes.c_int(n), _ptr(A), ctypes.c_int(n), _ptr(d), _ptr(e),
                                             _ptr(tau), _ptr(W), ctypes.c_int(nb))
    return A, d, e[: max(n - 1, 0)], tau[: max(n - 1, 0)]


def latrd(A, np_, nb):
    """One panel on the leading np_ x np_ block.  Returns (A_out, e, tau, W)."""
    A = _f(A).copy(order="F")
    n = A.shape[0]
    e = np.zeros(n)
    tau = np.zeros(n, dtype=A.dtype)
    W = np.zeros((n, nb), dtype=A.dtype, order="F")
    getattr(lib(), _pfx(A.dtype) + "latrd")(ctypes.c_int(np_), ctypes.c_int(nb), _ptr(A), ctypes.c_int(n),
                                             _ptr(e), _ptr(tau), _ptr(W), ctypes.c_int(n))
    return A, e, tau, W


def hetd2(A):
    A = _f(A).copy(order="F")
    n = A.shape[0]
    d = np.zeros(n)
    e = np.zeros(max(n, 1))
    tau = np.zeros(max(n, 1), dtype=A.dtype)
    getattr(lib(), _pfx(A.dtype) + "hetd2")(ctypes.c_int(n), _ptr(A), ctypes.c_int(n), _ptr(d), _ptr(e), _ptr(tau))
    return A, d, e[: max(n - 1, 0)], tau[: max(n - 1, 0)]


def steql(d, e):
    d = np.array(d, dtype=np.float64)
    n = d.shape[0]
    ee = np.zeros(max(n, 1))
    ee[: n - 1] = e[: n - 1]
    Q = np.zeros((n, n), order="F")
    fn = lib().oracle_steql
    fn.restype = ctypes.c_int
    info = fn(ctypes.c_int(n), _ptr(d), _ptr(ee), _ptr(Q), ctypes.c_int(n))
    return d, Q, info


def larft(V, tau, mi, K):
    V = _f(V)
    tau = np.ascontiguousarray(tau, dtype=V.dtype)
    T = np.zeros((K, K), dtype=V.dtype, order="F")
    getattr(lib(), _pfx(V.dtype) + "larft")(ctypes.c_int(mi), ctypes.c_int(K), _ptr(V), ctypes.c_int(V.shape[0]),
                                             _ptr(tau), _ptr(T), ctypes.c_int(K))
    return T


def larfb(V, T, C, mi, K):
    V = _f(V)
    T = _f(T)
    C = _f(C).copy(order="F")
    m = C.shape[1]
    getattr(lib(), _pfx(V.dtype) + "larfb")(ctypes.c_int(mi), ctypes.c_int(m), ctypes.c_int(K), _ptr(V),
                                             ctypes.c_int(V.shape[0]), _ptr(T), ctypes.c_int(T.shape[0]), _ptr(C),
                                             ctypes.c_int(C.shape[0]))
    return C


def heevd(A, il, iu, nb1=32, nb2=64):
    A = _f(A).copy(order="F")
    n = A.shape[0]
    Z = np.zeros((n, n), dtype=A.dtype, order="F")
    w = np.zeros(n)
    fn = getattr(lib(), _pfx(A.dtype) + "heevd")
    fn.restype = ctypes.c_int
    info = fn(ctypes.c_int(n), ctypes.c_int(il), ctypes.c_int(iu), _ptr(A), ctypes.c_int(n), _ptr(Z),
              ctypes.c_int(n), _ptr(w), ctypes.c_int(nb1), ctypes.c_int(nb2))
    return w, Z[:, : iu - il + 1], A, info


def hegvdx(A, B, il, iu):
    """Full driver.  Returns (w[N], Z[N,m], A_out, B_out(=U), info)."""
    A = _f(A).copy(order="F")
    B = _f(B).copy(order="F")
    n = A.shape[0]
    Z = np.zeros((n, n), dtype=A.dtype, order="F")
    w = np.zeros(n)
    fn = getattr(lib(), _pfx(A.dtype) + "hegvdx")
    fn.restype = ctypes.c_int
    info = fn(ctypes.c_int(n), _ptr(A), ctypes.c_int(n), _ptr(B), ctypes.c_int(n), _ptr(Z), ctypes.c_int(n),
              ctypes.c_int(il), ctypes.c_int(iu), _ptr(w))
    return w, Z[:, : iu - il + 1].copy(order="F"), A, B, info


def compare_1d(ref, got):
    """test_driver/toolbox.F90:36-83 -> (l2 relative error, max percent error)."""
    ref = np.ascontiguousarray(ref, dtype=np.float64)
    got = np.ascontiguousarray(got, dtype=np.float64)
    out = np.zeros(2)
    lib().oracle_compare_1d(ctypes.c_int(ref.shape[0]), _ptr(ref), _ptr(got), _ptr(out))
    return out[0], out[1]


def compare_abs2d(ref, got):
    """test_driver/toolbox.F90:85-176 (on |entries|) -> (l2 relative error, max percent error)."""
    ref = _f(ref)
    got = _f(got, dtype=ref.dtype)
    n, m = ref.shape
    out = np.zeros(2)
    getattr(lib(), _pfx(ref.dtype) + "compare_abs2d")(ctypes.c_int(n), ctypes.c_int(m), _ptr(ref), ctypes.c_int(n),
                                                     _ptr(got), ctypes.c_int(got.shape[0]), _ptr(out))
    return out[0], out[1]


# ---- acceptance metrics added by the build (SURVEY.md 8(c) (ii),(iii)) -----------------

def residual(A, B, w, Z):
    """||A Z - B Z diag(w)||_F / ||A||_F with A,B Hermitian-completed from their upper parts."""
    A = herm_from_upper(A)
    B = herm_from_upper(B)
    R = A @ Z - (B @ Z) * w[None, : Z.shape[1]]
    return np.linalg.norm(R) / np.linalg.norm(A)


def b_orthonormality(B, Z):
    B = herm_from_upper(B)
    G = Z.conj().T @ B @ Z
    return np.linalg.norm(G - np.eye(Z.shape[1]))


def herm_from_upper(A):
    U = np.triu(A)
    H = U + np.triu(A, 1).conj().T
    if np.iscomplexobj(H):
        H[np.diag_indices(H.shape[0])] = H.diagonal().real
    return H
