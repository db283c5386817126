"""CPU tests: the C oracle against the LAPACK golden fixtures (tests/golden/*.npz) and
against scipy's LAPACK live.  This is how the oracle is pinned -- the reference's own
tests hold no golden vectors and compare against CPU LAPACK ?hegvd in the same way
(test_driver/test_zhegvdx.F90:172-179,298-299)."""
import glob
import os

import numpy as np
import pytest

import oracle

EPS = np.finfo(np.float64).eps
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "[dz]*.npz")))


def load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    return {k: g[k] for k in g.files}


def test_fixtures_present():
    assert len(CASES) >= 6


@pytest.mark.parametrize("name", CASES)
def test_potrf_vs_lapack(golden_dir, name):
    g = load(golden_dir, name)
    U, info = oracle.potrf_upper(g["B"])
    assert info == 0
    assert np.abs(np.triu(U) - g["U"]).max() <= 1e-10 * np.abs(g["U"]).max()


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("nb", [448, 17])
def test_hegst_vs_lapack(golden_dir, name, nb):
    g = load(golden_dir, name)
    C = oracle.hegst(g["A"], g["U"], nb=nb)
    scale = np.abs(g["C"]).max()
    assert np.abs(np.triu(C) - g["C"]).max() <= 1e-9 * scale
    # strict lower triangle outside the diagonal blocks must be untouched (zero in fixture input)
    n = C.shape[0]
    for k in range(0, n, nb):
        assert np.all(C[k + nb:, k:k + nb] == 0)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("nb", [32, 8])
def test_hetrd_vs_lapack(golden_dir, name, nb):
    """Blocked reference-style trd gives LAPACK's (d,e,tau) to rounding (same conventions)."""
    g = load(golden_dir, name)
    Ao, d, e, tau = oracle.hetrd(g["C"], nb=nb)
    scale = np.abs(g["C"]).max()
    tol = 200 * g["C"].shape[0] * EPS * scale
    assert np.abs(d - g["d"]).max() <= tol
    assert np.abs(e - g["e"]).max() <= tol
    assert np.abs(tau - g["tau"]).max() <= 1e-9
    n = Ao.shape[0]
    # blocked part keeps the explicit 1 on the superdiagonal (zhetrd_gpu.F90:92)
    for j in range(33, n):
        assert Ao[j - 1, j] == 1.0


@pytest.mark.parametrize("name", CASES)
def test_hetrd_reconstructs(golden_dir, name):
    """Q^H C Q = T with Q from the stored reflectors (checks the V/tau layout)."""
    g = load(golden_dir, name)
    C = oracle.herm_from_upper(g["C"])
    n = C.shape[0]
    Ao, d, e, tau = oracle.hetrd(g["C"], nb=32)
    Q = np.eye(n, dtype=C.dtype)
    for j in range(n - 1):  # reflector j: v = [A(0:j, j+1); 1], length j+1
        v = np.zeros(n, dtype=C.dtype)
        v[:j] = Ao[:j, j + 1]
        v[j] = 1.0
        H = np.eye(n, dtype=C.dtype) - tau[j] * np.outer(v, v.conj())
        Q = H @ Q
    # Q = H_{n-2} ... H_0 (LAPACK ?hetrd 'U'),  T = Q^H C Q
    Tm = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    assert np.abs(Q.conj().T @ C @ Q - Tm).max() <= 1e-11 * np.abs(C).max()


@pytest.mark.parametrize("n", [1, 2, 5, 31, 32])
@pytest.mark.parametrize("cplx", [False, True])
def test_hetd2_small(n, cplx):
    from scipy.linalg import lapack
    A = oracle.gen_spd(n, 77 + n, cplx)
    Ao, d, e, tau = oracle.hetd2(A)
    _, dr, er, taur, info = (lapack.zhetrd if cplx else lapack.dsytrd)(A, lower=0)
    s = max(np.abs(A).max(), 1.0)
    assert np.abs(d - dr).max() <= 1e-12 * s
    if n > 1:
        assert np.abs(e - er).max() <= 1e-12 * s
        assert np.abs(tau - taur).max() <= 1e-10


@pytest.mark.parametrize("n", [1, 2, 17, 64, 150])
def test_steql_vs_lapack(n):
    from scipy.linalg import eigh_tridiagonal
    rng = np.random.default_rng(n)
    d = rng.standard_normal(n)
    e = rng.standard_normal(max(n - 1, 0))
    w, Q, info = oracle.steql(d, e)
    assert info == 0
    if n == 1:
        assert w[0] == d[0]
        return
    wr, Qr = eigh_tridiagonal(d, e)
    assert np.abs(w - wr).max() <= 1e-12 * max(1.0, np.abs(wr).max())
    assert np.abs(np.abs(Q) - np.abs(Qr)).max() <= 1e-8
    assert np.abs(Q.T @ Q - np.eye(n)).max() <= 1e-12


@pytest.mark.parametrize("name", CASES)
def test_hegvdx_vs_lapack(golden_dir, name):
    """End-to-end: the reference's own acceptance (compare() vs LAPACK ?hegvd) plus the
    residual and B-orthonormality gates of SURVEY.md 8(c)."""
    g = load(golden_dir, name)
    n = g["A"].shape[0]
    m = max(1, n // 4)
    w, Z, Ao, Bo, info = oracle.hegvdx(g["A"], g["B"], 1, m)
    assert info == 0
    l2w, _ = oracle.compare_1d(g["w"], w)
    l2z, _ = oracle.compare_abs2d(g["Zabs"][:, :m].astype(Z.dtype), Z)
    wc = name.endswith("wc")
    assert l2w <= (1e-13 if wc else 1e-8)
    assert l2z <= (1e-10 if wc else 1e-5)
    res = oracle.residual(g["A"], g["B"], w, Z)
    assert res <= n * EPS
    assert oracle.b_orthonormality(g["B"], Z) <= (1e-12 if wc else 1e-9)
    # contract: B <- U, strict lower(A) preserved (zero in the fixture inputs)
    assert np.abs(np.triu(Bo) - g["U"]).max() <= 1e-10 * np.abs(g["U"]).max()
    assert np.all(np.tril(Ao, -1) == 0)


@pytest.mark.parametrize("cplx", [False, True])
def test_hegvdx_il_gt_1(cplx):
    n = 40
    A = oracle.gen_spd(n, 5, cplx)
    B = oracle.gen_spd(n, 6, cplx, shift=n)
    w, Z, *_ , info = oracle.hegvdx(A, B, 5, 12)
    assert info == 0 and Z.shape[1] == 8
    # eigenvalues: all N ascending (zheevd_gpu.F90:111); vectors il..iu
    R = oracle.herm_from_upper(A) @ Z - (oracle.herm_from_upper(B) @ Z) * w[None, 4:12]
    assert np.linalg.norm(R) / np.linalg.norm(A) <= n * EPS


@pytest.mark.parametrize("cplx", [False, True])
def test_not_positive_definite(cplx):
    n = 12
    A = oracle.gen_spd(n, 5, cplx)
    B = oracle.gen_spd(n, 6, cplx)
    B[3, 3] = -1.0
    *_, info = oracle.hegvdx(A, B, 1, 4)
    assert info == -1


def test_generator_matches_fast_path():
    for cplx in (False, True):
        a = oracle.gen_spd(50, 123, cplx, shift=2.0)
        b = oracle.gen_spd_fast(50, 123, cplx, shift=2.0)
        assert np.abs(a - b).max() <= 1e-13 * np.abs(a).max()
    assert 0.0 <= oracle.u01(1, 2, 3, 0) < 1.0


# ---------------------------------------------------------------------------------------------
# The acceptance metric pinned against the REFERENCE's own compare() (module compare_utils,
# test_driver/toolbox.F90:36-176): outputs of the reference code itself, compiled from where it lies
# (oracle/Makefile -> oracle/_ref/ref_compare) and recorded by tests/golden/make_compare_golden.py.
# This is the one piece of the reference that runs in this image; it pins the metric, not the solver stages.
# ---------------------------------------------------------------------------------------------
def _compare_cases(golden_dir):
    import json
    with open(os.path.join(golden_dir, "compare_ref.json")) as f:
        return json.load(f)["cases"]


def _load_compare_generator(golden_dir):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_compare_golden", os.path.join(golden_dir, "make_compare_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _oracle_compare(kind, ref, got):
    return oracle.compare_1d(ref, got) if kind == 1 else oracle.compare_abs2d(ref, got)


def test_compare_metric_vs_reference_outputs(golden_dir):
    gen = _load_compare_generator(golden_dir)
    cases = _compare_cases(golden_dir)
    assert len(cases) >= 10
    for c in cases:
        ref, got = gen.case_arrays(c["kind"], c["n"], c["m"], c["seed"], c["noise"], c["zero_frac"])
        l2, mx = _oracle_compare(c["kind"], ref, got)
        if "EXACT MATCH" in c["report"]:
            assert l2 == 0.0, c
        else:
            # the reference prints ES10.3: four significant digits
            assert abs(l2 - c["l2"]) <= 6e-4 * c["l2"], (c, l2)
            assert abs(mx - c["maxerr_percent"]) <= 6e-4 * c["maxerr_percent"], (c, mx)


def test_compare_metric_vs_reference_binary_live(golden_dir):
    """Same check against the reference binary itself where it was built (oracle/_ref travels to the GPU box)."""
    gen = _load_compare_generator(golden_dir)
    if not os.path.exists(gen.EXE):
        pytest.skip("oracle/_ref/ref_compare not built (needs /root/reference + amdflang at build time)")
    rng = np.random.default_rng(99)
    for kind, n, m in ((1, 300, 1), (2, 40, 11), (3, 40, 11)):
        ref, got = gen.case_arrays(kind, n, m, int(rng.integers(1 << 30)), 1e-7, 0.1)
        line = gen.run_reference(kind, ref, got)
        tok = line.split()
        l2, mx = _oracle_compare(kind, ref, got)
        assert abs(l2 - float(tok[2])) <= 6e-4 * l2 and abs(mx - float(tok[5])) <= 6e-4 * mx, line
