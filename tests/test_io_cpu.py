"""Reference test-driver file format (test_driver/test_dsygvdx.F90:120-145) round trip + compare() report."""
import numpy as np

import oracle
from eigensolver_gpu_amd import io as eio


def test_unformatted_round_trip(tmp_path):
    for cplx in (False, True):
        A = oracle.gen_spd(37, 5, cplx)
        p = tmp_path / ("a%d.bin" % cplx)
        eio.write_matrix_file(str(p), A, m=9)
        B, n, m, lda = eio.read_matrix_file(str(p))
        assert (n, m, lda) == (37, 9, 37)
        assert B.dtype == A.dtype and np.array_equal(A, B)
        raw = p.read_bytes()
        assert raw[:4] == (12).to_bytes(4, "little")          # first record: three default integers


def test_compare_report_matches_oracle_numbers():
    rng = np.random.default_rng(0)
    ref = rng.standard_normal(50)
    got = ref * (1 + 1e-9 * rng.standard_normal(50))
    line = eio.compare_report(ref, got)
    l2, mx = oracle.compare_1d(ref, got)
    assert "l2norm error" in line and ("%10.3E" % l2) in line and ("%10.3E" % mx) in line
    assert eio.compare_report(ref, ref).strip() == "EXACT MATCH"
