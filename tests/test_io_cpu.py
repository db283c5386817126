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


def _golden_cases():
    import importlib.util
    import json
    import os
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_compare_golden", os.path.join(gd, "make_compare_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    with open(os.path.join(gd, "compare_ref.json")) as f:
        return gen, json.load(f)["cases"]


def test_compare_report_lines_match_the_reference_binary():
    """SURVEY.md 8(f) row 3, the report-format half: io.compare_report reproduces, character for character, the lines
    the reference's own compare() (compiled from /root/reference/test_driver/toolbox.F90, oracle/_ref) printed for the
    seeded cases of tests/golden/compare_ref.json -- all three shapes, incl. EXACT MATCH and skipped tiny entries."""
    gen, cases = _golden_cases()
    assert len(cases) >= 10
    for c in cases:
        ref, got = gen.case_arrays(c["kind"], c["n"], c["m"], c["seed"], c["noise"], c["zero_frac"])
        line = eio.compare_report(ref, got).strip()
        if c.get("l2", 1.0) < 1e-15:
            # a pure phase rotation: what is left is the last-bit difference between numpy's and Fortran's complex abs()
            assert line.startswith("l2norm error") and float(line.split()[2]) < 1e-15, (c, line)
            continue
        assert line == c["report"].strip(), c


def test_fortran_compare_utils_matches_the_reference_binary(tmp_path):
    """The Fortran counterpart (eigensolver_gpu_amd/fortran/compare_utils.F90, what the Fortran test drivers print)
    against the same reference-printed lines, through the seeded-vector driver linked with OUR module."""
    import os
    import struct
    import subprocess
    import pytest
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "eigensolver_gpu_amd", "fortran", "compare_driver")
    if not os.path.exists(exe):
        pytest.skip("Fortran compare_driver not built (no amdflang)")
    gen, cases = _golden_cases()
    for c in cases:
        ref, got = gen.case_arrays(c["kind"], c["n"], c["m"], c["seed"], c["noise"], c["zero_frac"])
        p = tmp_path / "case.bin"
        n = ref.shape[0]
        m = 1 if c["kind"] == 1 else ref.shape[1]
        p.write_bytes(struct.pack("<iii", c["kind"], n, m) + ref.tobytes(order="F") + got.tobytes(order="F"))
        out = subprocess.run([exe, str(p)], capture_output=True, text=True, check=True).stdout.strip()
        assert out == c["report"].strip(), (c, out)
