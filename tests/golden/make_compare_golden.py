"""Generates tests/golden/compare_ref.json: outputs of the REFERENCE's own acceptance metric (module compare_utils,
/root/reference/test_driver/toolbox.F90:36-176, compiled as oracle/_ref/ref_compare by oracle/Makefile) on seeded
vectors.  tests/test_oracle.py checks oracle.compare_1d / compare_abs2d against these printed values.

This pins only the acceptance metric of the oracle against the reference -- the solver stages of the reference
(CUDA Fortran) cannot run here.  Run in the build container:  python tests/golden/make_compare_golden.py
"""
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_compare")


def case_arrays(kind, n, m, seed, noise, zero_frac):
    """The seeded inputs of one case (the test regenerates them from these parameters)."""
    rng = np.random.default_rng(seed)
    shape = (n,) if kind == 1 else (n, m)
    ref = rng.standard_normal(shape)
    got = ref * (1.0 + noise * rng.standard_normal(shape))
    if kind == 3:
        ref = ref + 1j * rng.standard_normal(shape)
        # eigenvectors are defined up to a phase: compare() works on |entries| (toolbox.F90:101-103)
        got = (ref * np.exp(1j * rng.uniform(0, 2 * np.pi, size=(1, m)))) * (1.0 + noise * rng.standard_normal(shape))
    if zero_frac:
        mask = rng.uniform(size=shape) < zero_frac
        ref = np.where(mask, ref * 1e-13, ref)     # entries below the 1e-10 cut-off are skipped (toolbox.F90:55)
    if kind == 2 and seed % 2:
        got = -got                                  # sign flips of real eigenvectors are invisible to the 2-D compare
    return np.asfortranarray(ref), np.asfortranarray(got)


CASES = [  # kind, n, m, seed, noise, zero_frac
    (1, 64, 1, 11, 1e-9, 0.0), (1, 1000, 1, 12, 1e-13, 0.1), (1, 7, 1, 13, 0.0, 0.0), (1, 257, 1, 14, 1e-3, 0.3),
    (2, 33, 9, 21, 1e-8, 0.0), (2, 200, 50, 22, 1e-12, 0.2), (2, 5, 5, 23, 0.0, 0.0),
    (3, 33, 9, 31, 1e-8, 0.0), (3, 129, 40, 32, 1e-11, 0.2), (3, 6, 2, 33, 0.0, 0.0),
]


def run_reference(kind, ref, got):
    n = ref.shape[0]
    m = 1 if kind == 1 else ref.shape[1]
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(struct.pack("<iii", kind, n, m))
        f.write(ref.tobytes(order="F"))
        f.write(got.tobytes(order="F"))
        path = f.name
    try:
        out = subprocess.run([EXE, path], capture_output=True, text=True, check=True).stdout
    finally:
        os.unlink(path)
    return out.strip()


def main():
    if not os.path.exists(EXE):
        sys.exit("build oracle/_ref first: make -C oracle")
    rows = []
    for kind, n, m, seed, noise, zf in CASES:
        ref, got = case_arrays(kind, n, m, seed, noise, zf)
        line = run_reference(kind, ref, got)
        row = {"kind": kind, "n": n, "m": m, "seed": seed, "noise": noise, "zero_frac": zf, "report": line}
        if "EXACT MATCH" not in line:
            tok = line.split()
            # "l2norm error  x.xxxE-xx   max error x.xxxE+xx  % at  i [j]  cpu= ..."
            row["l2"] = float(tok[2])
            row["maxerr_percent"] = float(tok[5])
        rows.append(row)
        print(row)
    with open(os.path.join(ROOT, "tests", "golden", "compare_ref.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_compare_golden.py", "source": "test_driver/toolbox.F90 compare()",
                   "cases": rows}, f, indent=1)


if __name__ == "__main__":
    main()
