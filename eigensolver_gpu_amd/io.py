"""Matrix-file input of the reference's test driver.

test_driver/test_dsygvdx.F90:120-145 reads two Fortran *unformatted sequential* files, one per matrix:

    record 1:  n, m, lda          (three default integers)
    record 2:  A(1:n, 1:n)        (real(8), column-major; complex(8) for the z driver's analogue)

Each record is framed by a 4-byte little-endian length marker before and after (gfortran / PGI /
flang default).  `m` is the number of wanted eigenpairs.  No sample files ship with the reference;
`write_matrix_file` produces the same format so real Quantum ESPRESSO matrices dumped with
`write(unit) n, m, lda; write(unit) A(1:n,1:n)` can be replayed through this library.
"""
import struct

import numpy as np


def write_matrix_file(path, A, m, lda=None):
    A = np.asfortranarray(A)
    n = A.shape[0]
    lda = n if lda is None else lda
    with open(path, "wb") as f:
        hdr = struct.pack("<iii", n, m, lda)
        f.write(struct.pack("<i", len(hdr)) + hdr + struct.pack("<i", len(hdr)))
        body = A[:n, :n].tobytes(order="F")
        f.write(struct.pack("<i", len(body)) + body + struct.pack("<i", len(body)))


def read_matrix_file(path, dtype=None):
    """Returns (A[n,n] Fortran-ordered, n, m, lda).  dtype None -> inferred from the record size."""
    with open(path, "rb") as f:
        raw = f.read()
    off = 0

    def record():
        nonlocal off
        (ln,) = struct.unpack_from("<i", raw, off)
        data = raw[off + 4: off + 4 + ln]
        (ln2,) = struct.unpack_from("<i", raw, off + 4 + ln)
        if ln != ln2:
            raise ValueError("corrupt Fortran record markers in %s" % path)
        off += 8 + ln
        return data

    n, m, lda = struct.unpack("<iii", record())
    body = record()
    if dtype is None:
        dtype = np.complex128 if len(body) == 16 * n * n else np.float64
    A = np.frombuffer(body, dtype=dtype).reshape((n, n), order="F").copy(order="F")
    return A, n, m, lda


def compare_report(ref, got, kind="1d"):
    """Report line in the format of compare() (test_driver/toolbox.F90:70-74): l2 relative error and max
    percent error of |entries| (2-D variants compare absolute values, toolbox.F90:101-103)."""
    ref = np.asarray(ref)
    got = np.asarray(got)
    a, b = (np.abs(ref), np.abs(got)) if kind != "1d" else (ref.astype(float), got.astype(float))
    mask = np.abs(ref) >= 1e-10
    if not mask.any():
        return "     EXACT MATCH"
    l2 = np.sqrt(np.sum((a[mask] - b[mask]) ** 2))
    nrm = np.sqrt(np.sum(a[mask] ** 2))
    if l2 == 0.0:
        return "     EXACT MATCH"
    perr = np.where(mask, np.abs(a - b) / np.where(mask, np.abs(ref), 1.0) * 100.0, 0.0)
    idx = np.unravel_index(np.argmax(perr), perr.shape)
    return "%16s  %10.3E%12s%10.3E%6s%s%6s  %20.14E  %6s  %20.14E" % (
        "l2norm error", l2 / nrm, "max error", perr[idx], "% at", "".join("%5d" % (i + 1) for i in idx), "cpu=",
        a[idx], "gpu=", b[idx])
