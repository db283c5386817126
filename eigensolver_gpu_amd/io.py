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


def _fortran_e(x, width=20, digits=14):
    """Fortran E<width>.<digits> edit descriptor as amdflang / gfortran print it: mantissa in [0.1, 1), two-digit exponent,
    the leading zero dropped when it does not fit (negative values at E20.14)."""
    x = float(x)
    if x == 0.0:
        mant, exp = 0.0, 0
    else:
        exp = int(np.floor(np.log10(abs(x)))) + 1
        mant = abs(x) / 10.0 ** exp
        if round(mant, digits) >= 1.0:
            mant /= 10.0
            exp += 1
    body = ("%.*f" % (digits, mant))[1:]            # ".dddd"
    tail = "E%+03d" % exp
    sign = "-" if (x < 0 or (x == 0.0 and np.signbit(x))) else ""
    txt = sign + "0" + body + tail
    if len(txt) > width:
        txt = sign + body + tail
    return txt.rjust(width)


def _fortran_es(x):
    """ES10.3"""
    return ("%10.3E" % float(x))


def compare_report(ref, got, kind=None):
    """The report line of the reference test driver's compare() (test_driver/toolbox.F90:70-74, :119-123, :168-172) for a
    reference (CPU LAPACK) and a computed array: relative l2 error and the largest percent error with its position and
    the two values there.  1-D arrays are compared as they are, 2-D arrays through the magnitudes of their entries
    (eigenvectors are defined up to a sign / phase); entries with |ref| < 1e-10 are skipped; identical inputs give
    EXACT MATCH.  Character for character what the reference prints (pinned in tests/test_io_cpu.py against lines printed
    by the reference's own routine, tests/golden/compare_ref.json), including its quirk of printing the two values of the
    real 2-D case through single precision (REAL(x), toolbox.F90:120).  `kind` is accepted for backward compatibility."""
    ref = np.asarray(ref)
    got = np.asarray(got)
    two_d = ref.ndim == 2
    cx = np.iscomplexobj(ref) or np.iscomplexobj(got)
    rmag = np.abs(ref)
    a, b = (rmag, np.abs(got)) if two_d else (ref.astype(float), got.astype(float))
    use = rmag >= 1e-10
    l2 = np.sqrt(np.sum(((a - b) ** 2)[use]))
    if l2 == 0.0:
        return "     EXACT MATCH"
    l2 /= np.sqrt(np.sum((rmag ** 2)[use]))
    perr = np.where(use & (ref != 0) & (got != 0), np.abs(a - b) / np.where(use, rmag, 1.0) * 100.0, 0.0)
    # first maximum in column-major order (the reference scans j outer, i inner, with a strict >)
    flat = np.argmax(perr.ravel(order="F"))
    idx = np.unravel_index(flat, perr.shape, order="F")
    if perr[idx] <= 0.0:
        idx = tuple(0 for _ in perr.shape)
    head = "%16s  %s%12s%s%6s" % ("l2norm error", _fortran_es(l2), "max error", _fortran_es(perr[idx]), "% at")
    pos = "".join("%5d" % (i + 1) for i in idx)
    r, g = ref[idx], got[idx]
    if not two_d:
        return "%s%s%6s  %s  %6s  %s" % (head, pos, "cpu=", _fortran_e(r), "gpu=", _fortran_e(g))
    if not cx:
        r4, g4 = np.float32(r), np.float32(g)
        return "%s%s%6s  %s   %6s  %s " % (head, pos, "cpu=", _fortran_e(r4), "gpu=", _fortran_e(g4))
    return "%s%s%6s  %s %s  %6s  %s %s" % (head, pos, "cpu=", _fortran_e(r.real), _fortran_e(r.imag), "gpu=",
                                           _fortran_e(g.real), _fortran_e(g.imag))
