#!/usr/bin/env python3
"""Segments the LAST solve of a rocprofv3 kernel trace (rocpd sqlite, produced with tools/solve_trace.py under
EIGSOLVE_TRACE_MARKS=1) into the solver's phases and prints for every phase: span, sum of kernel time, idle gaps and the
per-kernel-name breakdown; optionally the time-ordered launch list of chosen phases.

The library launches an empty `phase_marker_kernel` with grid = 1 + 2*phase (begin) / 2 + 2*phase (end) at every phase
boundary (evd.hip), phases numbered as in eigsolve_get_phase_times: 0 potrf, 1 gst, 2 trd, 3 tridiagonal solver,
4 back-transform, 5 trsm, 6 D2H.  Everything between a begin marker and its end marker belongs to that phase -- no guessing
from kernel names.  The sections must add up to the traced solve (checked, printed in the last line).
Usage: python tools/trace_phases.py <results.db> [--list potrf,gst,bt] [out.txt]"""
import sqlite3
import sys

PHASES = ["potrf", "gst", "trd", "tridiag", "bt", "trsm", "d2h"]
args = sys.argv[1:]
listed = set()
if "--list" in args:
    i = args.index("--list")
    listed = set(args[i + 1].split(","))
    del args[i:i + 2]
db = sqlite3.connect(args[0])
rows = list(db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"))


def short(nm):
    nm = nm.replace("eig::(anonymous namespace)::", "").replace("eig::", "").replace("void ", "")
    for a, b in (("gemm_fast_kernel", "gemmF"), ("gemm_kernel", "gemmG"), ("cplx", "z"), ("double", "d"), (", ", ",")):
        nm = nm.replace(a, b)
    return nm.split("(")[0][:60]


def marker_id(r):
    return r[3] // max(r[6], 1) - 1 if "phase_marker_kernel" in r[0] else None


marks = [(i, marker_id(r)) for i, r in enumerate(rows) if marker_id(r) is not None]
if not marks:
    sys.exit("no phase_marker_kernel launches in the trace: run the solve with EIGSOLVE_TRACE_MARKS=1")
start = max(i for i, m in marks if m == 0)          # potrf begin of the last solve
sol = rows[start:]
out = []
cur, seg_start, total_busy, total_span = None, None, 0.0, 0.0
segments = []
for i, r in enumerate(sol):
    m = marker_id(r)
    if m is None:
        continue
    if m % 2 == 0:
        cur, seg_start = m // 2, i + 1
    elif cur == m // 2:
        segments.append((PHASES[cur], sol[seg_start:i]))
        cur = None
t_all0 = sol[0][2]
t_all1 = max(r[2] for r in sol)
for nm, seg in segments:
    seg = [r for r in seg if marker_id(r) is None]
    if not seg:
        continue
    span = (seg[-1][2] - seg[0][1]) / 1e3
    busy = sum(r[2] - r[1] for r in seg) / 1e3
    total_busy += busy
    total_span += span
    out.append("== %-10s launches %5d  span %10.1f us  kernel time %10.1f us  gaps %9.1f us" % (nm, len(seg), span, busy, span - busy))
    agg = {}
    for r in seg:
        k = short(r[0])
        c = agg.setdefault(k, [0, 0.0])
        c[0] += 1
        c[1] += (r[2] - r[1]) / 1e3
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
        out.append("     %-62s %6d  %10.1f us  avg %8.2f" % (k, c, t, t / c))
    if nm in listed:
        prev_end = seg[0][1]
        for r in seg:
            out.append("       +%9.1f gap %6.1f dur %8.1f  grid %5d,%4d,%3d  %s" % ((r[1] - t_all0) / 1e3, (r[1] - prev_end) / 1e3,
                       (r[2] - r[1]) / 1e3, r[3] // max(r[6], 1), r[4], r[5], short(r[0])))
            prev_end = r[2]
out.append("-- sections: span %.1f us, kernel time %.1f us; traced solve (first marker .. last kernel end) %.1f us; "
           "outside the sections: %.1f us (host gaps between phases: potrf info check, phase boundaries)"
           % (total_span, total_busy, (t_all1 - t_all0) / 1e3, (t_all1 - t_all0) / 1e3 - total_span))
txt = "\n".join(out)
print(txt)
if len(args) > 1:
    open(args[1], "w").write(txt + "\n")
