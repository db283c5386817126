#!/usr/bin/env python3
"""Segments the LAST solve of a rocprofv3 kernel trace (rocpd sqlite, produced with tools/solve_trace.py) into the
solver's phases by the kernels that delimit them, and prints for every phase: span, sum of kernel time, idle gaps, and the
per-kernel-name breakdown; optionally the time-ordered launch list of chosen phases.
Usage: python tools/trace_phases.py <results.db> [--list potrf,gst,bt] [out.txt]"""
import sqlite3
import sys

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


# the last solve starts at the last memset-free run beginning with a diag_block_kernel after an hetd2/trsm of the previous one
idx_diag = [i for i, r in enumerate(rows) if "diag_block_kernel" in r[0] or "chol_row_kernel" in r[0]]
idx_td2 = [i for i, r in enumerate(rows) if "hetd2_kernel" in r[0]]
# the first diag kernel of the last solve: all its diag kernels come after the previous solve's hetd2
prev_td2 = idx_td2[-2] if len(idx_td2) > 1 else -1
first_diag = min(i for i in idx_diag if i > prev_td2)
sol = rows[first_diag:]
names = [r[0] for r in sol]


def first(pred, start=0):
    for i in range(start, len(sol)):
        if pred(names[i]):
            return i
    return len(sol)


i_row = first(lambda s: "panel_row_kernel" in s)
i_merge_end = max([i for i in range(i_row) if "tri_merge_kernel" in names[i]] + [0]) + 1
i_td2 = first(lambda s: "hetd2_kernel" in s)
i_bt0 = first(lambda s: "widen_kernel" in s, i_td2)
i_fin = first(lambda s: "finish_T_kernel" in s, i_td2)
bt_start = min(x for x in (first(lambda s: "gemm" in s, i_bt0),) if x)
# back-transform ends with its last "C -= V Wk2^H" gemm; the final trsm follows: find the first copyBufferRect (256-base copy_back) after bt
i_end = len(sol)
phases = [("potrf", 0, i_merge_end), ("gst", i_merge_end, i_row), ("trd", i_row, i_td2 + 2), ("tridiag(D&C)", i_td2 + 2, i_bt0 + 1),
          ("bt+trsm", i_bt0 + 1, i_end)]
out = []
t_all0 = sol[0][1]
for nm, a, b in phases:
    seg = sol[a:b]
    if not seg:
        continue
    span = (seg[-1][2] - seg[0][1]) / 1e3
    busy = sum(r[2] - r[1] for r in seg) / 1e3
    out.append("== %-14s launches %5d  span %10.1f us  kernel time %10.1f us  gaps %9.1f us" % (nm, len(seg), span, busy, span - busy))
    agg = {}
    for r in seg:
        k = short(r[0])
        c = agg.setdefault(k, [0, 0.0])
        c[0] += 1
        c[1] += (r[2] - r[1]) / 1e3
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        out.append("     %-62s %6d  %10.1f us  avg %8.2f" % (k, c, t, t / c))
    if nm.split("(")[0].split("+")[0] in listed or nm in listed:
        prev_end = seg[0][1]
        for r in seg:
            out.append("       +%9.1f gap %6.1f dur %8.1f  grid %5d,%4d,%2d  %s" % ((r[1] - t_all0) / 1e3, (r[1] - prev_end) / 1e3,
                       (r[2] - r[1]) / 1e3, r[3] // max(r[6], 1), r[4], r[5], short(r[0])))
            prev_end = r[2]
txt = "\n".join(out)
print(txt)
if len(args) > 1:
    open(args[1], "w").write(txt + "\n")
