#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / avg / min / max.
Usage: python tools/rocpd_stats.py <results.db> [out.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                        "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
lines = ["# rocprofv3 --kernel-trace --stats summary (from %s)" % sys.argv[1], "total kernel time %.3f ms" % tot,
         "%-100s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%")]
for r in rows:
    lines.append("%-100s %8d %12.3f %10.2f %10.2f %10.2f %6.1f" % (r[0][:100], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
# the roofline leg of bench.py: hetrd_mv_sweep replays the mat-vec launches of one tridiagonalization back to back on an idle
# GPU (no panel_row_kernel in between), several times (warm-up + repetitions, and once more for order 8192).  A sweep walks the
# trailing order DOWN, so a new sweep starts where the grid of a panel_mv_kernel launch jumps back up; sweeps are reported
# one by one (launch count, average duration) -- the first order's repetitions are the ones `roofline.avg_launch_us` refers to.
seq = list(cur.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
runs, cur_run = [], []
for nm, st, en, gx, wx in seq + [("", 0, 0, 0, 1)]:
    if "panel_mv_kernel" in nm:
        cur_run.append((st, en, gx // max(wx, 1)))
    else:
        if "panel_row_kernel" in nm or nm == "":
            if len(cur_run) > 100:
                runs.append(cur_run)
            cur_run = []
if runs:
    best = max(runs, key=len)
    sweeps, cur_s = [], []
    prev = None
    for st, en, g in best:
        if prev is not None and g > 1.5 * prev and len(cur_s) > 50:
            sweeps.append(cur_s)
            cur_s = []
        cur_s.append((st, en))
        prev = g
    if cur_s:
        sweeps.append(cur_s)
    lines.append("")
    for k, sw in enumerate(sweeps):
        avg = sum(e - s for s, e in sw) / len(sw) / 1e3
        lines.append("roofline sweep %d (bench.py `roofline` leg): %d consecutive panel_mv_kernel launches, avg %.2f us, total %.3f ms"
                     % (k, len(sw), avg, sum(e - s for s, e in sw) / 1e6))
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
