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
# the roofline leg of bench.py: the longest run of consecutive panel_mv_kernel launches with no panel_row_kernel in
# between (hetrd_mv_sweep replays the mat-vec launches of one tridiagonalization back to back on an idle GPU)
seq = list(cur.execute("select name, start, end from kernels order by start"))
best = (0, 0, 0); run_start = None; run_len = 0
for idx, (nm, st, en) in enumerate(seq + [("", 0, 0)]):
    if "panel_mv_kernel" in nm:
        if run_len == 0: run_start = idx
        run_len += 1
    elif "panel_row_kernel" in nm or nm == "":
        if run_len > best[0]: best = (run_len, run_start, idx)
        run_len = 0
if best[0] > 100:
    sw = [r for r in seq[best[1]:best[2]] if "panel_mv_kernel" in r[0]]
    avg = sum(e - s for _, s, e in sw) / len(sw) / 1e3
    lines.append("")
    lines.append("roofline sweep (bench.py `roofline` leg): %d consecutive panel_mv_kernel launches, avg %.2f us, total %.3f ms"
                 % (len(sw), avg, sum(e - s for _, s, e in sw) / 1e6))
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
