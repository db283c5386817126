#!/usr/bin/env python3
"""Which hardware queue did each of the library's streams end up on?  Reads a rocprofv3 --kernel-trace rocpd database and
prints, per (queue, stream, thread) triple, the number of kernel dispatches and the busy time.
Usage: python tools/queue_map.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("columns:", cols)
qcol = [c for c in cols if "queue" in c.lower()]
scol = [c for c in cols if "stream" in c.lower()]
tcol = [c for c in cols if c.lower() in ("tid", "thread_id")]
sel = ", ".join(qcol + scol + tcol)
if not sel:
    sys.exit("no queue / stream columns in this trace")
for row in db.execute("select %s, count(*), sum(end-start)/1e6 from kernels group by %s order by 1, 2" % (sel, sel)):
    print(row)
