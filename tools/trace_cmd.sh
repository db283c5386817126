#!/bin/bash
# Kernel trace of any command on the GPU box, summarised per kernel (tools/rocpd_stats.py) + optional launch list.
# Usage: tools/trace_cmd.sh <tag> <launch-list-lines> -- <command...>     -> gpurun_out/<tag>_stats.txt, gpurun_out/<tag>_list.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; nlist=$2; shift 3
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tc_$tag
rocprofv3 --kernel-trace -d /tmp/tc_$tag -o t -- "$@" > $R/gpurun_out/${tag}_cmd.log 2>&1
db=$(find /tmp/tc_$tag -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db $R/gpurun_out/${tag}_stats.txt > /dev/null
if [ "$nlist" -gt 0 ]; then
python - "$db" "$nlist" > $R/gpurun_out/${tag}_list.txt <<PY
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2])
rows = list(db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"))
rows = rows[-n:]
t0 = rows[0][1]; prev = None
for nm, st, en, gx, gy, gz, wx in rows:
    gap = 0.0 if prev is None else (st - prev) / 1e3
    print("+%10.1f gap %6.1f dur %8.1f grid %6d,%4d,%3d  %s" % ((st - t0) / 1e3, gap, (en - st) / 1e3, gx // max(wx, 1), gy, gz, nm[:90]))
    prev = en
PY
fi
head -25 $R/gpurun_out/${tag}_stats.txt | cut -c1-200
