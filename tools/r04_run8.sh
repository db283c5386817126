#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_run8
mkdir -p $O
cd $R
B="python bench.py --no-cpu-baseline --no-host-tridiag --no-c5 --isolated-reps 3 --steps 5"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - "$O/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("%-28s value %.3f  ms/step %.1f  iso %.2f  trd %.2f  sweep frac %.4f" % (sys.argv[2], d["value"], d["ms_per_step"], d["ms_per_solve"], d["phase_ms_single_solve"]["trd"], d["roofline"]["frac"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
for hb in 96 128 160 224 240 272 288; do
EIGSOLVE_HEMV_BLOCKS=$hb run hemv${hb}_b8 $B --batch 8
done
