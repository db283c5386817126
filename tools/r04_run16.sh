#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_run16
mkdir -p $O
cd $R
B="python bench.py --no-cpu-baseline --no-host-tridiag --no-c5 --no-roofline --isolated-reps 1 --steps 6"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - "$O/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("%-28s value %.3f  ms/step %.1f" % (sys.argv[2], d["value"], d["ms_per_step"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
run default $B
EIGSOLVE_BATCH_FUSE=2 run fuse2 $B
EIGSOLVE_BATCH_FUSE=2 EIGSOLVE_BATCH_WORKERS=3 run fuse2_w3 $B
EIGSOLVE_BATCH_FUSE=2 EIGSOLVE_BATCH_WORKERS=2 run fuse2_w2 $B
EIGSOLVE_BATCH_WORKERS=3 run w3 $B
EIGSOLVE_BATCH_WORKERS=5 run w5 $B
run default_b16 $B --batch 16
