#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_run6
mkdir -p $O
cd $R
B="python bench.py --no-cpu-baseline --no-host-tridiag --no-roofline --no-c5 --isolated-reps 1 --steps 5"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - "$O/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("%-28s value %.3f  ms/step %.1f  iso %.2f" % (sys.argv[2], d["value"], d["ms_per_step"], d["ms_per_solve"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
run batch4 $B --batch 4
run batch8 $B --batch 8
run batch12 $B --batch 12
EIGSOLVE_HEMV_BLOCKS=256 run hemv256_b4 $B --batch 4
EIGSOLVE_HEMV_BLOCKS=384 run hemv384_b4 $B --batch 4
EIGSOLVE_HEMV_BLOCKS=768 run hemv768_b4 $B --batch 4
EIGSOLVE_TILE_MAP=0 run map0_b4 $B --batch 4
EIGSOLVE_TRD_FINISH=32 run fin32_b4 $B --batch 4
EIGSOLVE_BT_NB=128 run btnb128_b4 $B --batch 4
run batch4_again $B --batch 4
