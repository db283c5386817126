#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_run9
mkdir -p $O
cd $R
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - "$O/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("%-28s value %.3f  ms/step %.1f  iso %.2f  trd %.2f  sweep frac %s  c5 %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["ms_per_solve"], d["phase_ms_single_solve"]["trd"], d.get("roofline",{}).get("frac"), d.get("c5",{}).get("value")))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
run c3_default python bench.py --no-cpu-baseline --no-host-tridiag
run c2 python bench.py --real --n 2048 --no-c5 --batch 16 --no-cpu-baseline --no-host-tridiag
run c5 python bench.py --workload c5 --steps 3 --no-cpu-baseline --no-host-tridiag --no-roofline
EIGSOLVE_HEMV_BLOCKS=512 run c5_hemv512 python bench.py --workload c5 --steps 3 --no-cpu-baseline --no-host-tridiag --no-roofline
EIGSOLVE_HEMV_BLOCKS=512 run c2_hemv512 python bench.py --real --n 2048 --no-c5 --batch 16 --no-cpu-baseline --no-host-tridiag --no-roofline
