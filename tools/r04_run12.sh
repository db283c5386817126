#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_run12
mkdir -p $O
cd $R
for nb in 16 24 32 40 64; do EIGSOLVE_TRD_NB=$nb python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids; done
B="python bench.py --no-cpu-baseline --no-host-tridiag --no-c5 --no-roofline --isolated-reps 1 --steps 6"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - "$O/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("%-28s value %.3f  ms/step %.1f" % (sys.argv[2], d["value"], d["ms_per_step"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
for rep in 1 2; do
for nb in 24 32 48 64; do
EIGSOLVE_TRD_NB=$nb run nb${nb}_r$rep $B
done
done
echo "--- C5 / C2 batch"
for nb in 32 64; do
EIGSOLVE_TRD_NB=$nb run c5_nb${nb} python bench.py --workload c5 --steps 3 --no-cpu-baseline --no-host-tridiag --no-roofline
EIGSOLVE_TRD_NB=$nb run c2_nb${nb} python bench.py --real --n 2048 --no-c5 --batch 16 --no-cpu-baseline --no-host-tridiag --no-roofline
done
