#!/bin/bash
# One parametrised A/B runner for option / environment sweeps on the GPU box (replaces the per-run scripts of round 4).
#   tools/ab_sweep.sh <tag> <leg> [reps] -- "ENV1=a ENV2=b" "ENV1=c" ...
# legs:  iso  [n m real|cplx]   median per-phase ms of isolated solves (tools/iso_phases.py); default 4096 1024 cplx
#        c3                     bench.py default line (batch rate; no roofline / c5 / cpu legs)
#        c5                     bench.py --workload c5
#        c2                     bench.py --real --n 2048 --batch 16
#        c4                     bench.py --n 8192 --m 8192 --batch 1 (one isolated full-spectrum solve per step)
#        potrf [n real|cplx]    tools/potrf_bench.py
#        final                  full -m gpu suite, 200-case stress run, smoke()
# Every setting is a quoted string of VAR=value pairs ("" = defaults) applied to the leg's command; one result line per
# (setting, repetition) goes to stdout and to gpurun_out/<tag>/sweep.txt.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; leg=$2; shift 2
args=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
[ $# -gt 0 ] && shift
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
B="--no-cpu-baseline --no-host-tridiag --no-roofline"
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    ph = d.get("isolated_one_stream", {}).get("phase_ms", {})
    print("%-44s value %8.3f %s  ms/step %8.1f  isolated %7.2f ms  one-stream %s" % (
        sys.argv[2] or "(defaults)", d["value"], d["unit"], d["ms_per_step"], d.get("ms_per_solve", 0.0),
        " ".join("%s %.2f" % (k, ph[k]) for k in ("potrf", "gst", "trd", "backtransform", "trsm") if k in ph)))
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
}
run_one() {
    setting=$1; rep=$2
    name=$(echo "${setting:-defaults}" | tr ' =/' '__-')_r$rep
    case $leg in
        iso)   env $setting python tools/iso_phases.py "${args[@]:-}" 2>&1 | grep -v amdgpu.ids ;;
        potrf) env $setting python tools/potrf_bench.py "${args[@]:-}" 2>&1 | grep -v amdgpu.ids ;;
        c3)    env $setting python bench.py $B --no-c5 --isolated-reps 1 --steps 6 > $O/$name.json 2> $O/$name.err; line $O/$name.json "$setting" ;;
        c5)    env $setting python bench.py $B --workload c5 --steps 3 > $O/$name.json 2> $O/$name.err; line $O/$name.json "$setting" ;;
        c2)    env $setting python bench.py $B --real --n 2048 --no-c5 --batch 16 > $O/$name.json 2> $O/$name.err; line $O/$name.json "$setting" ;;
        c4)    env $setting python bench.py $B --n 8192 --m 8192 --batch 1 --steps 2 --warmup 1 --no-c5 --same-problems > $O/$name.json 2> $O/$name.err; line $O/$name.json "$setting" ;;
    esac
}
if [ "$leg" = final ]; then
    timeout 3000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
    timeout 1500 python tools/stress.py 200 17 3000 > $O/stress_200_cases.txt 2>&1; tail -1 $O/stress_200_cases.txt
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
    exit 0
fi
reps=1
if [[ "${args[0]:-}" =~ ^[0-9]+$ ]] && [ "$leg" != iso ] && [ "$leg" != potrf ]; then reps=${args[0]}; fi
[ $# -eq 0 ] && set -- ""
for setting in "$@"; do
    for rep in $(seq 1 $reps); do run_one "$setting" $rep; done
done | tee $O/sweep.txt
