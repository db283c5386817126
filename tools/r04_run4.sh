#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_run4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "hetrd" > $O/tests_hetrd.log 2>&1; tail -15 $O/tests_hetrd.log
timeout 600 python tools/trd_finish_bench.py > $O/trd_finish.txt 2>&1
cat $O/trd_finish.txt
