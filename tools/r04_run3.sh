#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_run3
mkdir -p $O
cd $R
timeout 300 tools/_build/microbench5 2000 > $O/microbench5.txt 2>&1
cat $O/microbench5.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "hetrd" > $O/tests_hetrd.log 2>&1; tail -3 $O/tests_hetrd.log
python tools/trd_finish_bench.py > $O/trd_finish.txt 2>&1
cat $O/trd_finish.txt
