#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_tests
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1
echo "rc=$?" >> $O/gpu_tests.log
tail -30 $O/gpu_tests.log
