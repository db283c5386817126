#!/bin/bash
# Round-4 final validation on the GPU box: full -m gpu suite, the 200-case stress run, smoke().
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_final
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log | head -3
timeout 1500 python tools/stress.py 200 17 3000 > $O/r04_stress_200_cases.txt 2>&1; tail -1 $O/r04_stress_200_cases.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
cp gpurun_out/r04_parity_full_size.json $O/ 2>/dev/null
