#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_run5
mkdir -p $O
cd $R
EIGSOLVE_GPU_LIB=$R/eigensolver_gpu_amd/lib/v_timing/libeigsolve_gpu.so timeout 300 python tools/multi_finish_timing.py 2>&1 | grep -v amdgpu.ids | tee $O/multi_timing.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "hetrd" > $O/tests_hetrd.log 2>&1; tail -4 $O/tests_hetrd.log
timeout 600 python tools/trd_finish_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/trd_finish.txt
