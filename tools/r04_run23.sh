#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -k "hetrd or hemv or batch or bit_identical or golden or c2_ or c3_" > gpurun_out/r04_run23_pytest.txt 2>&1; tail -2 gpurun_out/r04_run23_pytest.txt | cut -c1-200
