#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "optional_execution_modes or c4_ or full_spectrum" 2>&1 | tail -3
for i in 1 2; do
python tools/iso_phases.py 8192 8192 cplx 2 2>&1 | grep -v amdgpu.ids
EIGSOLVE_OVERLAP=1 python tools/iso_phases.py 8192 8192 cplx 2 2>&1 | grep -v amdgpu.ids
done
python tools/iso_phases.py 4096 4096 cplx 3 2>&1 | grep -v amdgpu.ids
EIGSOLVE_OVERLAP=1 python tools/iso_phases.py 4096 4096 cplx 3 2>&1 | grep -v amdgpu.ids
