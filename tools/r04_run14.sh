#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for i in 1 2; do
python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
EIGSOLVE_GPU_LIB=$R/eigensolver_gpu_amd/lib/v_chol6/libeigsolve_gpu.so python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
done
EIGSOLVE_OVERLAP=0 python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
EIGSOLVE_OVERLAP=0 EIGSOLVE_GPU_LIB=$R/eigensolver_gpu_amd/lib/v_chol6/libeigsolve_gpu.so python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
