#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
EIGSOLVE_GPU_LIB=$R/eigensolver_gpu_amd/lib/v_mv2/libeigsolve_gpu.so python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
EIGSOLVE_GPU_LIB=$R/eigensolver_gpu_amd/lib/v_mv2/libeigsolve_gpu.so python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
python tools/iso_phases.py 2048 512 real 2>&1 | grep -v amdgpu.ids
EIGSOLVE_GPU_LIB=$R/eigensolver_gpu_amd/lib/v_mv2/libeigsolve_gpu.so python tools/iso_phases.py 2048 512 real 2>&1 | grep -v amdgpu.ids
