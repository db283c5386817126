#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for i in 1 2; do
for v in "" zov; do
  if [ -n "$v" ]; then export EIGSOLVE_GPU_LIB=$R/eigensolver_gpu_amd/lib/$v/libeigsolve_gpu.so; else unset EIGSOLVE_GPU_LIB; fi
  echo "== variant '${v:-default}'"
  python tools/iso_phases.py 4096 1024 cplx 5 2>&1 | grep -v amdgpu.ids
  python tools/iso_phases.py 2048 512 real 7 2>&1 | grep -v amdgpu.ids
  python tools/iso_phases.py 2048 512 cplx 7 2>&1 | grep -v amdgpu.ids
  python tools/iso_phases.py 1024 256 cplx 7 2>&1 | grep -v amdgpu.ids
done
done
