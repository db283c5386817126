#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for v in "" res256 res384 res512; do
  if [ -n "$v" ]; then export EIGSOLVE_GPU_LIB=$R/eigensolver_gpu_amd/lib/$v/libeigsolve_gpu.so; else unset EIGSOLVE_GPU_LIB; fi
  echo "== variant '${v:-default}'"
  python tools/hemv_curve.py 4096 2>&1 | grep -v amdgpu.ids | tail -9
  python tools/iso_phases.py 4096 1024 cplx 3 2>&1 | grep -v amdgpu.ids
done
