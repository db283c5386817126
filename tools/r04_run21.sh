#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "hetrd_vs_oracle or hemv_vs_oracle" 2>&1 | tail -2
for i in 1 2; do
for v in "" gemvfirst; do
  if [ -n "$v" ]; then export EIGSOLVE_GPU_LIB=$R/eigensolver_gpu_amd/lib/$v/libeigsolve_gpu.so; else unset EIGSOLVE_GPU_LIB; fi
  echo "== variant '${v:-default(hemv first)}'"
  python tools/iso_phases.py 4096 1024 cplx 3 2>&1 | grep -v amdgpu.ids
  python tools/iso_phases.py 2048 512 real 5 2>&1 | grep -v amdgpu.ids
  python tools/iso_phases.py 2048 512 cplx 5 2>&1 | grep -v amdgpu.ids
done
done
