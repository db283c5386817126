#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; timeout 1500 python -m pytest tests/test_gpu_parity.py -q -k "bench_eight or real_path_il or batch_driver_error or liwork or fortran_real or contexts_die or finalize" 2>&1 | tail -3
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
V=$R/eigensolver_gpu_amd/lib/v_nt/libeigsolve_gpu.so
echo "== default loads"; python tools/hemv_curve.py 8192 2>&1 | grep -v amdgpu | tail -8
echo "== nt loads"; EIGSOLVE_GPU_LIB=$V python tools/hemv_curve.py 8192 2>&1 | grep -v amdgpu | tail -8
for i in 1 2; do
python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
EIGSOLVE_GPU_LIB=$V python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
done
B="python bench.py --no-cpu-baseline --no-host-tridiag --no-c5 --no-roofline --isolated-reps 1 --steps 6"
$B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch default', d['value'])"
EIGSOLVE_GPU_LIB=$V $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch nt', d['value'])"
