#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_run17
mkdir -p $O
cd $R
V=$R/eigensolver_gpu_amd/lib/v_nopersist/libeigsolve_gpu.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "gemm or her2k or triangular or potrf_and_trsm or hegst" 2>&1 | tail -3
python tools/gemm_shapes.py 2>&1 | grep -v amdgpu > $O/shapes_persist.txt
EIGSOLVE_GPU_LIB=$V python tools/gemm_shapes.py 2>&1 | grep -v amdgpu > $O/shapes_nopersist.txt
paste -d'|' $O/shapes_persist.txt $O/shapes_nopersist.txt | cut -c1-64,100-164
for i in 1 2; do
EIGSOLVE_OVERLAP=0 python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
EIGSOLVE_OVERLAP=0 EIGSOLVE_GPU_LIB=$V python tools/iso_phases.py 2>&1 | grep -v amdgpu.ids
done
