#!/bin/bash
# round 6, session 5: lean LDS-DMA form with the tile-count rule -- repeated A/B (shapes, isolated phases, batch rates)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/w3; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "whole_cu or lean_dma or optional_execution or staging_paths" 2>&1 | tail -4) > $O/pytest.log
tail -2 $O/pytest.log
for i in 1 2 3; do python tools/gemm_dma_ab.py lean 2>&1 | grep -v amdgpu.ids; done | tee $O/lean_ab.txt
bash tools/ab_sweep.sh w3 iso 4096 1024 cplx 7 -- "EIGSOLVE_OVERLAP=0" "EIGSOLVE_OVERLAP=0 EIGSOLVE_GEMM_LEAN=64" "EIGSOLVE_OVERLAP=0 EIGSOLVE_GEMM_LEAN=128" "EIGSOLVE_OVERLAP=0" "EIGSOLVE_OVERLAP=0 EIGSOLVE_GEMM_LEAN=64" "EIGSOLVE_OVERLAP=0 EIGSOLVE_GEMM_LEAN=128"
bash tools/ab_sweep.sh w3c3 c3 -- "" "EIGSOLVE_GEMM_LEAN=128" "" "EIGSOLVE_GEMM_LEAN=128" "EIGSOLVE_GEMM_LEAN=64" ""
bash tools/ab_sweep.sh w3c4 c4 -- "" "EIGSOLVE_GEMM_LEAN=128" "" "EIGSOLVE_GEMM_LEAN=128"
