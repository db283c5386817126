#!/bin/bash
# round 6, session 5: whole-CU workgroups for the 32 x 32 tiles (option gemm_wide), lean LDS-DMA form (option gemm_lean)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/w2; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "whole_cu or lean_dma or gemm_vs_numpy or her2k_vs or potrf_and_trsm or trsm_inverse or hegst_every or larft_and or c5_full_size or batch_driver_bit or optional_execution or stages_vs_golden or hegvdx_vs_oracle" 2>&1 | tail -6) > $O/pytest.log
tail -3 $O/pytest.log
echo "== 4x4x4 form (product)"; python tools/small_gemm_shapes.py 2>&1 | grep -v amdgpu.ids | tee $O/small_gemm_shapes.txt
echo "== 16x16x4 form (variant)"; EIGSOLVE_GPU_LIB=$R/eigensolver_gpu_amd/lib/v_w16/libeigsolve_gpu.so python tools/small_gemm_shapes.py 2>&1 | grep -v amdgpu.ids | tee $O/small_gemm_shapes_w16.txt
python tools/gemm_dma_ab.py lean 2>&1 | grep -v amdgpu.ids | tee $O/lean_ab.txt
bash tools/ab_sweep.sh w2 iso 4096 1024 cplx 5 -- "EIGSOLVE_OVERLAP=0 EIGSOLVE_GEMM_WIDE=0" "EIGSOLVE_OVERLAP=0 EIGSOLVE_GEMM_WIDE=1" "EIGSOLVE_OVERLAP=0 EIGSOLVE_GEMM_WIDE=2" "EIGSOLVE_OVERLAP=0 EIGSOLVE_GEMM_WIDE=2 EIGSOLVE_GEMM_LEAN=64" "EIGSOLVE_OVERLAP=0 EIGSOLVE_GEMM_WIDE=2 EIGSOLVE_GEMM_LEAN=128" "EIGSOLVE_OVERLAP=0 EIGSOLVE_GEMM_WIDE=2 EIGSOLVE_GEMM_LEAN=256" "EIGSOLVE_GEMM_WIDE=0" "EIGSOLVE_GEMM_WIDE=2" "EIGSOLVE_GEMM_WIDE=2 EIGSOLVE_GEMM_LEAN=128"
bash tools/ab_sweep.sh w2 iso 2048 512 cplx 5 -- "EIGSOLVE_OVERLAP=0 EIGSOLVE_GEMM_WIDE=0" "EIGSOLVE_OVERLAP=0 EIGSOLVE_GEMM_WIDE=1" "EIGSOLVE_OVERLAP=0 EIGSOLVE_GEMM_WIDE=2" "EIGSOLVE_OVERLAP=0 EIGSOLVE_GEMM_WIDE=2 EIGSOLVE_GEMM_LEAN=128"
bash tools/ab_sweep.sh w2c3 c3 -- "EIGSOLVE_GEMM_WIDE=0" "EIGSOLVE_GEMM_WIDE=2" "EIGSOLVE_GEMM_WIDE=2 EIGSOLVE_GEMM_LEAN=128" "EIGSOLVE_GEMM_WIDE=0"
bash tools/ab_sweep.sh w2c5 c5 -- "EIGSOLVE_GEMM_WIDE=0" "EIGSOLVE_GEMM_WIDE=2" "EIGSOLVE_GEMM_WIDE=2 EIGSOLVE_GEMM_LEAN=128" "EIGSOLVE_GEMM_WIDE=1"
