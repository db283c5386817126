cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
timeout 900 python tools/gemm_dma_ab.py check solve 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python tools/gemm_dma_ab.py rate phases 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_gemm_dma_rate.txt; cat gpurun_out/r06_gemm_dma_rate.txt
