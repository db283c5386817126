cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
timeout 900 python tools/gemm_dma_ab.py check solve 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_gemm_dma_check.txt; tail -4 gpurun_out/r06_gemm_dma_check.txt
timeout 900 python tools/gemm_dma_ab.py rate phases 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_gemm_dma_rate.txt; cat gpurun_out/r06_gemm_dma_rate.txt
timeout 600 python tools/mv_dma_ab.py rate 2>&1 | grep "hemv" > gpurun_out/r06_mv_rate.txt; cat gpurun_out/r06_mv_rate.txt
timeout 900 python tools/mv_dma_ab.py trd 2>&1 | grep "N=" > gpurun_out/r06_mv_trd.txt; cat gpurun_out/r06_mv_trd.txt
