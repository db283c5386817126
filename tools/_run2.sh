cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
L=$PWD/eigensolver_gpu_amd/lib
timeout 600 python tools/mv_dma_ab.py check 2>&1 | grep -v "True  rel\|True  finite" | tail -5
for m in 0 1; do EIGSOLVE_MV_DMA=$m EIGSOLVE_GPU_LIB=$L/v_timing/libeigsolve_gpu.so timeout 600 python tools/trd_phase_timing.py 2>&1 | grep -v "amdgpu.ids\|row"; done > gpurun_out/r06_mv_stamps3.txt
cat gpurun_out/r06_mv_stamps3.txt
timeout 600 python tools/mv_dma_ab.py rate 2>&1 | grep "hemv" > gpurun_out/r06_mv_rate2.txt; cat gpurun_out/r06_mv_rate2.txt
timeout 900 python tools/mv_dma_ab.py trd 2>&1 | grep "N=" > gpurun_out/r06_mv_trd2.txt; cat gpurun_out/r06_mv_trd2.txt
