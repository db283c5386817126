cd $GRAFT_REPO_ROOT
L=$PWD/eigensolver_gpu_amd/lib
for m in 0 1; do EIGSOLVE_MV_DMA=$m EIGSOLVE_GPU_LIB=$L/v_timing/libeigsolve_gpu.so timeout 600 python tools/trd_phase_timing.py 2>&1 | grep -v "amdgpu.ids"; done > gpurun_out/r06_mv_stamps.txt
cat gpurun_out/r06_mv_stamps.txt
