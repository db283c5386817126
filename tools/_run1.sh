set -x
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
timeout 600 python tools/mv_dma_ab.py check > gpurun_out/r06_mv_check.txt 2>&1; tail -5 gpurun_out/r06_mv_check.txt
timeout 600 python tools/mv_dma_ab.py rate > gpurun_out/r06_mv_rate.txt 2>&1
timeout 900 python tools/mv_dma_ab.py trd > gpurun_out/r06_mv_trd.txt 2>&1
cat gpurun_out/r06_mv_rate.txt gpurun_out/r06_mv_trd.txt
