cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06d; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "potrf or overlap or optional_execution or pipelined or leading_dim or full_spectrum or c4_" > $O/pytest_la.txt 2>&1; grep -n "passed\|failed" $O/pytest_la.txt
{
for s in "EIGSOLVE_OVERLAP=0" "EIGSOLVE_OVERLAP=4" "EIGSOLVE_OVERLAP=3" "EIGSOLVE_OVERLAP=7" "EIGSOLVE_OVERLAP=0" "EIGSOLVE_OVERLAP=4" "EIGSOLVE_OVERLAP=3" "EIGSOLVE_OVERLAP=7"; do
  env $s python tools/iso_phases.py 4096 1024 cplx 5 2>&1 | grep -v amdgpu.ids
done
for s in "EIGSOLVE_OVERLAP=0" "EIGSOLVE_OVERLAP=4" "EIGSOLVE_OVERLAP=3" "EIGSOLVE_OVERLAP=7"; do
  env $s python tools/iso_phases.py 8192 8192 cplx 2 2>&1 | grep -v amdgpu.ids
  env $s python tools/iso_phases.py 2048 512 real 5 2>&1 | grep -v amdgpu.ids
done
} > $O/lookahead_sweep.txt
cat $O/lookahead_sweep.txt
