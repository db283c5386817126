#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
timeout 600 python tools/large_order.py 16384 4096
timeout 600 python tools/large_order.py 16384 2048 real
timeout 900 python tools/large_order.py 24576 2048
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_large_order.txt
