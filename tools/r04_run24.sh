#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python tools/large_order.py 16384 4096 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_large_order_16384.txt
timeout 600 python tools/large_order.py 16384 2048 real 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_large_order_16384.txt
