#!/bin/bash
# Round-6 evidence run (on the GPU box): kernel-trace summaries of the driver's bench command (C3) and of C2, phase-segmented traces
# of one isolated solve of each, the counter passes, the un-traced bench lines, the Cholesky A/B, the streaming micro-benchmark and
# the two-stage launch skeleton.  Everything lands in gpurun_out/r06_ev/; the summaries are copied to profiles/ by hand.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_ev
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
# (1) the driver's command under --kernel-trace --stats
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o c3 -- python $R/bench.py --no-cpu-baseline --no-host-tridiag > $O/bench_c3_traced.json 2> $O/bench_c3_traced.err
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) $O/r06_c3_kernel_stats.txt > /dev/null
# (2) C2: dsygvdx N=2048 m=512
rm -rf /tmp/kt2; rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o c2 -- python $R/bench.py --real --n 2048 --no-c5 --no-cpu-baseline --no-host-tridiag > $O/bench_c2_traced.json 2> $O/bench_c2_traced.err
python $R/tools/rocpd_stats.py $(find /tmp/kt2 -name "*.db" | head -1) $O/r06_c2_kernel_stats.txt > /dev/null
# (3) phase-segmented traces of one isolated solve on ONE stream and in the default form
rm -rf /tmp/tr; EIGSOLVE_OVERLAP=0 EIGSOLVE_TRACE_MARKS=1 rocprofv3 --kernel-trace -d /tmp/tr -o c3 -- python $R/tools/solve_trace.py 4096 1024 1 > $O/trace_c3.log 2>&1
python $R/tools/trace_phases.py $(find /tmp/tr -name "*.db" | head -1) --list potrf,gst,bt,trsm $O/r06_phase_trace_c3.txt > /dev/null
rm -rf /tmp/tr3; EIGSOLVE_TRACE_MARKS=1 rocprofv3 --kernel-trace -d /tmp/tr3 -o c3p -- python $R/tools/solve_trace.py 4096 1024 1 > $O/trace_c3_pipelined.log 2>&1
python $R/tools/trace_phases.py $(find /tmp/tr3 -name "*.db" | head -1) $O/r06_phase_trace_c3_pipelined.txt > /dev/null
rm -rf /tmp/tr2; EIGSOLVE_OVERLAP=0 EIGSOLVE_TRACE_MARKS=1 rocprofv3 --kernel-trace -d /tmp/tr2 -o c2 -- python $R/tools/solve_trace.py 2048 512 1 real > $O/trace_c2.log 2>&1
python $R/tools/trace_phases.py $(find /tmp/tr2 -name "*.db" | head -1) --list potrf $O/r06_phase_trace_c2.txt > /dev/null
# (4) counters (bounded: counter collection serialises every dispatch)
cd $R; timeout 900 bash tools/pmc_collect.sh gpurun_out/r06_pmc $O/r06_pmc_summary.txt $O/r06_hemv_traffic.json
# (5) un-traced bench lines: the default (C3), C2, C5 as the timed workload, C4
python bench.py > $O/r06_bench_c3.json 2> $O/r06_bench_c3.err
python bench.py --real --n 2048 --no-c5 --batch 16 > $O/r06_bench_c2_dsygvdx_n2048.json 2> $O/r06_bench_c2.err
python bench.py --workload c5 --steps 3 --no-cpu-baseline --no-host-tridiag > $O/r06_bench_c5_1gpu.json 2> $O/r06_bench_c5.err
python bench.py --n 8192 --m 8192 --batch 1 --steps 2 --warmup 1 --no-c5 --no-cpu-baseline --no-host-tridiag --same-problems > $O/r06_bench_c4_n8192_full.json 2> $O/r06_bench_c4.err
# (6) round-6 kernels on their own shapes: whole-CU workgroups for the 32 x 32 tiles, lean LDS-DMA form, staging paths
python tools/small_gemm_shapes.py 2>&1 | grep -v amdgpu.ids > $O/r06_small_gemm_shapes.txt
python tools/gemm_dma_ab.py lean rate 2>&1 | grep -v amdgpu.ids > $O/r06_gemm_lean_ab.txt
ls -la $O
