#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_c5prof
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kt5; rocprofv3 --kernel-trace --stats -d /tmp/kt5 -o c5 -- python $R/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-host-tridiag --no-roofline > $O/bench_c5_traced.json 2> $O/bench_c5_traced.err
python $R/tools/rocpd_stats.py $(find /tmp/kt5 -name "*.db" | head -1) $O/r04_c5_kernel_stats.txt > /dev/null
head -30 $O/r04_c5_kernel_stats.txt | cut -c1-190
tail -1 $O/bench_c5_traced.json | cut -c1-400
