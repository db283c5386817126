#!/bin/bash
# Round-4 second GPU run: where the time goes now (phase traces), the tile map's HBM traffic, the real-tile slab depth, the finish kernel.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_run2
mkdir -p $O
cd $R
python tools/trd_finish_bench.py > $O/trd_finish.txt 2>&1
cat $O/trd_finish.txt
# real tiles: slab depth 16/32 (product) vs 32/64 (variant build)
python tools/gemm_shapes.py real > $O/gemm_shapes_real_bk16.txt 2>&1
EIGSOLVE_GPU_LIB=$R/eigensolver_gpu_amd/lib/v_realbk/libeigsolve_gpu.so python tools/gemm_shapes.py real > $O/gemm_shapes_real_bk32.txt 2>&1
paste -d'|' $O/gemm_shapes_real_bk16.txt $O/gemm_shapes_real_bk32.txt | cut -c1-200
python bench.py --real --n 2048 --no-c5 --batch 16 --no-cpu-baseline --no-host-tridiag > $O/bench_c2_bk16.json 2> $O/bench_c2_bk16.err
EIGSOLVE_GPU_LIB=$R/eigensolver_gpu_amd/lib/v_realbk/libeigsolve_gpu.so python bench.py --real --n 2048 --no-c5 --batch 16 --no-cpu-baseline --no-host-tridiag > $O/bench_c2_bk32.json 2> $O/bench_c2_bk32.err
python - <<'PY'
import json
for t in ("bk16","bk32"):
    try:
        d=json.load(open("gpurun_out/r04_run2/bench_c2_%s.json"%t))
        print(t,"value",d["value"],"iso",d["ms_per_solve"],"one_stream",d["isolated_one_stream"]["phase_ms"], "roofline", d["roofline"]["frac"])
    except Exception as e:
        print(t,"failed",e)
PY
cd /tmp; export TMPDIR=/tmp
# phase-segmented traces of one isolated solve on ONE stream
rm -rf /tmp/tr; EIGSOLVE_OVERLAP=0 EIGSOLVE_TRACE_MARKS=1 rocprofv3 --kernel-trace -d /tmp/tr -o c3 -- python $R/tools/solve_trace.py 4096 1024 1 > $O/trace_c3.log 2>&1
python $R/tools/trace_phases.py $(find /tmp/tr -name "*.db" | head -1) --list potrf,gst,bt,trsm $O/r04_phase_trace_c3.txt > /dev/null
rm -rf /tmp/tr2; EIGSOLVE_OVERLAP=0 EIGSOLVE_TRACE_MARKS=1 rocprofv3 --kernel-trace -d /tmp/tr2 -o c2 -- python $R/tools/solve_trace.py 2048 512 1 real > $O/trace_c2.log 2>&1
python $R/tools/trace_phases.py $(find /tmp/tr2 -name "*.db" | head -1) --list potrf,gst,bt,trsm,tridiag $O/r04_phase_trace_c2.txt > /dev/null
grep -v "^       +" $O/r04_phase_trace_c3.txt | head -80
grep -v "^       +" $O/r04_phase_trace_c2.txt | head -80
# tile map: FETCH_SIZE with the map on / off
for mp in 1 0; do
  rm -rf /tmp/pm$mp; EIGSOLVE_TILE_MAP=$mp rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pm$mp -o p -- python $R/tools/pmc_map_targets.py > $O/pmc_map$mp.log 2>&1
  f=$(find /tmp/pm$mp -name "*counter_collection.csv" | head -1)
  cp "$f" $O/pmc_map${mp}_counters.csv 2>/dev/null
done
python - <<'PY'
import csv, collections
for mp in (1, 0):
    try:
        rows = list(csv.DictReader(open("/root/repo/gpurun_out/r04_run2/pmc_map%d_counters.csv" % mp)))
    except Exception as e:
        print("map", mp, "no csv", e); continue
    print("== tile_map =", mp)
    for r in rows:
        if "gemm_fast" in r.get("Kernel_Name", "") and r.get("Counter_Name") == "FETCH_SIZE":
            nm = r["Kernel_Name"].split("gemm_fast_kernel")[1][:40]
            print("  grid %8s  %-40s FETCH_SIZE x2 = %9.1f MB" % (r.get("Grid_Size"), nm, 2 * float(r["Counter_Value"]) / 1024.0))
PY
