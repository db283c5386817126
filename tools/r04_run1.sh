#!/bin/bash
# Round-4 first GPU run: parity of the rewritten engine / solves / finish kernel, then shapes and the C3 line.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_run1
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "gemm or her2k or potrf or trsm or hegst or hetrd or stages or hegvdx_vs_oracle or triangular or larft or finalize or error_paths or batch_driver" > $O/tests_subset.log 2>&1
echo "subset rc=$?" >> $O/tests_subset.log
tail -25 $O/tests_subset.log
python tools/gemm_shapes.py > $O/gemm_shapes_map1.txt 2>&1
EIGSOLVE_TILE_MAP=0 python tools/gemm_shapes.py > $O/gemm_shapes_map0.txt 2>&1
python bench.py --no-cpu-baseline --no-host-tridiag --steps 5 > $O/bench_c3.json 2> $O/bench_c3.err
tail -3 $O/bench_c3.err
python - <<'PY'
import json,sys
try:
    d=json.load(open("gpurun_out/r04_run1/bench_c3.json"))
    print("value",d["value"],"iso",d["ms_per_solve"],"phases",d["phase_ms_single_solve"])
    print("one_stream",d["isolated_one_stream"])
    print("roofline",d.get("roofline",{}).get("frac"),d.get("roofline",{}).get("avg_launch_us"))
    print("mfma",{k:(v if not isinstance(v,dict) else v.get("frac")) for k,v in d.get("roofline_mfma",{}).items()})
    print("resid",d["residual"],d["strict_gate"])
    print("c5",d.get("c5",{}).get("value"))
except Exception as e:
    print("no bench json",e)
PY
paste -d'|' $O/gemm_shapes_map1.txt $O/gemm_shapes_map0.txt | cut -c1-230
