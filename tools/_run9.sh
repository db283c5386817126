cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
L=$PWD/eigensolver_gpu_amd/lib
O=gpurun_out/r06c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hemv or hetrd or dma_ring or either_data_path or batch_driver_bit" > $O/pytest_snake.txt 2>&1; grep -n "passed\|failed" $O/pytest_snake.txt
for v in snake nosnake snake nosnake; do
  if [ $v = nosnake ]; then export EIGSOLVE_GPU_LIB=$L/v_nosnake/libeigsolve_gpu.so; else unset EIGSOLVE_GPU_LIB; fi
  echo "== $v"; timeout 900 python tools/mv_dma_ab.py trd 2>&1 | grep "mv_dma=   0"
done > $O/snake_trd.txt
cat $O/snake_trd.txt
unset EIGSOLVE_GPU_LIB
python bench.py --no-cpu-baseline --no-host-tridiag > $O/bench_c3_snake.json 2> $O/bench_c3_snake.err
EIGSOLVE_GPU_LIB=$L/v_nosnake/libeigsolve_gpu.so python bench.py --no-cpu-baseline --no-host-tridiag > $O/bench_c3_nosnake.json 2> $O/bench_c3_nosnake.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06c/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["value"],3), round(d["ms_per_step"],1), "roof", round(d["roofline"]["frac"],4), d["roofline"]["avg_launch_us"], "n8192", d["roofline"].get("sweep_n8192",{}).get("frac"), "iso", json.dumps(d.get("isolated_one_stream",{}).get("phase_ms",{})), "c5", d.get("c5",{}).get("value"))
PY
