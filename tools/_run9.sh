cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06c; mkdir -p $O
timeout 900 python tools/mv_dma_ab.py trd 2>&1 | grep "N=4096\|z N=8192" > $O/snake_dma_trd.txt
cat $O/snake_dma_trd.txt
python bench.py --n 8192 --m 8192 --batch 1 --steps 2 --warmup 1 --no-c5 --no-cpu-baseline --no-host-tridiag --same-problems > $O/bench_c4_snake.json 2> $O/bench_c4.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06c/bench_c4_snake.json").read().strip().splitlines()[-1])
print("C4", d["value"], d["ms_per_step"], json.dumps(d.get("isolated_one_stream",{}).get("phase_ms",{})), d["config"]["workload"][-200:])
PY
