cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
for z in 0 1 2 3 0 3; do
EIGSOLVE_BATCH_ZIP=$z python bench.py --workload c5 --steps 3 --no-cpu-baseline --no-host-tridiag --no-roofline > $O/bench_c5_zip$z.json 2> $O/bench_c5_zip$z.err
EIGSOLVE_BATCH_ZIP=$z python bench.py --real --n 2048 --no-c5 --batch 16 --no-cpu-baseline --no-host-tridiag --no-roofline > $O/bench_c2_zip$z.json 2> $O/bench_c2_zip$z.err
python - <<PY
import json
for f in ("gpurun_out/r06b/bench_c5_zip$z.json","gpurun_out/r06b/bench_c2_zip$z.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print("zip=$z", d["metric"], round(d["value"],2), round(d["ms_per_step"],2))
PY
done
