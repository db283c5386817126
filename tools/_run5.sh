cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "staging or dma_ring or either_data_path or gemm_vs_numpy or her2k" 2>&1 | tail -5
python bench.py --no-cpu-baseline --no-host-tridiag > $O/bench_c3_dma3.json 2> $O/bench_c3_dma3.err
EIGSOLVE_GEMM_DMA=0 python bench.py --no-cpu-baseline --no-host-tridiag > $O/bench_c3_dma0.json 2> $O/bench_c3_dma0.err
python bench.py --n 8192 --m 8192 --batch 1 --steps 2 --warmup 1 --no-c5 --no-cpu-baseline --no-host-tridiag --same-problems > $O/bench_c4_dma3.json 2> $O/bench_c4.err
EIGSOLVE_GEMM_DMA=0 python bench.py --n 8192 --m 8192 --batch 1 --steps 2 --warmup 1 --no-c5 --no-cpu-baseline --no-host-tridiag --same-problems > $O/bench_c4_dma0.json 2>> $O/bench_c4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06a/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unparsed", e); continue
    print(f, d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), json.dumps(d.get("isolated_one_stream",{}))[:400], json.dumps(d.get("c5",{}))[:200])
PY
