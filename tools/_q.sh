timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "stedc or heevd or tridiag or golden or c1_ or batch" 2>&1 | grep -a -E "passed|failed"
for r in 1 2; do timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-tridiag 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); p=j['phase_ms_single_solve']; print(round(j['value'],2), {k:round(v,2) for k,v in p.items()})
"; done
