for r in 1 2 3; do timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-tridiag 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); p=j['phase_ms_single_solve']; print(round(j['value'],2), {k:round(v,2) for k,v in p.items()})
"; done
