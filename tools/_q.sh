export GPU_MAX_HW_QUEUES=8
timeout 250 python tools/corun.py 2>&1 | grep -v amdgpu.ids
for r in 1 2; do timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-tridiag 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); p=j['phase_ms_single_solve']; print(round(j['value'],2), {k:round(v,2) for k,v in p.items()}, round(j['roofline']['frac'],4))
"; done
