for tf in 0 2048 1024; do echo "EIGSOLVE_TRD_FUSE=$tf"; EIGSOLVE_TRD_FUSE=$tf timeout 300 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-host-tridiag 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('c5', round(j['value'],2))
"; EIGSOLVE_TRD_FUSE=$tf timeout 300 python bench.py --real --n 2048 --no-c5 --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-host-tridiag 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('c2 batch 16', round(j['value'],2))
"; done
