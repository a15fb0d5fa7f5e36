R=$GRAFT_REPO_ROOT; cd $R
for k in 0 48 80 112; do
python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline --walk-cus $k 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d=json.loads(l); print('walk-cus $k', round(d['value']/1e9,3), d['stage_ms_per_call_group'])"
done
python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline --no-overlap 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d=json.loads(l); print('no-overlap', round(d['value']/1e9,3), d['stage_ms_per_call_group'])"
