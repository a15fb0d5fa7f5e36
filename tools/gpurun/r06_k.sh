R=$GRAFT_REPO_ROOT; cd $R
for k in 4 2 1 4 2; do
WGAMD_SAMPLE_K=$k python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d=json.loads(l); print('K=$k', round(d['value']/1e9,3), d['stage_ms_per_call_group'])"
done
