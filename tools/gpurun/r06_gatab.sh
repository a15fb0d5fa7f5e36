R=$GRAFT_REPO_ROOT; cd $R
for m in 100000000000 16384 100000000000 16384; do
WGAMD_HEADS_WGRAD_MIN_ROWS=$m python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d=json.loads(l); v=d['variants']; print('min_rows $m', 'gat_train', round(v['gat_train_step']['value']/1e9,3), v['gat_train_step'].get('ms_per_call_group'), 'gat_fwd', round(v['gat_loader_api']['value']/1e9,3), 'train', round(v['train_step']['value']/1e9,3))"
done
