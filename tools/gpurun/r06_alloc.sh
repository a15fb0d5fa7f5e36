R=$GRAFT_REPO_ROOT; cd $R
for conf in "" "expandable_segments:True" "" "expandable_segments:True"; do
PYTORCH_HIP_ALLOC_CONF=$conf PYTORCH_CUDA_ALLOC_CONF=$conf python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d=json.loads(l); v=d['variants']; print('conf=[$conf]', round(d['value']/1e9,3), 'train', round(v['train_step']['value']/1e9,3), 'gat_train', round(v['gat_train_step']['value']/1e9,3), 'pb', round(v['train_step_per_batch']['value']/1e9,3), 'loader', round(v['loader_api']['value']/1e9,3))"
done
