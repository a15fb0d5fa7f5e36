R=$GRAFT_REPO_ROOT; cd $R
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d=json.loads(l); print(round(d['value']/1e9,3), {k:(round(v['value']/1e9,3) if v.get('value') else v) for k,v in d['variants'].items()}); print(d['variants']['train_step'].get('loss_first_last'), d['variants']['train_step'].get('forward_loss_ms'), d['variants']['train_step'].get('backward_step_ms'))"
python bench.py --workload mag --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d=json.loads(l); print('mag', round(d['value']/1e9,3), {k:(round(v['value']/1e9,3) if v.get('value') else v) for k,v in d['variants'].items()})"
