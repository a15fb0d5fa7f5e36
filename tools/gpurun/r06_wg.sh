R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_gpu_sage_train.py tests/test_gpu_cross_entropy.py tests/test_gpu_mag_pipeline.py tests/test_gpu_per_batch_step.py -m gpu -q 2>&1 | tail -3
for v in 1 0 1 0; do
WGAMD_WGRAD_TR64=$v python bench.py --workload mag --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d=json.loads(l); print('TR64=$v mag', round(d['value']/1e9,3), {k:(round(v['value']/1e9,3) if v.get('value') else v) for k,v in d['variants'].items()})"
done
