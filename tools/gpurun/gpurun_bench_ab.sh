# bench.py (driver command, no variants / CPU baseline) A/B over WGAMD_SAGE_WS on one box
R=$GRAFT_REPO_ROOT; TAG=${1:-benchab}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for ws in 0 1 0 1; do
  WGAMD_SAGE_WS=$ws timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants 2>/dev/null | grep '^{"metric' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('WS=$ws value %.4g G  ms_per_step %.3f  stages %s' % (d['value'] / 1e9, d['ms_per_step'], json.dumps(d.get('stage_ms_per_call_group'))))
"
done | tee $OUT/bench_ab.log
