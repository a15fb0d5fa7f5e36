# bench.py (driver command, no variants / CPU baseline) with different option sets on one box: each argument is one option string
R=$GRAFT_REPO_ROOT; TAG=${1:-benchargs}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; shift
for rep in 1 2; do for a in "$@"; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants $a 2>/dev/null | grep '^{"metric' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('[$a] value %.4g G  ms_per_step %.3f' % (d['value'] / 1e9, d['ms_per_step']))
"
done; done | tee $OUT/bench_args.log
