# bench.py (driver command, no variants / CPU baseline) A/B over library builds in cugraph-gnn_amd/lib/variants on one box
R=$GRAFT_REPO_ROOT; TAG=${1:-benchlibs}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
cp cugraph-gnn_amd/lib/libwholegraph_amd.so /tmp/keep.so
for rep in 1 2; do for v in $2; do
  cp cugraph-gnn_amd/lib/variants/libwholegraph_amd.$v.so cugraph-gnn_amd/lib/libwholegraph_amd.so
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants ${BENCH_ARGS} 2>/dev/null | grep '^{"metric' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v value %.4g G  ms_per_step %.3f  stages %s' % (d['value'] / 1e9, d['ms_per_step'], json.dumps(d.get('stage_ms_per_call_group'))))
"
done; done | tee $OUT/bench_libs.log
cp /tmp/keep.so cugraph-gnn_amd/lib/libwholegraph_amd.so
