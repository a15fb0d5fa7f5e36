R=$GRAFT_REPO_ROOT; TAG=${1:-mag}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_gat_transform.py tests/test_gpu_mag_pipeline.py -q -x -n 4 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "^\.*s*\.* *\[" $OUT/pytest.log | tail -25
for mode in split fused split fused; do
  WGAMD_GAT_LAYER=$mode timeout 600 python bench.py --workload mag --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$mode value %.4g G  ms_per_step %.3f  stages %s' % (d['value'] / 1e9, d['ms_per_step'], json.dumps(d.get('stage_ms_per_call_group'))))
"
done | tee $OUT/mag_ab.log
