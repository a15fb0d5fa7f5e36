set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_walk_gw; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_gpu_callgroup.py tests/test_gpu_pyg_loader.py tests/test_gpu_call_group_loader.py tests/test_gpu_sampling.py tests/test_gpu_mag_pipeline.py -x -q 2>&1 | tail -3 > $OUT/tests.log
cat $OUT/tests.log
cd /tmp
for X in 0 1; do
  echo "EXACT=$X" >> $OUT/summary.txt
  WGAMD_SAMPLE_EXACT_WIDTH=$X python $R/tools/profile_walk.py 2>&1 | tail -1 >> $OUT/summary.txt
done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wp_x -o wp -- python $R/tools/profile_walk.py > $OUT/trace.log 2>&1
cp /tmp/wp_x/wp_kernel_stats.csv $OUT/wp_kernel_stats.csv
cat $OUT/summary.txt
python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_plain.log 2>&1; grep "^{\"metric" $OUT/bench_plain.log | tail -1 > $OUT/bench_n1.json
python - <<PY
import json
d=json.load(open("$OUT/bench_n1.json")); print(d["value"], d["ms_per_step"], d.get("stage_ms_per_call_group")); print({k:v.get("value") for k,v in d["variants"].items()})
PY
