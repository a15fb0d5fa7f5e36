set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_pb; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_per_batch_step.py -x -q 2>&1 | tail -30 > $OUT/tests.log
cat $OUT/tests.log
timeout 900 python -m pytest tests/test_gpu_callgroup.py tests/test_gpu_call_group_loader.py tests/test_gpu_sage_train.py -x -q 2>&1 | tail -3 >> $OUT/tests.log
tail -3 $OUT/tests.log
python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_plain.log 2>&1; grep "^{\"metric" $OUT/bench_plain.log | tail -1 > $OUT/bench_n1.json
tail -5 $OUT/bench_plain.log | cut -c1-600
python - <<PY
import json
d=json.load(open("$OUT/bench_n1.json")); print(d["value"], d["ms_per_step"], d.get("stage_ms_per_call_group"));
for k,v in d["variants"].items(): print(k, {a:b for a,b in v.items() if a!="note" and a!="wgrad_roofline"})
PY
