set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_call_group_loader.py tests/test_gpu_pyg_loader.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "^\.*s*\.* *\[" $OUT/pytest.log | tail -40
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2> $OUT/bench.err; tail -3 $OUT/bench.err
python - <<PY
import json
l=[x for x in open("$OUT/bench.log") if x.startswith('{"metric')][-1]
d=json.loads(l)
print("headline", d["value"], d["ms_per_step"], d["stage_ms_per_call_group"])
for k,v in d["variants"].items(): print(k, {a:b for a,b in v.items() if a!="note"})
PY
