set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_dedup; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_dedup_fetch.py tests/test_gpu_renumber_gather.py -x -q 2>&1 | tail -12 | cut -c1-300
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1; grep "^{\"metric" $OUT/bench.log | tail -1 > $OUT/bench_n1.json; tail -3 $OUT/bench.log | cut -c1-300
python - <<PY
import json
d=json.load(open("$OUT/bench_n1.json")); print(d["value"], d["ms_per_step"], d.get("stage_ms_per_call_group"), d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("also",{}).get("frac"), d["feature_fetch"]);
for k,v in d["variants"].items(): print(k, {a:b for a,b in v.items() if a not in ("note","wgrad_roofline","buffers")})
PY
timeout 1200 python -m pytest tests/test_gpu_bench_multirank.py -x -q 2>&1 | tail -8 | cut -c1-300
