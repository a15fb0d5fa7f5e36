set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_final; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 3000 python -m pytest tests/ -q -m gpu 2>&1 | tail -6 > $OUT/tests_all.log
tail -4 $OUT/tests_all.log | cut -c1-300
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_plain.log 2>&1; grep "^{\"metric" $OUT/bench_plain.log | tail -1 > $OUT/bench_n1.json
python $R/bench.py --workload mag --steps 6 --warmup 2 > $OUT/bench_mag.log 2>&1; grep "^{\"metric" $OUT/bench_mag.log | tail -1 > $OUT/bench_mag_hetero_n1.json
python - <<PY
import json
d=json.load(open("$OUT/bench_n1.json")); print(d["value"], d["ms_per_step"], d.get("stage_ms_per_call_group"), d["roofline"]["frac"], d["roofline"].get("frac_profiled"), d["cpu_baseline"]["value"]);
for k,v in d["variants"].items(): print(k, {a:b for a,b in v.items() if a not in ("note","wgrad_roofline","buffers")})
print(d["variants"]["train_step"].get("wgrad_roofline"))
m=json.load(open("$OUT/bench_mag_hetero_n1.json")); print("mag", m["value"], m["variants"]["train_step"]["value"], m["roofline"]["frac"], m["roofline"].get("traffic_over_algorithmic"))
PY
