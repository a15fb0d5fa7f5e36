set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_full; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
cd $R
timeout 3000 python -m pytest tests/ -q -m gpu 2>&1 | tail -15 > $OUT/tests_all.log
tail -8 $OUT/tests_all.log | cut -c1-300
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_plain.log 2>&1; grep "^{\"metric" $OUT/bench_plain.log | tail -1 > $OUT/bench_n1.json
python - <<PY
import json
d=json.load(open("$OUT/bench_n1.json")); print(d["value"], d["ms_per_step"], d.get("stage_ms_per_call_group"), d["roofline"]["frac"], d["cpu_baseline"]["value"]);
for k,v in d["variants"].items(): print(k, {a:b for a,b in v.items() if a not in ("note","wgrad_roofline")})
PY
G=128 python $R/tools/profile_hetero_walk.py 2>&1 | tail -3; RELS=r5 G=128 python $R/tools/profile_hetero_walk.py 2>&1 | tail -3
