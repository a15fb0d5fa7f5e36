set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_full; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_per_batch_step.py -x -q 2>&1 | tail -30 > $OUT/tests_pb.log
tail -12 $OUT/tests_pb.log | cut -c1-300
timeout 3000 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -15 > $OUT/tests_all.log
tail -15 $OUT/tests_all.log | cut -c1-300
python $R/bench.py --workload mag --steps 6 --warmup 2 > $OUT/bench_mag.log 2>&1; grep "^{\"metric" $OUT/bench_mag.log | tail -1 > $OUT/bench_mag_hetero_n1.json
python $R/bench.py --workload mag --mag-rels r5 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_mag_r5.log 2>&1; grep "^{\"metric" $OUT/bench_mag_r5.log | tail -1 > $OUT/bench_mag_r5rels.json
python - <<PY
import json
for f in ("bench_mag_hetero_n1.json","bench_mag_r5rels.json"):
    d=json.load(open("$OUT/"+f)); print(f, d["value"], d["ms_per_step"], d.get("stage_ms_per_call_group")); print(d.get("variants"))
PY
