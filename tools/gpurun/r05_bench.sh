R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_bench; mkdir -p $OUT; cd $R
( time python bench.py --gpus 1 --steps 20 --warmup 5 $@ ) > $OUT/bench.log 2>&1; grep "^{\"metric" $OUT/bench.log | tail -1 > $OUT/bench_n1.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_bench/bench_n1.json"))
print("headline %.3f G  ms/step %.2f" % (d["value"]/1e9, d["ms_per_step"]))
for k,v in d["variants"].items():
    print(" ", k, "%.3f G" % ((v.get("value") or 0)/1e9), {a:b for a,b in v.items() if a in ("ms_per_call_group","ms_per_batch","forward_loss_ms","backward_step_ms","error")})
print(d["stage_ms_per_call_group"]); print({k:d["roofline"][k] for k in ("kernel","frac","avg_launch_ms")})
PY
