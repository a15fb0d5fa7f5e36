# GAT relations reading the feature tables through the node lists (fetch in the layer): tests, then bench_mag A/B on one box
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_mag_lazy; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_gpu_mag_pipeline.py tests/test_gpu_call_group_loader.py -m gpu -q -x -n 4 2>&1 | tail -3
for lazy in 1 1; do
  WGAMD_GAT_FETCH_IN_LAYER=$lazy timeout 900 python bench.py --workload mag --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_$lazy.log 2>&1
  python - $OUT/bench_$lazy.log $lazy <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]
d = json.loads(l)
print("fetch_in_layer", sys.argv[2], "value", round(d["value"] / 1e9, 3), "ms/group", round(d["ms_per_step"] / 8, 3), {k: v for k, v in d["config"].get("stage_ms_per_call_group", d.get("stage_ms_per_call_group", {})).items()} if False else "")
st = d.get("stage_ms_per_call_group") or d["config"].get("stage_ms_per_call_group") or {}
print("   ", {k: round(v, 3) for k, v in st.items()})
print("   ", (d.get("roofline") or {}).get("frac"), ((d.get("roofline") or {}).get("kernel_all_launches") or {}).get("frac"))
PY
done
