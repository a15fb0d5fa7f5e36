# round 6: the shortened walk — parity tests, then the walk alone under the kernel trace with and without the vertex-grouped order
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_walk; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_gpu_callgroup.py tests/test_gpu_sampling.py tests/test_gpu_renumber_gather.py tests/test_gpu_pyg_loader.py tests/test_gpu_call_group_loader.py tests/test_gpu_partitioned_csr.py tests/test_reference_py_fixtures.py tests/test_gpu_c_abi.py -x -q 2>&1 | tail -15 > $OUT/tests.log
tail -5 $OUT/tests.log
cd /tmp
for L in 0 1048576; do
  WGAMD_SAMPLE_LOCALITY=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wp$L -o wp -- python $R/tools/profile_walk.py > $OUT/walk_trace_$L.log 2>&1
  cp /tmp/wp$L/wp_kernel_stats.csv $OUT/wp_kernel_stats_loc$L.csv
  WGAMD_SAMPLE_LOCALITY=$L python $R/tools/profile_walk.py > $OUT/walk_plain_$L.log 2>&1
  tail -1 $OUT/walk_plain_$L.log
done
for WL in papers100m rmat26; do for L in 0 1048576; do
  WORKLOAD=$WL G=64 ITERS=8 WGAMD_SAMPLE_LOCALITY=$L timeout 600 python $R/tools/profile_walk.py > $OUT/walk_${WL}_$L.log 2>&1; tail -1 $OUT/walk_${WL}_$L.log
done; done
python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_plain.log 2>&1; grep "^{\"metric" $OUT/bench_plain.log | tail -1 > $OUT/bench_n1.json
python - <<PY
import json
d=json.load(open("$OUT/bench_n1.json")); print(d["value"], d["ms_per_step"], d.get("stage_ms_per_call_group"))
PY
