# rocprofv3 passes of bench.py (run on the GPU box through gpurun); summaries land in gpurun_out/prof_rNN/
set -x
R=$GRAFT_REPO_ROOT; TAG=${1:-r01}; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_$TAG -o $TAG -- python $R/bench.py --steps 640 --warmup 64 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
cp /tmp/pt_$TAG/${TAG}_kernel_stats.csv $OUT/
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "row_copy|spmm_csr|sample_uniform|table_insert" --output-format csv -d /tmp/pc_${TAG}_$C -o ${TAG}_$C -- python $R/bench.py --steps 256 --warmup 64 --no-cpu-baseline > $OUT/bench_$C.log 2>&1
  cp /tmp/pc_${TAG}_$C/*counter_collection.csv $OUT/
done
tail -c 1500 $OUT/bench_trace.log | tail -1 > $OUT/bench_under_rocprof.json
ls -la $OUT
