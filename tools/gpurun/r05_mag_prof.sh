# mag (BASELINE configs[4]) profile set of round 5: un-profiled line (with variants.train_step), kernel trace, PMC passes, training trace
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r05_mag; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
python $R/bench.py --workload mag --steps 6 --warmup 2 > $OUT/bench_mag_plain.log 2>&1; grep "^{\"metric" $OUT/bench_mag_plain.log | tail -1 > $OUT/bench_mag_hetero_n1.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_mag -o mag -- python $R/bench.py --workload mag --steps 6 --warmup 2 --no-cpu-baseline --no-variants > $OUT/mag_trace.log 2>&1
cp /tmp/pt_mag/mag_kernel_stats.csv $OUT/
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "gat_layer_fused|gat_aggregate_heads|gather_terms" --output-format csv -d /tmp/pc_mag_$C -o mag_$C -- python $R/bench.py --workload mag --steps 4 --warmup 2 --no-cpu-baseline --no-variants > $OUT/mag_$C.log 2>&1
  cp /tmp/pc_mag_$C/*counter_collection.csv $OUT/
done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_magtrain -o mag_train -- python $R/tools/profile_mag_train.py 4 128 > $OUT/mag_train_trace.log 2>&1
cp /tmp/pt_magtrain/mag_train_kernel_stats.csv $OUT/
tail -1 $OUT/mag_train_trace.log
ls -la $OUT
