# round 6 profiles: rocprofv3 passes of the DRIVER's command (python bench.py --steps 20 --warmup 5), of the walk alone, of the
# per-mini-batch training step and of the mag workload (8 edge types); kernel trace + stats first, then SEPARATE --pmc passes.
set -x
R=$GRAFT_REPO_ROOT; TAG=${1:-r06}; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
# (0) the un-profiled line
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_plain.log 2>&1; grep "^{\"metric" $OUT/bench_plain.log | tail -1 > $OUT/bench_n1.json
# (1) driver command under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_$TAG -o $TAG -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
cp /tmp/pt_$TAG/${TAG}_kernel_stats.csv $OUT/
python $R/tools/trace_large_launches.py /tmp/pt_$TAG/${TAG}_kernel_trace.csv $OUT/${TAG}_kernel_stats_large.csv
grep "^{\"metric" $OUT/bench_trace.log | tail -1 > $OUT/bench_under_rocprof.json
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "row_copy|spmm_csr|sage_layer_fused|sage_layer_mfma|sample_uniform|renumber_lds|bucket_sort|renumber_emit|first_bits|scan_tile|sample_count|unique_" --output-format csv -d /tmp/pc_${TAG}_$C -o ${TAG}_$C -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $OUT/bench_$C.log 2>&1
  cp /tmp/pc_${TAG}_$C/*counter_collection.csv $OUT/
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "Cijk|sage_layer_fused|sage_layer_mfma" --output-format csv -d /tmp/pc_${TAG}_MFMA -o ${TAG}_MFMA -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $OUT/bench_MFMA.log 2>&1
cp /tmp/pc_${TAG}_MFMA/*counter_collection.csv $OUT/
# (2) the walk alone (what each of its launches costs with the chip to itself)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_walk -o walk -- python $R/tools/profile_walk.py > $OUT/walk_trace.log 2>&1
cp /tmp/pt_walk/walk_kernel_stats.csv $OUT/
python $R/tools/profile_walk.py 2>&1 | tail -1 > $OUT/walk_alone.txt
DEDUP=1 python $R/tools/profile_walk.py 2>&1 | tail -1 > $OUT/walk_dedup_alone.txt
DEDUP=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_walkd -o walkd -- python $R/tools/profile_walk.py > /dev/null 2>&1
cp /tmp/pt_walkd/walkd_kernel_stats.csv $OUT/walk_dedup_kernel_stats.csv
# (3) the training loops: per call group (profile_train_groups) and per mini-batch (PerBatchStep)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_train -o train -- python $R/tools/profile_train_groups.py 16 > $OUT/train_trace.log 2>&1
cp /tmp/pt_train/train_kernel_stats.csv $OUT/
python $R/tools/trace_large_launches.py /tmp/pt_train/train_kernel_trace.csv $OUT/train_kernel_stats_large.csv
tail -1 $OUT/train_trace.log | grep "^{" > $OUT/train_under_rocprof.json
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "sage_wgrad|wgrad_reduce|sage_layer_mfma" --output-format csv -d /tmp/pc_train_$C -o train_$C -- python $R/tools/profile_train_groups.py 8 > $OUT/train_$C.log 2>&1
  cp /tmp/pc_train_$C/*counter_collection.csv $OUT/
done
GROUPS=1 TRAIN=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_pb -o pb -- python $R/tools/profile_per_batch_step.py > $OUT/per_batch_trace.log 2>&1
cp /tmp/pt_pb/pb_kernel_stats.csv $OUT/per_batch_step_kernel_stats.csv
grep "^{" $OUT/per_batch_trace.log | tail -1 > $OUT/per_batch_under_rocprof.txt
# (4) mag (BASELINE configs[4], all 8 directed edge types) through the package API; the 6-type graph of rounds 3-5 next to it
python $R/bench.py --workload mag --steps 6 --warmup 2 > $OUT/bench_mag_plain.log 2>&1; grep "^{\"metric" $OUT/bench_mag_plain.log | tail -1 > $OUT/bench_mag_hetero_n1.json
python $R/bench.py --workload mag --mag-rels r5 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_mag_r5.log 2>&1; grep "^{\"metric" $OUT/bench_mag_r5.log | tail -1 > $OUT/bench_mag_6types_n1.json
python $R/bench.py --workload mag --call-group 64 --steps 6 --warmup 2 --no-cpu-baseline --no-variants > $OUT/bench_mag_g64.log 2>&1; grep "^{\"metric" $OUT/bench_mag_g64.log | tail -1 > $OUT/bench_mag_hetero_g64_n1.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_mag -o mag -- python $R/bench.py --workload mag --steps 6 --warmup 2 --no-cpu-baseline > $OUT/mag_trace.log 2>&1
cp /tmp/pt_mag/mag_kernel_stats.csv $OUT/
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "gat_layer_fused|gat_aggregate_heads|gather_terms" --output-format csv -d /tmp/pc_mag_$C -o mag_$C -- python $R/bench.py --workload mag --steps 4 --warmup 2 --no-cpu-baseline --no-variants > $OUT/mag_$C.log 2>&1
  cp /tmp/pc_mag_$C/*counter_collection.csv $OUT/
done
ls -la $OUT
