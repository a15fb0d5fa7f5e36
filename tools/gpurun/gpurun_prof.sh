# rocprofv3 passes of bench.py (run on the GPU box through gpurun); summaries land in gpurun_out/prof_<tag>/
# Every pass runs the DRIVER'S command (python bench.py --steps 20 --warmup 5): the launch shape is fixed (8 call groups of 191 per step).
# kernel trace + stats first, then SEPARATE --pmc passes (never combined with a trace option).
set -x
R=$GRAFT_REPO_ROOT; TAG=${1:-r04}; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_$TAG -o $TAG -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
cp /tmp/pt_$TAG/${TAG}_kernel_stats.csv $OUT/
# rocprofv3's per-kernel average mixes every launch shape of a kernel (the call-group launches of the headline pipeline, the
# 1024-seed launches of the per-batch variant, ...): the dominant-shape statistics come from the trace itself
python $R/tools/trace_large_launches.py /tmp/pt_$TAG/${TAG}_kernel_trace.csv $OUT/${TAG}_kernel_stats_large.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "row_copy|spmm_csr|sage_layer_fused|sage_layer_mfma|sample_uniform|renumber_lds|bucket_sort|renumber_emit|scan_tile|sample_count" --output-format csv -d /tmp/pc_${TAG}_$C -o ${TAG}_$C -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $OUT/bench_$C.log 2>&1
  cp /tmp/pc_${TAG}_$C/*counter_collection.csv $OUT/
done
# MFMA utilisation of the dense tail (hipBLASLt fp32 GEMMs): busy cycles of the matrix pipe vs GPU-active cycles
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "Cijk|sage_layer_fused|sage_layer_mfma" --output-format csv -d /tmp/pc_${TAG}_MFMA -o ${TAG}_MFMA -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $OUT/bench_MFMA.log 2>&1
cp /tmp/pc_${TAG}_MFMA/*counter_collection.csv $OUT/
grep "^{\"metric" $OUT/bench_trace.log | tail -1 > $OUT/bench_under_rocprof.json
ls -la $OUT
