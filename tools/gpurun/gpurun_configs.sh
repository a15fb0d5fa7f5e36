# BASELINE configs[2..4] on ONE MI355X, each as one JSON line carrying roofline + cpu_baseline, a rocprofv3 kernel-trace
# summary of a short profiled pass of the same command, and SEPARATE --pmc FETCH_SIZE / WRITE_SIZE passes (never combined with
# a trace option; WORKLOADS="mag" restricts the run) of its hot kernels -> gpurun_out/configs_<tag>/ ; tools/pmc_summary.py turns the counter CSVs into
# pmc_traffic_<workload>.json, which the bench lines then quote as roofline.traffic / frac_profiled.
set -x
R=$GRAFT_REPO_ROOT; TAG=${1:-r04}; OUT=$R/gpurun_out/configs_$TAG; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
HOT="row_copy|spmm_csr|sage_layer_fused|sage_layer_mfma|sample_uniform|renumber_lds|bucket_sort|renumber_emit|gat_aggregate_heads|gat_layer_fused|gat_transform|gather_terms|gat_csr"
for W in ${WORKLOADS:-papers100m rmat26 mag}; do
  EXTRA="--no-variants"; [ $W = mag ] && EXTRA=""
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$W -o $W -- python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/prof_$W.log 2>&1
  cp /tmp/pc_$W/${W}_kernel_stats.csv $OUT/
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-include-regex "$HOT" --output-format csv -d /tmp/pm_${W}_$C -o ${W}_$C -- python $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/pmc_${W}_$C.log 2>&1
    cp /tmp/pm_${W}_$C/*counter_collection.csv $OUT/
  done
  python $R/tools/pmc_summary.py $OUT $W pmc_traffic_$W.json > $OUT/pmc_summary_$W.txt 2>&1
done
# the lines themselves, AFTER the profiles exist next to the repo's committed ones (copy them in so the lines can quote them)
mkdir -p $R/profiles/$TAG; cp $OUT/*_kernel_stats.csv $OUT/pmc_traffic_*.json $R/profiles/$TAG/ 2>/dev/null
cd $R
for W in ${WORKLOADS:-papers100m rmat26 mag}; do
  case $W in papers100m) ST="--steps 10 --warmup 3";; *) ST="--steps 6 --warmup 2";; esac
  N=$W; [ $W = mag ] && N=mag_hetero
  python bench.py --workload $W $ST --cpu-budget 10 > $OUT/$W.log 2> $OUT/$W.err
  grep '^{"metric' $OUT/$W.log | tail -1 > $OUT/bench_${N}_n1.json
done
ls -la $OUT; tail -c 400 $OUT/*.err; head -c 600 $OUT/bench_*_n1.json
