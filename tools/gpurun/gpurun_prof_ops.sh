# rocprofv3 kernel stats of bench_ops.py (per-operator table) -> gpurun_out/prof_ops_<tag>/
R=$GRAFT_REPO_ROOT; TAG=${1:-r01}; OUT=$R/gpurun_out/prof_ops_$TAG; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/po_$TAG -o $TAG -- python $R/bench_ops.py --json $OUT/ops_n1.json > $OUT/ops.log 2>&1
cp /tmp/po_$TAG/${TAG}_kernel_stats.csv $OUT/
head -30 $OUT/${TAG}_kernel_stats.csv | cut -c1-200
