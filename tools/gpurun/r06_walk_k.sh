# K seeds per lane group x list / vertex-grouped order: parity, then the walk alone
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_walk_k; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_gpu_callgroup.py tests/test_gpu_pyg_loader.py tests/test_gpu_call_group_loader.py -x -q 2>&1 | tail -3 > $OUT/tests.log
cat $OUT/tests.log
cd /tmp
for K in 1 2 4; do for L in 0 1048576; do
  echo "K=$K L=$L" >> $OUT/summary.txt
  WGAMD_SAMPLE_K=$K WGAMD_SAMPLE_LOCALITY=$L python $R/tools/profile_walk.py 2>&1 | tail -1 >> $OUT/summary.txt
done; done
for K in 2 4; do for L in 0 1048576; do
WGAMD_SAMPLE_K=$K WGAMD_SAMPLE_LOCALITY=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wp_${K}_$L -o wp -- python $R/tools/profile_walk.py > $OUT/trace_${K}_$L.log 2>&1
cp /tmp/wp_${K}_$L/wp_kernel_stats.csv $OUT/wp_kernel_stats_K${K}_loc$L.csv
done; done
cat $OUT/summary.txt
