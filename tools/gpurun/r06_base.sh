# round 6 baseline: the new tests of this round, the walk alone under the kernel trace, the plain bench line
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_base; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_bench_multirank.py tests/test_gpu_dist_tensor.py tests/test_gpu_mapped_lazy_rows.py tests/test_gpu_sage_train.py -x -q 2>&1 | tail -15 > $OUT/tests.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wp -o wp -- python $R/tools/profile_walk.py > $OUT/walk.log 2>&1
cp /tmp/wp/wp_kernel_stats.csv $OUT/
python $R/tools/profile_walk.py > $OUT/walk_plain.log 2>&1
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_plain.log 2>&1; grep "^{\"metric" $OUT/bench_plain.log | tail -1 > $OUT/bench_n1.json
cat $OUT/tests.log $OUT/walk_plain.log
