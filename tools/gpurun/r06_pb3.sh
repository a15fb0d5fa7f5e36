set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_pb3; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_per_batch_step.py -x -q 2>&1 | tail -30 > $OUT/tests.log
tail -30 $OUT/tests.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_sage_train.py tests/test_gpu_call_group_loader.py tests/test_gpu_example_training.py -x -q 2>&1 | tail -3 >> $OUT/tests.log
tail -3 $OUT/tests.log
cd /tmp
for T in 1 0; do
GROUPS=1 TRAIN=$T rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pbt$T -o pb -- python $R/tools/profile_per_batch_step.py > $OUT/pb_trace_$T.log 2>&1
cp /tmp/pbt$T/pb_kernel_stats.csv $OUT/pb_kernel_stats_train$T.csv
grep "^{" $OUT/pb_trace_$T.log | cut -c1-500
done
GROUPS=3 TRAIN=1 python $R/tools/profile_per_batch_step.py 2>&1 | grep "^{" | cut -c1-500
