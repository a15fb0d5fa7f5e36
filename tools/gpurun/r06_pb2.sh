set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_pb2; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
cd $R
python tools/debug_pb.py > $OUT/debug.log 2>&1; tail -40 $OUT/debug.log
cd /tmp
python $R/tools/profile_walk.py 2>&1 | tail -1 > $OUT/walk.txt; cat $OUT/walk.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wpw -o wp -- python $R/tools/profile_walk.py > $OUT/walk_trace.log 2>&1
cp /tmp/wpw/wp_kernel_stats.csv $OUT/walk_kernel_stats.csv
for T in 1 0; do
GROUPS=1 TRAIN=$T rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pbt$T -o pb -- python $R/tools/profile_per_batch_step.py > $OUT/pb_trace_$T.log 2>&1
cp /tmp/pbt$T/pb_kernel_stats.csv $OUT/pb_kernel_stats_train$T.csv
tail -2 $OUT/pb_trace_$T.log | cut -c1-500
done
