# layer_cols as a (batch, chunk) block kernel: loader tests, the driver's line (loader_api / train_step), kernel trace of the training loop
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_lc; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_gpu_call_group_loader.py -m gpu -q -x -n 4 2>&1 | tail -2
true
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_train -o train -- python $R/tools/profile_train_groups.py 12 > $OUT/train_trace.log 2>&1
cp /tmp/pt_train/train_kernel_stats.csv $OUT/
grep -i "layer_cols\|target_rows" $OUT/train_kernel_stats.csv | cut -c1-60,150-260
