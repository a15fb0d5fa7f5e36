# round 5: training tests, a short bench.py run (train_step variant), and a kernel trace of the training loop
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_train; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_gpu_sage_train.py -m gpu -q -x -n 4 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "^\.*s*\.* *\[" $OUT/pytest.log | tail -15
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench.log 2>&1; tail -c 6000 $OUT/bench.log | grep -o '"train_step": {[^}]*}'
tail -c 6000 $OUT/bench.log | grep -o '"value": [0-9.]*' | head -1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_train -o train -- python $R/tools/profile_train_groups.py 12 > $OUT/train_trace.log 2>&1
cp /tmp/pt_train/train_kernel_stats.csv $OUT/
tail -2 $OUT/train_trace.log
head -25 $OUT/train_kernel_stats.csv | cut -c1-200
