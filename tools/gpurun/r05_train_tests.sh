# round 5: training-path parity tests + the aggregation suite (forward must be unchanged)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_train; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_gpu_sage_train.py tests/test_gpu_aggregate.py -m gpu -q -x -n 4 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "^\.*s*\.* *\[" $OUT/pytest.log | tail -60
