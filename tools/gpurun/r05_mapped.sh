R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_mapped; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_gpu_mapped_lazy_rows.py tests/test_gpu_sage_train.py tests/test_gpu_aggregate.py tests/test_gpu_callgroup.py -m gpu -q -x -n 4 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "^\.*s*\.* *\[" $OUT/pytest.log | tail -40
