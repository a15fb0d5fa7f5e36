R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04d; mkdir -p $OUT; cd $R
timeout 1700 python -m pytest tests/test_gpu_mag_pipeline.py tests/test_gpu_bench_multirank.py -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "^\.*s*\.* *\[" $OUT/pytest.log | tail -60
