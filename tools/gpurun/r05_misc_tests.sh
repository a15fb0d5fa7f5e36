R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_misc; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_gpu_dist_tensor.py tests/test_gpu_call_group_loader.py tests/test_gpu_embedding_rw_cache.py tests/test_gpu_embedding_cache.py tests/test_gpu_pyg_loader.py -m gpu -q -n 4 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "^\.*s*\.* *\[" $OUT/pytest.log | tail -50
