R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_pb; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_call_group_loader.py tests/test_gpu_pyg_loader.py tests/test_gpu_pyg_reference_mirror.py tests/test_gpu_example_training.py tests/test_gpu_sage_train.py -m gpu -q -x -n 4 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "^\.*s*\.* *\[" $OUT/pytest.log | tail -30
python tools/profile_per_batch_host.py 400 2>&1 | grep -v amdgpu.ids | cut -c1-150 > $OUT/per_batch_host.txt; head -34 $OUT/per_batch_host.txt
