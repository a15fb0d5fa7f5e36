R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_gpu_cross_entropy.py tests/test_gpu_mag_pipeline.py tests/test_gpu_call_group_loader.py -m gpu -q 2>&1 | tail -4
bash tools/gpurun/r06_tr.sh
