R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_sage_train.py tests/test_gpu_mapped_lazy_rows.py -m gpu -q -x -n 4 2>&1 | tail -3
bash tools/gpurun/r05_wgrad_abl.sh NO_STAGGER ABL_NO_MFMA ABL_NO_Z
