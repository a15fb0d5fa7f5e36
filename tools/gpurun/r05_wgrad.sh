R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_wgrad; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_sage_train.py -m gpu -q -x -n 4 2>&1 | tail -3
timeout 600 python tools/bench_wgrad.py 2>&1 | tee $OUT/wgrad.log | tail
