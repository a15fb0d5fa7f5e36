set -x
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_per_batch_step.py tests/test_gpu_sage_train.py tests/test_gpu_example_training.py tests/test_gpu_aggregate.py -x -q 2>&1 | tail -4 | cut -c1-300
GROUPS=3 TRAIN=1 python tools/profile_per_batch_step.py 2>&1 | grep "^{" | cut -c1-400
GROUPS=3 TRAIN=0 python tools/profile_per_batch_step.py 2>&1 | grep "^{" | cut -c1-400
