cd $GRAFT_REPO_ROOT
for b in tools/tune/bin/harness_*; do timeout 120 $b 550000; done
for i in 1 2 3; do python -m pytest tests/test_gpu_aggregate.py -q -m gpu -k "sage_layer_fused_matches and 4-128" 2>&1 | grep -E "passed|failed|AssertionError: np" ; done
python -m pytest tests/test_gpu_aggregate.py -q -m gpu 2>&1 | tail -8
