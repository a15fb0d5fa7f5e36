cd $GRAFT_REPO_ROOT
for n in base gb4; do for m in 0 2 6 1; do echo "$n: $(tools/tune/bin/abl_$n 550000 $m | tail -1)"; done; done
timeout 900 python -m pytest tests/test_gpu_aggregate.py -x -q -k fused 2>&1 | tail -3
