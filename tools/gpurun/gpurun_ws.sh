# weight-stationary SAGE layer: parity tests + A/B against the producer / consumer kernel
R=$GRAFT_REPO_ROOT; TAG=${1:-ws}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_aggregate.py -q -x -n 4 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "^\.*s*\.* *\[" $OUT/pytest.log | tail -15
timeout 600 python tools/ab_sage_ws.py 2>&1 | tee $OUT/ab.log
