R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_mag; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_call_group_loader.py -m gpu -q -x -n 4 > $OUT/pytest2.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest2.log
grep -v "^\.*s*\.* *\[" $OUT/pytest2.log | tail -40
