# full GPU test suite (the driver's round-end command), log under gpurun_out/<tag>/
R=$GRAFT_REPO_ROOT; TAG=${1:-tests}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
shift
timeout 2400 python -m pytest tests -m gpu -q -n 4 "$@" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "^\.*s*\.* *\[" $OUT/pytest.log | tail -40
