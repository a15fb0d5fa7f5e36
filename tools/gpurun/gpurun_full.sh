# the driver's round-end sequence: GPU tests (serial, -x), smoke, bench
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/full; mkdir -p $OUT; cd $R
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "^\.*s*\.* *\[" $OUT/pytest.log | tail -15
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
