# end-of-round check on the final tree: full GPU suite (xdist), smoke, the driver's bench command, the per-operator table
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04f; mkdir -p $OUT; cd $R
( time timeout 2400 python -m pytest tests/ -q -m gpu -n 4 ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "^\.*s*\.* *\[" $OUT/pytest.log | tail -25
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench_n1.json
timeout 1100 python bench_ops.py --json $OUT/ops_n1.json > $OUT/ops.log 2>&1; echo "ops rc=$?"; grep -n "host table\|writeback" $OUT/ops.log | cut -c1-330
