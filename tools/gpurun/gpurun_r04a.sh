# round 4, first contact: GPU tests -> walk-only kernel breakdown -> driver bench line
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04a; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wp -o wp -- python $R/tools/profile_walk.py > $OUT/walk.log 2>&1
cp /tmp/wp/wp_kernel_stats.csv $OUT/walk_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/walk_kernel_stats.csv")))
for r in rows[:30]:
    if "wgamd" in r["Name"]:
        print("%-90s calls %5s avg %9.1f us min %8.1f"%(r["Name"][:90],r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3))
PY
tail -2 $OUT/walk.log
cd $R
python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2> $OUT/bench.err; tail -c 3000 $OUT/bench.log; tail -5 $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-overlap --no-variants --no-cpu-baseline > $OUT/bench_serial.log 2>&1; grep -o '"value": [0-9.e+]*\|"ms_per_step": [0-9.]*' $OUT/bench_serial.log | head -4
