R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hd -o hd -- python $R/bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/hd/hd_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:45]:
    print("%-100s calls %5s avg %9.1f us %5.1f%%"%(r['Name'][:100].replace('void ','').replace('wgamd::(anonymous namespace)::',''),r['Calls'],float(r['AverageNs'])/1e3,100*float(r['TotalDurationNs'])/tot))
PY
