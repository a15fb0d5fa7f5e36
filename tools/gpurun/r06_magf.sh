R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mf -o mf -- python $R/bench.py --workload mag --steps 6 --warmup 2 --no-variants --no-cpu-baseline > /tmp/mf.log 2>&1
grep '^{"metric' /tmp/mf.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('ms_per_step'), d.get('stage_ms_per_call_group'))"
python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/mf/mf_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:34]:
    print("%-100s calls %5s avg %9.1f us %5.1f%%"%(r['Name'][:100].replace('void ','').replace('wgamd::(anonymous namespace)::',''),r['Calls'],float(r['AverageNs'])/1e3,100*float(r['TotalDurationNs'])/tot))
PY
