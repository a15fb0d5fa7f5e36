R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
python $R/tools/profile_per_batch.py 2>&1 | tail -12
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lit -o lit -- python $R/tools/profile_per_batch.py > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/lit/lit_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print("%-100s calls %5s avg %9.1f us %5.1f%%"%(r['Name'][:100].replace('void ','').replace('wgamd::(anonymous namespace)::',''),r['Calls'],float(r['AverageNs'])/1e3,100*float(r['TotalDurationNs'])/tot))
PY
