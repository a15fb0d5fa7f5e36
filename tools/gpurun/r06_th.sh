R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/th -o th -- python $R/tools/profile_train_host.py 9 > /tmp/th.log 2>&1
tail -6 /tmp/th.log
python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/th/th_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print("%-100s calls %5s avg %9.1f us %5.1f%%"%(r['Name'][:100].replace('void ','').replace('wgamd::(anonymous namespace)::',''),r['Calls'],float(r['AverageNs'])/1e3,100*float(r['TotalDurationNs'])/tot))
PY
