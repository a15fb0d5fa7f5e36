R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp -o tp -- python $R/tools/profile_train_groups.py 12 > /tmp/tp.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/tp/tp_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:40]:
    print("%-100s calls %5s avg %9.1f us %5.1f%%"%(r['Name'][:100].replace('void ','').replace('wgamd::(anonymous namespace)::',''),r['Calls'],float(r['AverageNs'])/1e3,100*float(r['TotalDurationNs'])/tot))
PY
