R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mt -o mt -- python $R/tools/profile_mag_train.py 4 128 > /tmp/mt.log 2>&1
tail -2 /tmp/mt.log
python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/mt/mt_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:28]:
    print("%-110s calls %5s avg %9.1f us %5.1f%%"%(r['Name'][:110].replace('void ','').replace('wgamd::(anonymous namespace)::',''),r['Calls'],float(r['AverageNs'])/1e3,100*float(r['TotalDurationNs'])/tot))
PY
