R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
G=128 python $R/tools/profile_hetero_walk.py 2>&1 | tail -2
G=128 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hw -o hw -- python $R/tools/profile_hetero_walk.py > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/hw/hw_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
n=0
for r in rows:
    c=int(r['Calls'])
    if c>=4:
        print("%-95s calls/run %6.1f avg %8.1f us  per-run %7.1f us"%(r['Name'][:95],c/4,float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/4e3)); n+=1
    if n>45: break
print("launches per run", sum(int(r['Calls']) for r in rows)/4)
PY
