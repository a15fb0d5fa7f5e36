R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
DEDUP=1 python $R/tools/profile_walk.py 2>&1 | tail -1
DEDUP=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wpu -o wp -- python $R/tools/profile_walk.py > /dev/null 2>&1
python - <<PY
import csv
for r in csv.DictReader(open('/tmp/wpu/wp_kernel_stats.csv')):
    if 'unique' in r['Name'] or 'scan_' in r['Name'] or 'fillBuffer' in r['Name']:
        print("%-80s calls %5s avg %8.1f us"%(r['Name'][:80],r['Calls'],float(r['AverageNs'])/1e3))
PY
cd $R; python -m pytest tests/test_gpu_dedup_fetch.py tests/test_gpu_renumber_gather.py -m gpu -x -q -k "unique or dedup" 2>&1 | tail -3
