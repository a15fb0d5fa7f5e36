R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_pb; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_pb -o pb -- python $R/tools/profile_per_batch_wall.py > $OUT/trace.log 2>&1
cp /tmp/pt_pb/pb_kernel_stats.csv $OUT/
tail -1 $OUT/trace.log
python - <<'PY'
import csv,re
rows=list(csv.DictReader(open('/tmp/pt_pb/pb_kernel_stats.csv')))
for r in rows[:14]:
    n=re.sub(r'\(anonymous namespace\)::|wgamd::','',r['Name'])
    print(f"{n[:100]:100s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f}us {float(r['TotalDurationNs'])/1e6:8.2f}ms")
PY
