R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_mag; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_mag -o mag -- python $R/bench.py --workload mag --steps 6 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
cp /tmp/pt_mag/mag_kernel_stats.csv $OUT/
python $R/tools/trace_overlap.py /tmp/pt_mag/mag_kernel_trace.csv 0.3
python - <<'PY'
import csv,re
from collections import defaultdict
rows=list(csv.DictReader(open('/tmp/pt_mag/mag_kernel_trace.csv')))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
n=len(rows); rows=rows[int(n*0.3):int(n*0.8)]
# idle gaps and per-queue stats in the steady window
qs=defaultdict(list)
for r in rows: qs[r["Queue_Id"]].append(r)
for q,rs in qs.items():
    tot=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rs)
    span=int(rs[-1]["End_Timestamp"])-int(rs[0]["Start_Timestamp"])
    print("queue",q,"kernels",len(rs),"busy %.1f ms of span %.1f ms"%(tot/1e6,span/1e6))
# top kernels by total time in window
agg=defaultdict(lambda:[0,0])
for r in rows:
    k=re.sub(r'\(anonymous namespace\)::|wgamd::','',r["Kernel_Name"])[:70]
    agg[k][0]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"]); agg[k][1]+=1
for k,(t,c) in sorted(agg.items(), key=lambda kv:-kv[1][0])[:18]:
    print(f"{k:70s} {c:5d} {t/1e6:8.2f} ms  avg {t/c/1e3:7.1f} us")
PY
grep -o '"value": [0-9.]*' $OUT/trace.log | head -1
