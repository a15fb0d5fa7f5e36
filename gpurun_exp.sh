cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_weighted_golden.py -x -q 2>&1 | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wexp_new -o w -- python tools/profile_weighted.py > gpurun_out/wexp_new.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/wexp_new/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
spans=[]; cur=None
for r in rows:
    n=r["Kernel_Name"]
    if "sample_weighted" in n or "copy_short_rows" in n:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        if cur is None: cur=[s,e,0]
        cur[1]=max(cur[1],e); cur[2]+=e-s
    elif "sample_count_kernel" in n and cur is not None:
        spans.append(cur); cur=None
if cur: spans.append(cur)
for s,e,b in spans: print("biased hop span %.1f us, kernel-time sum %.1f us"%((e-s)/1e3,b/1e3))
PY
