#!/bin/bash
# per-kernel walk breakdown, no stream overlap: bash gpurun_walkprof.sh <tag> [extra bench args]
TAG=${1:-wp}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG -o wp -- python bench.py --steps 10 --warmup 3 --no-overlap --no-variants --no-cpu-baseline "$@" > gpurun_out/$TAG.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/$TAG/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:45]:
    if "wgamd" in r["Name"]:
        print("%-80s calls %5s avg %9.1f us min %8.1f"%(r["Name"][:80],r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3))
PY
