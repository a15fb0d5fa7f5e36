R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_gpu_aggregate.py -m gpu -x -q -k "transpose" 2>&1 | tail -5
python -m pytest tests/test_gpu_per_batch_step.py tests/test_gpu_sage_train.py tests/test_gpu_cross_entropy.py tests/test_gpu_example_training.py -m gpu -x -q 2>&1 | tail -5
cd /tmp; export TMPDIR=/tmp
GROUPS=2 TRAIN=1 python $R/tools/profile_per_batch_step.py 2>&1 | tail -1
GROUPS=2 TRAIN=0 python $R/tools/profile_per_batch_step.py 2>&1 | tail -1
GROUPS=1 TRAIN=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_pb -o pb -- python $R/tools/profile_per_batch_step.py > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/pt_pb/pb_kernel_stats.csv')))
for r in rows:
    c=int(r['Calls'])
    if c>=900: print("%-100s calls %5d avg %7.1f us"%(r['Name'][:100],c,float(r['AverageNs'])/1e3))
PY
