# PMC passes over the walk alone: what the hop-2 sampling kernel fetches in list order and in vertex-grouped order
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_walk_pmc; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for L in 0 1048576; do
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
    T=$(echo $C | tr ' ' '_')
    ITERS=4 WGAMD_SAMPLE_LOCALITY=$L rocprofv3 --pmc $C --kernel-include-regex "sample_uniform|locality|renumber_lds|renumber_emit|bucket_sort|first_bits" --output-format csv -d /tmp/pc_${L}_$T -o pc -- python $R/tools/profile_walk.py > $OUT/log_${L}_$T.log 2>&1
    cp /tmp/pc_${L}_$T/pc_counter_collection.csv $OUT/pc_${L}_$T.csv
  done
done
python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$OUT/pc_*.csv")):
    acc=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        k=(r["Kernel_Name"][:60],r["Counter_Name"])
        acc[k][0]+=1; acc[k][1]+=float(r["Counter_Value"])
    print("==",f.split("/")[-1])
    for k,(n,v) in sorted(acc.items()):
        print("  %-62s %-22s launches %4d avg %.4g"%(k[0],k[1],n,v/n))
PY
