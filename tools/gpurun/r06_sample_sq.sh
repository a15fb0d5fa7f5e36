set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_sample_sq; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  ITERS=3 rocprofv3 --pmc $C --kernel-include-regex "sample_uniform" --output-format csv -d /tmp/sq_$T -o pc -- python $R/tools/profile_walk.py > $OUT/log_$T.log 2>&1
  cp /tmp/sq_$T/pc_counter_collection.csv $OUT/pc_$T.csv
done
python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$OUT/pc_*.csv")):
    acc=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        big = float(r["Grid_Size"])>1e7 if "Grid_Size" in r else True
        k=(r["Kernel_Name"][60:100],r["Counter_Name"], r.get("Grid_Size"))
        acc[k][0]+=1; acc[k][1]+=float(r["Counter_Value"])
    for k,(n,v) in sorted(acc.items()):
        print("  %-42s grid %-10s %-34s launches %3d avg %.5g"%(k[0],k[2],k[1],n,v/n))
PY
