# final un-profiled lines of round 6 on the final tree: the driver's command, then configs 3 and 4 with their traces / PMC passes
R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r06_final; mkdir -p $OUT
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_plain.log 2>&1; grep '^{"metric' $OUT/bench_plain.log | tail -1 > $OUT/bench_n1.json
WORKLOADS="papers100m rmat26" bash tools/gpurun/gpurun_configs.sh r06 > $OUT/configs.log 2>&1
tail -c 600 $OUT/bench_n1.json
