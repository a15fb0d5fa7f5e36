# BASELINE configs[2..4] on ONE MI355X, each as one JSON line carrying roofline + cpu_baseline, plus a rocprofv3
# kernel-trace summary of a short profiled pass of the same command -> gpurun_out/configs_<tag>/
set -x
R=$GRAFT_REPO_ROOT; TAG=${1:-r03}; OUT=$R/gpurun_out/configs_$TAG; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
python $R/bench.py --workload papers100m --steps 10 --warmup 3 --cpu-budget 10 > $OUT/papers100m.log 2> $OUT/papers100m.err
grep '^{"metric' $OUT/papers100m.log | tail -1 > $OUT/bench_papers100m_n1.json
python $R/bench.py --workload rmat26 --steps 6 --warmup 2 --cpu-budget 10 > $OUT/rmat26.log 2> $OUT/rmat26.err
grep '^{"metric' $OUT/rmat26.log | tail -1 > $OUT/bench_rmat26_n1.json
python $R/bench.py --workload mag --steps 6 --warmup 2 --cpu-budget 10 > $OUT/mag.log 2> $OUT/mag.err
grep '^{"metric' $OUT/mag.log | tail -1 > $OUT/bench_mag_hetero_n1.json
for W in papers100m rmat26; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$W -o $W -- python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $OUT/prof_$W.log 2>&1
  cp /tmp/pc_$W/${W}_kernel_stats.csv $OUT/
done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_mag -o mag -- python $R/bench.py --workload mag --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_mag.log 2>&1
cp /tmp/pc_mag/mag_kernel_stats.csv $OUT/
ls -la $OUT; tail -c 600 $OUT/*.err
