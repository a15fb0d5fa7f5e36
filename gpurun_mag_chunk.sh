R=$GRAFT_REPO_ROOT; TAG=${1:-magchunk}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for ch in 0 32768 65536 131072 0 65536; do
  WGAMD_MAG_AGG_CHUNK=$ch timeout 600 python bench.py --workload mag --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('chunk=$ch value %.4g G  ms_per_step %.3f' % (d['value'] / 1e9, d['ms_per_step']))
"
done | tee $OUT/chunk.log
