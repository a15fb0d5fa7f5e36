R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r06_last; mkdir -p $OUT
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_plain.log 2>&1; grep '^{"metric' $OUT/bench_plain.log | tail -1 > $OUT/bench_n1.json
python bench.py --workload mag --steps 6 --warmup 2 > $OUT/bench_mag_plain.log 2>&1; grep '^{"metric' $OUT/bench_mag_plain.log | tail -1 > $OUT/bench_mag_hetero_n1.json
python -c "
import json
d=json.load(open('$OUT/bench_n1.json')); print(round(d['value']/1e9,3), d['value_row_for_row_fetch'], {k:(round(v['value']/1e9,3) if v.get('value') else v) for k,v in d['variants'].items()})
m=json.load(open('$OUT/bench_mag_hetero_n1.json')); print('mag', round(m['value']/1e9,3), {k:round(v['value']/1e9,3) for k,v in m['variants'].items()})"
