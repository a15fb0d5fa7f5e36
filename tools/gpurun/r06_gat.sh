set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_gat; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_gpu_mag_pipeline.py tests/test_gpu_call_group_loader.py tests/test_gpu_aggregate.py -x -q 2>&1 | tail -5 > $OUT/tests.log; cat $OUT/tests.log | cut -c1-300
python tools/bench_gat_agg_bwd.py 2>&1 | tail -3
python tools/profile_gat_products.py 2>&1 | tail -3 | cut -c1-600
