R=$GRAFT_REPO_ROOT; cd $R
python tools/bench_gat_agg_bwd.py 2>&1 | tail -6
python -m pytest tests/test_gpu_mag_pipeline.py tests/test_gpu_call_group_loader.py tests/test_gpu_aggregate.py -m gpu -q -k "gat or hetero or homogeneous" 2>&1 | tail -3
