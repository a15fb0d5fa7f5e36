R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_mag; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_mag_pipeline.py tests/test_gpu_pyg_loader.py -m gpu -q -x -n 4 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "^\.*s*\.* *\[" $OUT/pytest.log | tail -40
timeout 900 python bench.py --workload mag --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_mag.log 2>&1; tail -c 3000 $OUT/bench_mag.log | grep -o '"value": [0-9.]*\|"stage_ms_per_call_group": {[^}]*}\|"kernel_all_launches": {[^}]*}' ; tail -5 $OUT/bench_mag.log | grep -v "^{" | tail -5
