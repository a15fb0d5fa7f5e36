R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_callgroup.py tests/test_gpu_pyg_loader.py -x -q 2>&1 | tail -2
for i in 1 2; do python tools/profile_walk.py 2>&1 | tail -1; done
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wpx -o wp -- python $R/tools/profile_walk.py > /dev/null 2>&1; grep "sample_uniform" /tmp/wpx/wp_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
