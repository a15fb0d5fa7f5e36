cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_aggregate.py -x -q -k fused 2>&1 | tail -3
for D in 0 1 2; do
echo "products F=100 debug=$D: $(WGAMD_SAGE_DEBUG=$D python tools/bench_sage_fused.py 2>&1 | grep -v amdgpu.ids | sed 's/.*gemm = //')"
echo "papers F=128 debug=$D: $(WGAMD_SAGE_DEBUG=$D F=128 python tools/bench_sage_fused.py 2>&1 | grep -v amdgpu.ids | sed 's/.*gemm = //')"
done
