R=$GRAFT_REPO_ROOT; TAG=${1:-ws}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 600 python tools/ab_sage_ws.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.log
