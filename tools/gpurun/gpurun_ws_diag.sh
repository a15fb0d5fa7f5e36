R=$GRAFT_REPO_ROOT; TAG=${1:-wsdiag}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for d in 0 1 2 8 16 17 3 ; do
  echo "== WGAMD_SAGE_DEBUG=$d"
  WGAMD_SAGE_DEBUG=$d WGAMD_SAGE_WS=1 AB_ARM=1 AB_OUT=/tmp/x.pt timeout 300 python tools/ab_sage_ws.py 2>&1 | grep -v amdgpu.ids
done | tee $OUT/diag.log
