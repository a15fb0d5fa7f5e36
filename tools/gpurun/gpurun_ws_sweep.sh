# A/B of library builds (cugraph-gnn_amd/lib/variants/libwholegraph_amd.<v>.so) on ONE box: tools/ab_sage_ws.py arm per build
R=$GRAFT_REPO_ROOT; TAG=${1:-wssweep}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
cp cugraph-gnn_amd/lib/libwholegraph_amd.so /tmp/keep.so
for rep in 1 2; do for v in $2; do
  cp cugraph-gnn_amd/lib/variants/libwholegraph_amd.$v.so cugraph-gnn_amd/lib/libwholegraph_amd.so
  echo "== variant $v"
  WGAMD_SAGE_WS=${WS:-0} AB_ARM=1 AB_OUT=/tmp/x.pt timeout 300 python tools/ab_sage_ws.py 2>&1 | grep -v amdgpu.ids
done; done | tee $OUT/sweep.log
cp /tmp/keep.so cugraph-gnn_amd/lib/libwholegraph_amd.so
