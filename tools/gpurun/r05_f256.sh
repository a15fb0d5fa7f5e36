cd $GRAFT_REPO_ROOT
for v in base p3; do for dbg in 0 64; do
  echo "== lib $v WGAMD_SAGE_DEBUG=$dbg"; WGAMD_SAGE_DEBUG=$dbg F=256 N=256 ND=1000000 NS=4000000 WGAMD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/tune/bin/libmf_$v.so timeout 300 python tools/bench_sage_fused.py 2>&1 | grep "fused" | sed 's/.*| fused/fused/'
done; done
