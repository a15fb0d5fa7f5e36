R=$GRAFT_REPO_ROOT; cd $R
for v in base $@; do
  echo "== $v"; WGAMD_LIBRARY_PATH=$R/tools/tune/bin/libwg_$v.so timeout 300 python tools/bench_wgrad.py 2>&1 | grep "^wgrad"
done
