# LDS renumber-table kernel: size / thread-count variants, built on the GPU box, timed with the walk-only pass
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/tune_lds; mkdir -p $OUT; cd $R
run() {
  tag=$1; shift
  touch cugraph-gnn_amd/csrc/wg_append_unique.hip
  make -C cugraph-gnn_amd/csrc -j8 -s EXTRA="$*" > $OUT/build_$tag.log 2>&1 || { echo "$tag: BUILD FAILED"; tail -5 $OUT/build_$tag.log; return; }
  for i in 1 2; do python tools/profile_walk.py 2>/dev/null | tail -1 | sed "s/^/$tag: /"; done
}
run default
run t1024u4 -DWG_LDS_SLOTS=10000 -DWG_LDS_KEYS=3000 -DWG_LDS_THREADS=1024 -DWG_LDS_UNROLL=4
run s6500 -DWG_LDS_SLOTS=6500 -DWG_LDS_KEYS=2000 -DWG_LDS_THREADS=512 -DWG_LDS_UNROLL=5
run s5000 -DWG_LDS_SLOTS=5000 -DWG_LDS_KEYS=1500 -DWG_LDS_THREADS=512 -DWG_LDS_UNROLL=4
run s5000t256 -DWG_LDS_SLOTS=5000 -DWG_LDS_KEYS=1500 -DWG_LDS_THREADS=256 -DWG_LDS_UNROLL=7
run s3300t256 -DWG_LDS_SLOTS=3300 -DWG_LDS_KEYS=1000 -DWG_LDS_THREADS=256 -DWG_LDS_UNROLL=5
run s10000k4000 -DWG_LDS_SLOTS=10000 -DWG_LDS_KEYS=4000 -DWG_LDS_THREADS=512 -DWG_LDS_UNROLL=9
