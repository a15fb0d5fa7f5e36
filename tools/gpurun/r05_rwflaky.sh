cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/rwflaky
for p in 1 2 3 4 5 6; do (for i in 1 2 3 4 5 6; do python -m pytest tests/test_gpu_embedding_rw_cache.py -m gpu -q -k "write_back" > /tmp/rw_${p}_$i.log 2>&1 || cp /tmp/rw_${p}_$i.log gpurun_out/rwflaky/; done) & done; wait
ls gpurun_out/rwflaky | wc -l
for f in gpurun_out/rwflaky/*; do grep -E "^E  |test_gpu_embedding_rw_cache.py:[0-9]+: " $f | cut -c1-300 | head -12; echo ----; done 2>/dev/null | head -60
