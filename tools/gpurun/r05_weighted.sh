# biased hop: parity tests, kernel split of a products call group's hop 2 (rocprofv3), op timing for 1 / 4 / 16 / 64 mini-batches.
# (WGAMD_WEIGHTED_SPLIT selected the hub-split experiment of round 5 — built, parity-green, no gain, removed: DESIGN 3.2)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/weighted; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_weighted_golden.py tests/test_gpu_callgroup.py -m gpu -x -q -n 3 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
for split in 1; do
  WGAMD_WEIGHTED_SPLIT=$split rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw_$split -o w -- python $R/tools/profile_weighted.py > $OUT/prof_$split.log 2>&1
  cp /tmp/pw_$split/w_kernel_stats.csv $OUT/weighted_split${split}_kernel_stats.csv
  cp /tmp/pw_$split/w_kernel_trace.csv $OUT/weighted_split${split}_kernel_trace.csv 2>/dev/null
  echo "== split=$split"; grep -i "weighted\|copy_short\|sample_count" $OUT/weighted_split${split}_kernel_stats.csv | sed -E 's/\(.*\)"/"/' | cut -c40-200
done
tail -14 $OUT/prof_1.log
cd $R
for split in 1; do
  WGAMD_WEIGHTED_SPLIT=$split python - <<'PY'
import os, sys
sys.path[:0] = [os.environ["GRAFT_REPO_ROOT"], os.path.join(os.environ["GRAFT_REPO_ROOT"], "cugraph-gnn_amd")]
import torch
from bench import rmat_csr, V_PRODUCTS, E_UNDIRECTED
from bench_ops import timed
from wholegraph_amd import wholegraph_ops
dev = torch.device("cuda", 0)
row_ptr, col = rmat_csr(V_PRODUCTS, E_UNDIRECTED, 0, dev)
g = torch.Generator(device=dev).manual_seed(3)
w = torch.rand(col.shape[0], generator=g, device=dev) + 0.01
seeds = torch.randperm(V_PRODUCTS, generator=g, device=dev)[:64 * 1024]
hop1 = wholegraph_ops.unweighted_sample_without_replacement(row_ptr, col, seeds, 25, random_seed=1)
frontier = torch.unique(hop1[1])
t = timed(lambda: wholegraph_ops.weighted_sample_without_replacement(row_ptr, col, w, frontier, 10, random_seed=62))
tu = timed(lambda: wholegraph_ops.unweighted_sample_without_replacement(row_ptr, col, frontier, 10, random_seed=62))
print("split", os.environ["WGAMD_WEIGHTED_SPLIT"], "call group of 64: weighted hop2 ms", t, "uniform", tu)
for nb in (1, 4, 16):
    hop1 = wholegraph_ops.unweighted_sample_without_replacement(row_ptr, col, seeds[:nb * 1024], 25, random_seed=1)
    fr = torch.unique(hop1[1])
    deg = row_ptr[fr + 1] - row_ptr[fr]
    t = timed(lambda: wholegraph_ops.weighted_sample_without_replacement(row_ptr, col, w, fr, 10, random_seed=62))
    tu = timed(lambda: wholegraph_ops.unweighted_sample_without_replacement(row_ptr, col, fr, 10, random_seed=62))
    print("  %d mini-batch(es): frontier %d, rows > 1024: %d, hubs: %d, longest %d: weighted ms %.4f uniform %.4f" % (
        nb, fr.numel(), int((deg > 1024).sum()), int((deg > 16384).sum()), int(deg.max()), t * 1e3, tu * 1e3))
PY
done
