"""Kernel split of one biased hop (wholegraph_csr_weighted_sample_without_replacement, hop-2 shape of the products
workload); run under rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch
from bench import rmat_csr, V_PRODUCTS, E_UNDIRECTED
from wholegraph_amd import wholegraph_ops
dev = torch.device("cuda", 0)
row_ptr, col = rmat_csr(V_PRODUCTS, E_UNDIRECTED, 0, dev)
g = torch.Generator(device=dev).manual_seed(3)
w = torch.rand(col.shape[0], generator=g, device=dev) + 0.01
seeds = torch.randperm(V_PRODUCTS, generator=g, device=dev)[:64 * 1024]
hop1 = wholegraph_ops.unweighted_sample_without_replacement(row_ptr, col, seeds, 25, random_seed=1)
frontier = torch.unique(hop1[1])
print("frontier", frontier.numel(), flush=True)
deg = (row_ptr[frontier + 1] - row_ptr[frontier]).cpu()
for lo, hi in [(0, 10), (10, 16), (16, 32), (32, 64), (64, 128), (128, 256), (256, 512), (512, 1024), (1024, 12288), (12288, 16384), (16384, 1 << 30)]:
    m = (deg > lo) & (deg <= hi)
    print("deg (%d, %d]: rows %d candidates %d" % (lo, hi, int(m.sum()), int(deg[m].sum())), flush=True)
for i in range(6):
    out = wholegraph_ops.weighted_sample_without_replacement(row_ptr, col, w, frontier, 10, random_seed=5 + i)
torch.cuda.synchronize()
print("edges", out[1].numel())
