"""Wall-clock split (no profiler, no device sync) of the literal per-mini-batch loop: loader next() / layer 1 / layer 2."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch
from bench import rmat_csr
from cugraph_pyg_amd.data import FeatureStore, GraphStore
from cugraph_pyg_amd.loader import NeighborLoader
from wholegraph_amd import nn
dev = torch.device("cuda", 0)
V, E2 = 2_449_029, 61_859_140
row_ptr, col = rmat_csr(V, E2, 0, dev)
dst = torch.repeat_interleave(torch.arange(V, device=dev), row_ptr[1:] - row_ptr[:-1])
gs, fs = GraphStore(), FeatureStore()
gs[("n", "e", "n"), "coo", False, (V, V)] = torch.stack([col, dst])
fs["n", "x", None] = torch.rand((V, 100), device=dev)
del row_ptr, col, dst
convs = [nn.SAGEConv(100, 256).to(dev), nn.SAGEConv(256, 47).to(dev)]
B, n_b = 1024, 170
seeds = torch.randperm(V, device=dev)[:B * (n_b + 16)]
loader = NeighborLoader((fs, gs), [25, 10], input_nodes=seeds, batch_size=B, shuffle=False, random_state=62)
acc = [0.0, 0.0, 0.0, 0.0]
with torch.no_grad():
    it = iter(loader)
    for _ in range(16):
        next(it)
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for _ in range(n_b):
        t0 = time.perf_counter()
        batch = next(it)
        t1 = time.perf_counter()
        x, ei = batch.x, batch.edge_index
        t2 = time.perf_counter()
        h = convs[0](x, ei, act="relu")
        t3 = time.perf_counter()
        out = convs[1](h, ei)[:batch.batch_size]
        t4 = time.perf_counter()
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3
    host = time.perf_counter() - t_all
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_all
print("per batch (us): next() %.1f | attrs %.1f | conv1 %.1f | conv2 %.1f | host loop %.1f | wall incl. device drain %.1f"
      % tuple(v / n_b * 1e6 for v in acc + [host, wall]))
