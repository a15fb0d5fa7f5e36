"""Where a classic per-mini-batch PyG loop spends its time on this stack: `for batch in NeighborLoader: SAGEConv(batch.x,
batch.edge_index) x 2` on the products-like graph (host timers with a device sync after every step)."""
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
B = 1024
seeds = torch.randperm(V, device=dev)[:B * 160]
loader = NeighborLoader((fs, gs), [25, 10], input_nodes=seeds, batch_size=B, shuffle=False)
acc = {}
def tick(name, t0):
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()
n = 0
with torch.no_grad():
    it = iter(loader)
    for _ in range(16):
        next(it)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for batch in it:
        t = tick("loader next()", t)
        x, ei = batch.x, batch.edge_index
        t = tick("batch.x / edge_index", t)
        rp, ci = nn._to_csr(ei, x.shape[0])
        t = tick("coo -> csr", t)
        agg = nn.spmm_csr(x, rp, ci, "mean")
        t = tick("spmm 1", t)
        h = torch.relu_(convs[0].lin_l(agg) + convs[0].lin_r(x))
        t = tick("linears 1", t)
        agg = nn.spmm_csr(h, rp, ci, "mean")
        t = tick("spmm 2", t)
        out = convs[1].lin_l(agg) + convs[1].lin_r(h)
        t = tick("linears 2", t)
        n += 1
print("%d batches" % n)
for k, v in acc.items():
    print("  %-24s %.3f ms per batch" % (k, v / n * 1e3))
