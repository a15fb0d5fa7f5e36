"""Throughput of LinkNeighborLoader (edge seeds + binary negatives) on the products-like graph."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch
from bench import rmat_csr
from cugraph_pyg_amd.data import FeatureStore, GraphStore
from cugraph_pyg_amd.loader import LinkNeighborLoader
dev = torch.device("cuda", 0)
V, E2 = 2_449_029, 61_859_140
row_ptr, col = rmat_csr(V, E2, 0, dev)
dst = torch.repeat_interleave(torch.arange(V, device=dev), row_ptr[1:] - row_ptr[:-1])
gs, fs = GraphStore(), FeatureStore()
gs[("n", "e", "n"), "coo", False, (V, V)] = torch.stack([col, dst])
fs["n", "x", None] = torch.rand((V, 100), device=dev)
E = col.shape[0]
sel = torch.randint(0, E, (512 * 200,), device=dev)
eli = torch.stack([col[sel], dst[sel]])
del row_ptr
loader = LinkNeighborLoader((fs, gs), num_neighbors=[25, 10], edge_label_index=eli, batch_size=512, neg_sampling=("binary", 1.0),
                            shuffle=False)
it = iter(loader); next(it); next(it)
torch.cuda.synchronize(); t0 = time.perf_counter(); edges = n = 0
for batch in it:
    edges += int(batch.edge_index.shape[1]); n += 1
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("LinkNeighborLoader batch 512 (+512 negatives): %.3f ms/batch, %.3f G sampled-edges/s (%d batches)" % (dt / n * 1e3, edges / dt / 1e9, n))
