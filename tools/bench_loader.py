"""End-to-end throughput of the cugraph_pyg-shaped NeighborLoader (what a PyG training loop iterates): products-like RMAT,
fan-out [25, 10], batch 1024, features [V, 100] fp32 in a FeatureStore; every batch is a Data with x, edge_index, n_id..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch
from bench import rmat_csr
from cugraph_pyg_amd.data import FeatureStore, GraphStore
from cugraph_pyg_amd.loader import NeighborLoader
dev = torch.device("cuda", 0)
V, E2 = 2_449_029, 61_859_140
row_ptr, col = rmat_csr(V, E2, 0, dev)
dst = torch.repeat_interleave(torch.arange(V, device=dev), row_ptr[1:] - row_ptr[:-1])
gs, fs = GraphStore(), FeatureStore()
gs[("n", "e", "n"), "coo", False, (V, V)] = torch.stack([col, dst])
fs["n", "x", None] = torch.rand((V, 100), device=dev)
del row_ptr, col, dst
B = 1024
for calls in (1, 64, None):      # None = the loader's own default (sized from device memory, sampler.default_local_seeds_per_call)
    n_batches = 64 * 8
    seeds = torch.randperm(V, device=dev)[:B * n_batches]
    loader = NeighborLoader((fs, gs), [25, 10], input_nodes=seeds, batch_size=B,
                            local_seeds_per_call=None if calls is None else B * calls, shuffle=False)
    it = iter(loader); next(it)
    torch.cuda.synchronize(); t0 = time.perf_counter(); edges = 0; n = 0
    for batch in it:
        edges += int(batch.edge_index.shape[1]); n += 1
        _ = batch.x
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("call group %s: %.3f ms/batch, %.3f G sampled-edges/s (%d batches, x fetched per batch)" % ("default" if calls is None else "%3d" % calls, dt / n * 1e3, edges / dt / 1e9, n), flush=True)
