"""Host profile (cProfile) of the literal PyG loop `for batch in NeighborLoader: SAGEConv(batch.x, batch.edge_index) x 2` on the
products-like graph — bench.py's `loader_api_per_batch` variant: which Python functions the ~0.2 ms per mini-batch go to."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch  # noqa: E402
from bench import rmat_csr  # noqa: E402
from cugraph_pyg_amd.data import FeatureStore, GraphStore  # noqa: E402
from cugraph_pyg_amd.loader import NeighborLoader  # noqa: E402
from wholegraph_amd import nn  # noqa: E402

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
n_b = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seeds = torch.randperm(V, device=dev)[:B * (n_b + 16)]
loader = NeighborLoader((fs, gs), [25, 10], input_nodes=seeds, batch_size=B, shuffle=False, random_state=62)
prof = cProfile.Profile()
n, edges = 0, 0
with torch.no_grad():
    it = iter(loader)
    for _ in range(16):
        next(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prof.enable()
    for batch in it:
        h = batch.x
        for j, c in enumerate(convs):
            h = c(h, batch.edge_index, act="relu" if j == 0 else None)
        _ = h[:batch.batch_size]
        edges += int(batch.edge_index.shape[1])
        n += 1
    prof.disable()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("%d batches, %.3f ms per batch (under cProfile), %.3f G edges/s" % (n, dt / n * 1e3, edges / dt / 1e9))
st = pstats.Stats(prof)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)
