"""Where does a PyG-style training step of examples/sage_node_classification.py spend its time? (host wall per section)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd"), os.path.join(ROOT, "examples")]
import torch
import torch.nn.functional as F
import sage_node_classification as ex
from cugraph_pyg_amd.data import FeatureStore, GraphStore
from cugraph_pyg_amd.loader import NeighborLoader
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
V, E = 200_000, 4_000_000
src = (torch.rand(E, generator=g, device=dev) ** 2 * V).long().clamp_(max=V - 1)
dst = torch.randint(0, V, (E,), generator=g, device=dev)
gs, fs = GraphStore(), FeatureStore()
gs[("node", "to", "node"), "coo", False, (V, V)] = torch.stack([src, dst])
fs["node", "x", None] = torch.randn((V, 100), generator=g, device=dev)
fs["node", "y", None] = torch.randint(0, 16, (V,), generator=g, device=dev)
loader = NeighborLoader((fs, gs), num_neighbors=[25, 10], input_nodes=torch.randperm(V, device=dev)[:100 * 1024], batch_size=1024,
                        shuffle=False, local_seeds_per_call=16 * 1024)
model = ex.SAGE(100, 128, 16, 2).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=0.01)
acc = {"loader": 0.0, "forward": 0.0, "backward": 0.0, "optimizer": 0.0}
it = iter(loader)
n = 0
while True:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    try:
        batch = next(it)
    except StopIteration:
        break
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out = model(batch.x, batch.edge_index, batch.num_sampled_nodes, batch.num_sampled_edges)[: batch.batch_size]
    loss = F.cross_entropy(out, batch.y[: batch.batch_size])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    opt.zero_grad(); loss.backward()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    opt.step()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    if n >= 20:
        acc["loader"] += t1 - t0; acc["forward"] += t2 - t1; acc["backward"] += t3 - t2; acc["optimizer"] += t4 - t3
    n += 1
print({k: round(v / (n - 20) * 1e3, 3) for k, v in acc.items()}, "ms per batch;", batch.edge_index.shape[1], "edges/batch")
