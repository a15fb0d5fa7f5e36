"""Training step of BASELINE configs[4] through the package API: NeighborLoader.call_groups() (HeteroCallGroup) -> 2 x
nn.HeteroConv{GATConv 4x64} (autograd: relation by relation through GATConv) -> cross-entropy on synthetic labels -> backward ->
SGD, one optimizer step per call group.  usage: python tools/profile_mag_train.py [groups] [mini-batches per group]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch
import bench_mag as bm

n_groups = int(sys.argv[1]) if len(sys.argv) > 1 else 6
G = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
graphs, num_nodes = bm.build_mag_like(dev)
etypes, ntypes = sorted(graphs), sorted(num_nodes)
g = torch.Generator(device=dev).manual_seed(5)
tables = {t: torch.rand((num_nodes[t], bm.F_IN), generator=g, device=dev) * 2 - 1 for t in ntypes}
model = bm.build_model(bm.make_params(etypes, ntypes, dev), etypes, ntypes, dev)
params = [p for m in model for p in m.parameters()]
for p in params:
    p.requires_grad_(True)
head = torch.nn.Linear(bm.HC, 16).to(dev)
opt = torch.optim.SGD(params + list(head.parameters()), lr=1e-3)
B, warm = 1024, 2
seeds = torch.randperm(num_nodes["paper"], generator=g, device=dev)[:(n_groups + warm) * G * B]
labels = torch.randint(0, 16, (num_nodes["paper"],), generator=g, device=dev)
loader = bm.make_loader(bm.build_mag_like.graph_store, tables, seeds, B, G)
n, edges, t0 = 0, 0, None
for grp in loader.call_groups():
    if n == warm:
        torch.cuda.synchronize(); t0, edges = time.perf_counter(), 0
    out = head(bm.forward_group(model, grp))
    y = labels[seeds[n * G * B:(n + 1) * G * B]]
    loss = torch.nn.functional.cross_entropy(out, y)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    edges += grp.num_edges
    n += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("mag train step: %d call groups of %d mini-batches: %.3f G sampled edges/s, %.2f ms per group, loss %.4f" % (
    n - warm, G, edges / dt / 1e9, dt / (n - warm) * 1e3, float(loss.detach())))
