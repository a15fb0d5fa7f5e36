"""Host side of the call-group training loop (bench.py's `train_step`): how long the HOST takes to enqueue each phase of a group
(no device syncs inside the loop), next to the wall time per group — the loop is device-bound only while the first stays below
the second.   usage: python tools/profile_train_host.py [n_groups]   (<= 9: the epoch's last groups — a shorter group, the
ragged mini-batch — build their own walk objects once per process, ~100 ms each, and are kept out of the average)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from wholegraph_amd import nn  # noqa: E402
from cugraph_pyg_amd.data import FeatureStore, GraphStore  # noqa: E402
from cugraph_pyg_amd.loader import NeighborLoader  # noqa: E402

n_groups = min(int(sys.argv[1]) if len(sys.argv) > 1 else 9, 9)
dev = torch.device("cuda", 0)
V, E_u, F, C, fan = bench.WORKLOADS["products"]
row_ptr, col = bench.rmat_csr(V, E_u, seed=0, device=dev)
table = torch.rand((V, F), generator=torch.Generator(device=dev).manual_seed(100), device=dev) * 2 - 1
dst = torch.repeat_interleave(torch.arange(V, device=dev), row_ptr[1:] - row_ptr[:-1])
gs, fs = GraphStore(), FeatureStore()
gs[("n", "e", "n"), "coo", False, (V, V)] = torch.stack([col, dst])
fs["n", "x", None] = table
del dst
convs = torch.nn.ModuleList([nn.SAGEConv(F, bench.HIDDEN), nn.SAGEConv(bench.HIDDEN, C)]).to(dev)
opt = torch.optim.SGD(convs.parameters(), lr=0.01)
labels = torch.randint(0, C, (V,), device=dev)
seeds = torch.randperm(V, device=dev)
loader = NeighborLoader((fs, gs), fan, input_nodes=seeds, batch_size=bench.BATCH, shuffle=False, random_state=62)
inner = {}


def _timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            inner[name] = inner.get(name, 0.0) + time.perf_counter() - t
    return w


if os.environ.get("INNER", "1") == "1":      # host time inside the backward pass, by piece (the engine runs it on its own thread)
    nn.sage_wgrad = _timed("sage_wgrad", nn.sage_wgrad)
    nn._sage_dx = _timed("_sage_dx", nn._sage_dx)
    nn._csr_transpose = _timed("_csr_transpose", nn._csr_transpose)
    nn.sage_layer_fused_forward = _timed("sage_layer_fused_forward", nn.sage_layer_fused_forward)
    nn._SageLayer.backward = staticmethod(_timed("_SageLayer.backward", nn._SageLayer.backward))
    nn._SoftmaxXent.backward = staticmethod(_timed("_SoftmaxXent.backward", nn._SoftmaxXent.backward))
acc, n, t_all = {}, 0, None
t_prev = time.perf_counter()
for grp in loader.call_groups():
    t0 = time.perf_counter()
    if n == 3:
        torch.cuda.synchronize()
        acc, t_all, t0 = {}, time.perf_counter(), time.perf_counter()
        inner.clear()
        t_prev = t0
    acc["loader (next group: wait for its sizes)"] = acc.get("loader (next group: wait for its sizes)", 0.0) + t0 - t_prev
    h = grp.x
    for j, c in enumerate(convs):
        h = c(h, grp.layer_graph(j), act="relu" if j == 0 else None)
    t1 = time.perf_counter()
    loss = nn.cross_entropy(h, labels[grp.batch])
    t2 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    for k, v in (("forward (2 layers)", t1 - t0), ("loss", t2 - t1), ("backward", t3 - t2), ("optimizer step", t4 - t3)):
        acc[k] = acc.get(k, 0.0) + v
    t_prev = t4
    n += 1
    if n == 3 + n_groups:
        break
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) / n_groups * 1e3
print("wall %.2f ms per group; host per group:" % wall)
for k, v in acc.items():
    print("  %-44s %.3f ms" % (k, v / n_groups * 1e3))
for k, v in sorted(inner.items(), key=lambda kv: -kv[1]):
    print("    inside: %-36s %.3f ms" % (k, v / n_groups * 1e3))
if os.environ.get("COMPARE", "0") == "1":     # bench.py's own train_step over the same graph, same process
    for c in convs:
        c.in_channels = (c.in_channels, c.in_channels) if isinstance(c.in_channels, int) else c.in_channels
    G = grp.n_batches
    out = bench.loader_api_variants(row_ptr, col, table, convs, seeds, 9, 188, which=("train_step",))
    print({k: (round(v["value"] / 1e9, 3), round(v["ms_per_call_group"], 2)) for k, v in out.items()})
