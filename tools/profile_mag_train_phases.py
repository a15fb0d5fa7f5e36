"""Where a mag training step spends its time: per phase the HOST time to enqueue it (no sync) and the DEVICE time until it is done
(sync after the phase), un-profiled.  usage: python tools/profile_mag_train_phases.py [mini-batches per group]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch
import bench_mag as bm

G = int(sys.argv[1]) if len(sys.argv) > 1 else 64
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
B, n_groups = 1024, 6
seeds = torch.randperm(num_nodes["paper"], generator=g, device=dev)[:n_groups * G * B]
labels = torch.randint(0, 16, (num_nodes["paper"],), generator=g, device=dev)
loader = bm.make_loader(bm.build_mag_like.graph_store, tables, seeds, B, G)
acc = {}


def phase(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    h, d = acc.setdefault(name, [0.0, 0.0])
    acc[name] = [h + (t1 - t0), d + (t2 - t0)]
    return out


n = 0
it = iter(loader.call_groups())
while True:
    grp = phase("next(call group): walk + sizes", lambda: next(it, None))
    if grp is None:
        break
    if n == 2:
        acc.clear()
    h = grp.x_dict
    lg = [phase("layer_graph(%d)" % j, lambda j=j: grp.layer_graph(j)) for j in range(2)]
    h = phase("layer 1 forward", lambda: model[0](h, lg[0], act="relu"))
    h = phase("layer 2 forward", lambda: model[1](h, lg[1], act="relu"))
    loss = phase("head + loss", lambda: torch.nn.functional.cross_entropy(head(h["paper"]), labels[seeds[n * G * B:(n + 1) * G * B]]))
    phase("zero_grad", lambda: opt.zero_grad(set_to_none=True))
    phase("backward", lambda: loss.backward())
    phase("optimizer step", lambda: opt.step())
    n += 1
k = n - 2
print("per call group of %d mini-batches (%d groups): phase  host-enqueue ms  until-done ms" % (G, k))
for name, (h, d) in acc.items():
    print("  %-34s %8.2f %8.2f" % (name, h / k * 1e3, d / k * 1e3))
print("  %-34s %8.2f %8.2f" % ("sum", sum(v[0] for v in acc.values()) / k * 1e3, sum(v[1] for v in acc.values()) / k * 1e3))
