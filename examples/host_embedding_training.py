#!/usr/bin/env python
"""Trainable node embeddings that live in HOST memory behind a READWRITE device cache, trained with a sparse optimizer
next to a GraphSAGE classifier — the shape of the reference's WholeGraph examples run with
``--train-embedding --embedding-memory-location cpu --cache-type all_devices --cache-ratio R``
(python/pylibwholegraph/examples/node_classfication.py + pylibwholegraph/torch/common_options.py:137-205): nodes have no
input features, the model learns one vector per node.

    python examples/host_embedding_training.py [--nodes 200000] [--dim 64] [--cache-ratio 0.1] [--epochs 3]

The table ([nodes, dim] fp32 + LazyAdam's m / v / beta powers) sits in pinned host memory; every rank keeps a write-back
cache of its own rows in HBM (embedding row and optimizer state behind one tag, csrc/wg_embedding.hip).  A forward gather
and the optimizer step of a hot row touch HBM only; `writeback_all_cache` before `save` makes the host copy current.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import wholegraph_amd as wg  # noqa: E402                                  (reference: pylibwholegraph.torch as wgth)
from cugraph_pyg_amd.data import FeatureStore, GraphStore  # noqa: E402    (reference: cugraph_pyg.data)
from cugraph_pyg_amd.loader import NeighborLoader  # noqa: E402
from wholegraph_amd.nn import SAGEConv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=200_000)
    ap.add_argument("--avg-degree", type=int, default=15)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--classes", type=int, default=8)
    ap.add_argument("--batch-size", type=int, default=1024)
    ap.add_argument("--fanout", type=int, nargs="+", default=[10, 5])
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--cache-ratio", type=float, default=0.1)
    ap.add_argument("--embedding-memory-location", choices=["cpu", "cuda"], default="cpu")
    ap.add_argument("--lr", type=float, default=0.02)
    args = ap.parse_args()
    assert torch.cuda.is_available(), "needs an MI355X (there is no CPU fallback)"
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    V, E = args.nodes, args.nodes * args.avg_degree
    # communities: edges stay inside a community with probability 0.9, the label is the community
    label = torch.randint(0, args.classes, (V,), generator=g, device=dev)
    order = torch.argsort(label)
    bounds = torch.searchsorted(label[order].contiguous(), torch.arange(args.classes + 1, device=dev))
    src = (torch.rand(E, generator=g, device=dev) ** 2 * V).long().clamp_(max=V - 1)
    same = torch.rand(E, generator=g, device=dev) < 0.9
    lo, hi = bounds[label[src]], bounds[label[src] + 1]
    inside = order[(lo + (torch.rand(E, generator=g, device=dev) * (hi - lo).clamp(min=1)).long()).clamp_(max=V - 1)]
    dst = torch.where(same, inside, torch.randint(0, V, (E,), generator=g, device=dev))
    graph_store, feature_store = GraphStore(), FeatureStore()
    graph_store[("n", "e", "n"), "coo", False, (V, V)] = torch.stack([src, dst])
    feature_store["n", "y", None] = label

    # ---- the trainable table: host-resident, READWRITE device cache, sparse LazyAdam ---------------------------------
    comm = wg.get_global_communicator()
    policy = wg.create_builtin_cache_policy("all_devices", "distributed", args.embedding_memory_location, "readwrite",
                                            args.cache_ratio)
    table = wg.create_embedding(comm, "distributed", args.embedding_memory_location, torch.float32, [V, args.dim],
                                cache_policy=policy, random_init=True)
    sparse_opt = wg.create_wholememory_optimizer(table, "adam", {})
    lookup = wg.WholeMemoryEmbeddingModule(table)

    dims = [args.dim, 128, args.classes]
    convs = torch.nn.ModuleList(SAGEConv(dims[i], dims[i + 1]) for i in range(2)).to(dev)
    dense_opt = torch.optim.Adam(convs.parameters(), lr=args.lr)
    perm = torch.randperm(V, generator=g, device=dev)
    train_nodes = perm[: V // 2]
    loader = NeighborLoader((feature_store, graph_store), args.fanout, input_nodes=train_nodes, batch_size=args.batch_size,
                            shuffle=True, random_state=1)
    held_out = perm[V // 2: V // 2 + min(V // 10, 20 * args.batch_size)]
    eval_loader = NeighborLoader((feature_store, graph_store), args.fanout, input_nodes=held_out, batch_size=args.batch_size,
                                 shuffle=False, random_state=2)

    def forward(batch, x):
        n_nodes, n_edges = [int(v) for v in batch.num_sampled_nodes], [int(v) for v in batch.num_sampled_edges]
        for l, conv in enumerate(convs):
            hops = len(convs) - l
            e_keep, n_dst = sum(n_edges[:hops]), sum(n_nodes[:hops])
            x = conv((x, x[:n_dst]), batch.edge_index[:, :e_keep])
            if l + 1 < len(convs):
                x = F.relu(x)
        return x[:batch.batch_size]

    for epoch in range(args.epochs):
        lookup.train()
        t0, total, seen, correct = time.perf_counter(), 0.0, 0, 0
        for batch in loader:
            x = lookup(batch.n_id)                                   # rows through the cache; autograd parks the row gradients
            out, y = forward(batch, x), batch.y[:batch.batch_size]
            loss = F.cross_entropy(out, y)
            dense_opt.zero_grad()
            loss.backward()
            dense_opt.step()
            sparse_opt.step(args.lr)                                 # routed (row, gradient) pairs -> owners -> cache lines
            total += float(loss) * batch.batch_size
            seen += batch.batch_size
            correct += int((out.argmax(1) == y).sum())
        lookup.eval()
        ok = n_eval = 0
        with torch.no_grad():
            for batch in eval_loader:            # held-out nodes: their own rows were only ever trained as NEIGHBOURS
                ok += int((forward(batch, lookup(batch.n_id)).argmax(1) == batch.y[:batch.batch_size]).sum())
                n_eval += batch.batch_size
        hits, looked, lines = table.cache_stats()
        print("epoch %d: loss %.4f  train acc %.3f  held-out acc %.3f  %.2f s  cache: %d lines, hit rate %.3f" % (
            epoch, total / seen, correct / seen, ok / max(n_eval, 1), time.perf_counter() - t0, lines, hits / max(looked, 1)),
            flush=True)
    table.writeback_all_cache()                                      # the host table is current again
    host_rows = table.get_embedding_tensor().get_local_tensor()[0]
    print("table on %s, |row| mean %.4f after training" % (host_rows.device.type, float(host_rows.norm(dim=1).mean())))
    wg.destroy_embedding(table)
    wg.destroy_wholememory_optimizer(sparse_opt)
    wg.destroy_wholememory_cache_policy(policy)


if __name__ == "__main__":
    main()
