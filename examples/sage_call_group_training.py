#!/usr/bin/env python
"""The same GraphSAGE training as examples/sage_node_classification.py, iterated in CALL GROUPS: `loader.call_groups()` hands
out `local_seeds_per_call` seeds at a time as one block-diagonal graph (the reference samples that many seeds per library call
anyway, python/cugraph-pyg/cugraph_pyg/sampler/distributed_sampler.py:279-343, and only then cuts mini-batches out of it), the
features stay lazy (`grp.x`: the first layer's kernel reads the table through `n_id`), every SAGEConv layer is ONE kernel over the
group's trimmed layer graph — forward AND backward (`wholegraph_amd.nn._SageLayer`) — and the optimizer steps once per group.
On the products-like workload of bench.py this loop moves 2.3 G sampled edges/s through forward + loss + backward + SGD
(`variants.train_step`); the per-mini-batch loop of the other example is bound by its launches.

`--per-batch` keeps the reference's optimizer semantics instead — ONE step per mini-batch (pylibwholegraph/torch/gnn_model.py:
119-125) — while still sampling per call group: `cugraph_pyg_amd.loader.PerBatchStep` stages every mini-batch of a group into
fixed-size buffers with one launch and replays the whole step (forward, loss, backward, Adam) as one HIP graph.

    python examples/sage_call_group_training.py [--nodes 200000] [--epochs 3] [--per-batch]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]

import torch  # noqa: E402

from cugraph_pyg_amd.data import FeatureStore, GraphStore  # noqa: E402
from cugraph_pyg_amd.loader import NeighborLoader, PerBatchStep  # noqa: E402
from wholegraph_amd import nn as wnn  # noqa: E402
from wholegraph_amd.nn import SAGEConv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=200_000)
    ap.add_argument("--avg-degree", type=int, default=20)
    ap.add_argument("--features", type=int, default=100)
    ap.add_argument("--classes", type=int, default=16)
    ap.add_argument("--batch-size", type=int, default=1024)
    ap.add_argument("--group", type=int, default=16, help="mini-batches per call group (= per optimizer step)")
    ap.add_argument("--fanout", type=int, nargs="+", default=[25, 10])
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--per-batch", action="store_true", help="one optimizer step per mini-batch (the reference's semantics) through "
                                                             "loader.PerBatchStep instead of one per call group")
    args = ap.parse_args()
    assert torch.cuda.is_available(), "needs an MI355X (there is no CPU fallback)"
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    V, E = args.nodes, args.nodes * args.avg_degree
    src = (torch.rand(E, generator=g, device=dev) ** 2 * V).long().clamp_(max=V - 1)
    dst = torch.randint(0, V, (E,), generator=g, device=dev)
    community = torch.arange(V, device=dev) % args.classes
    same = torch.rand(E, generator=g, device=dev) < 0.7
    peer = (torch.randint(0, max(V // args.classes, 1), (E,), generator=g, device=dev) * args.classes + community[src]).clamp_(max=V - 1)
    dst = torch.where(same, peer, dst)
    x = torch.randn((V, args.features), generator=g, device=dev)
    x[torch.arange(V, device=dev), community % args.features] += 1.0
    graph_store, feature_store = GraphStore(), FeatureStore()
    graph_store[("node", "to", "node"), "coo", False, (V, V)] = torch.stack([src, dst])
    feature_store["node", "x", None] = x
    train_ids = torch.randperm(V, generator=g, device=dev)[: V // 2]
    loader = NeighborLoader((feature_store, graph_store), num_neighbors=args.fanout, input_nodes=train_ids,
                            batch_size=args.batch_size, shuffle=True, local_seeds_per_call=args.group * args.batch_size)
    L = len(args.fanout)
    dims = [args.features] + [128] * (L - 1) + [args.classes]
    convs = torch.nn.ModuleList(SAGEConv(dims[i], dims[i + 1]) for i in range(L)).to(dev)
    opt = torch.optim.Adam(convs.parameters(), lr=0.01, capturable=args.per_batch)
    if args.per_batch:
        return train_per_batch(args, loader, convs, opt, community, x, L)
    for epoch in range(args.epochs):
        t0, total, correct, seen, edges = time.perf_counter(), 0.0, 0, 0, 0
        for grp in loader.call_groups():
            h = grp.x                                               # LazyRows: nothing gathered
            for j, conv in enumerate(convs):
                h = conv(h, grp.layer_graph(j), act="relu" if j + 1 < L else None)
            y = community[grp.batch]                                # labels of the group's seeds, batch-major like h
            loss = wnn.cross_entropy(h, y)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            total += float(loss.detach()) * grp.num_seeds
            correct += int((h.argmax(1) == y).sum())
            seen += grp.num_seeds
            edges += grp.num_edges
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"epoch {epoch}: loss {total / seen:.4f}  train acc {correct / seen:.3f}  {edges / dt / 1e6:.1f} M sampled edges/s "
              f"(sampling + forward + backward + Adam, one step per {args.group} mini-batches), {dt:.2f} s")
    return total / seen, correct / seen


def train_per_batch(args, loader, convs, opt, community, table, L):
    """One optimizer step per mini-batch: the step below is what a reference user writes inside `for batch in loader`; it is
    captured once as a HIP graph over the staged mini-batch's fixed-size buffers and replayed for every mini-batch."""
    def step(batch):
        opt.zero_grad(set_to_none=True)
        h = batch.x                                                 # LazyRows over the staged n_id
        for j, conv in enumerate(convs):
            h = conv(h, batch.layer_graph(j), act="relu" if j + 1 < L else None)
        B = batch.batch_size
        y = community[batch.seeds]
        mask = batch.seed_mask[:B]                                  # a ragged last mini-batch has fewer than B live seeds
        # mean over the live seeds of the row_cap[0] output rows, no slice: two launches instead of torch's ten
        loss = wnn.cross_entropy(h, community[batch.n_id[:h.shape[0]]], batch.seed_mask)
        loss.backward()
        opt.step()
        return loss.detach(), ((h[:B].argmax(1) == y).float() * mask).sum()
    stepper = PerBatchStep(step, table=table, optimizer=opt)
    for epoch in range(args.epochs):
        t0, seen, edges, steps = time.perf_counter(), 0, 0, 0
        total = torch.zeros((), device=table.device)
        correct = torch.zeros((), device=table.device)
        for grp in loader.call_groups():
            for b in range(grp.n_batches):
                loss, ok = stepper(grp, b)
                total += loss                                       # (device-side running sums: no read-back per step)
                correct += ok
            seen += grp.num_seeds
            edges += grp.num_edges
            steps += grp.n_batches
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"epoch {epoch}: mean step loss {float(total) / steps:.4f}  train acc {float(correct) / seen:.3f}  "
              f"{edges / dt / 1e6:.1f} M sampled edges/s ({steps} Adam steps of {args.batch_size} seeds, {dt / steps * 1e3:.3f} ms each), {dt:.2f} s")
    return float(total) / steps, float(correct) / seen


if __name__ == "__main__":
    main()
