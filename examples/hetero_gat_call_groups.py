#!/usr/bin/env python
"""A heterogeneous GAT through the package's call groups — the shape of the reference's ogbn-mag example
(python/cugraph-pyg/cugraph_pyg/examples/mag_lp_mnmg.py:141: `HeteroConv({edge_type: GATConv(...)})` over a `NeighborLoader` on
a heterogeneous `GraphStore`) on a planted graph: papers carry almost no signal of their class, their AUTHORS do, so the model
has to use the `author -writes-> paper` relation.

  * training: `loader.call_groups()` -> `HeteroCallGroup` -> 2 x `wholegraph_amd.nn.HeteroConv` relation by relation through
    `GATConv` (autograd), one optimizer step per call group;
  * inference: the same layers without gradients take the call-group route of `nn.HeteroConv` — aggregate-first relations as
    one kernel each, the feature tables read THROUGH the group's node lists (`x_dict` stays a dict of `LazyRows`: no per-group
    copy of the rows), attention terms of the tables' rows.  The example checks that this route and the gather-first one give
    the same bits.

    python examples/hetero_gat_call_groups.py [--papers 40000] [--epochs 3]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from cugraph_pyg_amd.data import FeatureStore, GraphStore  # noqa: E402
from cugraph_pyg_amd.loader import NeighborLoader  # noqa: E402
from wholegraph_amd import nn  # noqa: E402

F_IN, HEADS, CH = 128, 4, 64


def forward(layers, head, grp):
    h = grp.x_dict
    for j, layer in enumerate(layers):
        h = layer(h, grp.layer_graph(j), act="relu")
    return head(h["paper"])          # rows = the seeds of all mini-batches, in input order


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--papers", type=int, default=40_000)
    ap.add_argument("--authors", type=int, default=20_000)
    ap.add_argument("--classes", type=int, default=8)
    ap.add_argument("--batch-size", type=int, default=512)
    ap.add_argument("--group", type=int, default=4, help="mini-batches per call group (= per optimizer step)")
    ap.add_argument("--epochs", type=int, default=3)
    args = ap.parse_args()
    assert torch.cuda.is_available(), "needs an MI355X (there is no CPU fallback)"
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    P, A, C = args.papers, args.authors, args.classes
    paper_class = torch.randint(0, C, (P,), generator=g, device=dev)
    author_class = torch.randint(0, C, (A,), generator=g, device=dev)
    # every paper has ~4 authors, 85 % of them from the paper's class
    n_w = 4 * P
    w_paper = torch.arange(n_w, device=dev) % P
    pick = torch.randint(0, A, (n_w,), generator=g, device=dev)
    by_class = torch.argsort(author_class, stable=True)
    first = torch.searchsorted(author_class[by_class], torch.arange(C + 1, device=dev))
    want = paper_class[w_paper]
    span = (first[want + 1] - first[want]).clamp_(min=1)
    same = by_class[(first[want] + (torch.rand(n_w, generator=g, device=dev) * span).long()).clamp_(max=A - 1)]
    w_author = torch.where(torch.rand(n_w, generator=g, device=dev) < 0.85, same, pick)
    c_src, c_dst = torch.randint(0, P, (3 * P,), generator=g, device=dev), torch.randint(0, P, (3 * P,), generator=g, device=dev)
    gs, fs = GraphStore(), FeatureStore()
    gs[("author", "writes", "paper"), "coo", False, (A, P)] = torch.stack([w_author, w_paper])
    gs[("paper", "rev_writes", "author"), "coo", False, (P, A)] = torch.stack([w_paper, w_author])
    gs[("paper", "cites", "paper"), "coo", False, (P, P)] = torch.stack([c_src, c_dst])
    x_author = torch.randn((A, F_IN), generator=g, device=dev)
    x_author[torch.arange(A, device=dev), author_class] += 2.0
    fs["author", "x", None] = x_author
    fs["paper", "x", None] = torch.randn((P, F_IN), generator=g, device=dev)          # no class signal of its own
    etypes = [("author", "writes", "paper"), ("paper", "cites", "paper"), ("paper", "rev_writes", "author")]
    perm = torch.randperm(P, generator=g, device=dev)
    train_ids, test_ids = perm[: P // 2], perm[P // 2: P // 2 + 8 * args.batch_size]

    def loader_over(ids, shuffle):
        return NeighborLoader((fs, gs), {et: [10, 5] for et in etypes}, input_nodes=("paper", ids), batch_size=args.batch_size,
                              shuffle=shuffle, local_seeds_per_call=args.group * args.batch_size)

    layers = torch.nn.ModuleList(
        nn.HeteroConv({et: nn.GATConv(fin, CH, heads=HEADS, add_self_loops=False) for et in etypes}) for fin in (F_IN, HEADS * CH)).to(dev)
    head = torch.nn.Linear(HEADS * CH, C).to(dev)
    opt = torch.optim.Adam(list(layers.parameters()) + list(head.parameters()), lr=0.01)
    loss = None
    for epoch in range(args.epochs):
        t0, edges = time.perf_counter(), 0
        loader = loader_over(train_ids, True)
        for grp in loader.call_groups():
            y = paper_class[grp.n_id["paper"][seed_rows(grp)]]
            out = forward(layers, head, grp)
            loss = F.cross_entropy(out, y)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            edges += grp.num_edges
        torch.cuda.synchronize()
        print("epoch %d: loss %.4f, %.2f M sampled edges/s (training, relation by relation)" % (
            epoch, float(loss.detach()), edges / (time.perf_counter() - t0) / 1e6))
    # ---- inference through the call-group route; x stays lazy ---------------------------------------------------------------
    hit = total = 0
    same_bits = True
    t0, edges = time.perf_counter(), 0
    with torch.no_grad():
        for grp in loader_over(test_ids, False).call_groups():
            y = paper_class[grp.n_id["paper"][seed_rows(grp)]]
            out = forward(layers, head, grp)
            for layer in layers:
                layer.fetch_in_layer = False            # gather x[n_id] first, then the same kernels over the copy
            same_bits &= bool(torch.equal(forward(layers, head, grp), out))
            for layer in layers:
                layer.fetch_in_layer = True
            hit += int((out.argmax(1) == y).sum())
            total += int(y.numel())
            edges += grp.num_edges
    torch.cuda.synchronize()
    acc = hit / max(total, 1)
    print("test accuracy %.3f over %d papers (chance %.3f); lazy == gathered bit for bit: %s" % (acc, total, 1.0 / C, same_bits))
    return float(loss.detach()), acc, same_bits


def seed_rows(grp):
    """Rows of the group's seeds in ``n_id['paper']``: every mini-batch's vertex list starts with its seeds."""
    ptr = grp.node_ptr["paper"].long()
    bp = grp.batch_ptr.long()
    G = grp.n_batches
    per = bp[1:G + 1] - bp[:G]
    start = torch.repeat_interleave(ptr[:G], per)
    within = torch.arange(int(bp[G]), device=ptr.device) - torch.repeat_interleave(bp[:G], per)
    return start + within


if __name__ == "__main__":
    main()
