#!/usr/bin/env python
"""End-to-end mini-batch GraphSAGE training on a synthetic power-law graph with the cugraph_pyg-compatible stack of this
repo: GraphStore / FeatureStore -> NeighborLoader (call-group sampling on the HIP kernels) -> SAGEConv (HIP aggregation)
-> loss / backward / optimizer step in plain PyTorch-ROCm.  It is the shape of the reference's single-GPU examples
(python/cugraph-pyg/cugraph_pyg/examples/gcn_dist_sg.py in upstream releases: `NeighborLoader((feature_store, graph_store),
num_neighbors, input_nodes, batch_size)` feeding a PyG model); only the imports change.

    python examples/sage_node_classification.py [--nodes 200000] [--epochs 2]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from cugraph_pyg_amd.data import FeatureStore, GraphStore  # noqa: E402   (reference: cugraph_pyg.data)
from cugraph_pyg_amd.loader import NeighborLoader  # noqa: E402           (reference: cugraph_pyg.loader)
from wholegraph_amd.nn import SAGEConv  # noqa: E402                      (reference: torch_geometric.nn.SAGEConv)


class SAGE(torch.nn.Module):
    def __init__(self, in_dim, hidden, classes, layers):
        super().__init__()
        dims = [in_dim] + [hidden] * (layers - 1) + [classes]
        self.convs = torch.nn.ModuleList(SAGEConv(dims[i], dims[i + 1]) for i in range(layers))

    def forward(self, x, edge_index, num_sampled_nodes, num_sampled_edges):
        # trim the subgraph hop by hop like torch_geometric.utils.trim_to_layer: layer l only needs the edges of hops
        # 0 .. L-1-l and produces the vertices those edges point at
        n_nodes, n_edges = [int(v) for v in num_sampled_nodes], [int(v) for v in num_sampled_edges]
        for l, conv in enumerate(self.convs):
            hops = len(self.convs) - l
            e_keep, n_dst = sum(n_edges[:hops]), sum(n_nodes[:hops])
            x = conv((x, x[:n_dst]), edge_index[:, :e_keep])
            if l + 1 < len(self.convs):
                x = F.relu(x)
        return x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=200_000)
    ap.add_argument("--avg-degree", type=int, default=20)
    ap.add_argument("--features", type=int, default=100)
    ap.add_argument("--classes", type=int, default=16)
    ap.add_argument("--batch-size", type=int, default=1024)
    ap.add_argument("--fanout", type=int, nargs="+", default=[25, 10])
    ap.add_argument("--epochs", type=int, default=2)
    args = ap.parse_args()
    assert torch.cuda.is_available(), "needs an MI355X (there is no CPU fallback)"
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    V, E = args.nodes, args.nodes * args.avg_degree
    # power-law-ish endpoints; labels depend on a hidden community id that the features encode noisily
    src = (torch.rand(E, generator=g, device=dev) ** 2 * V).long().clamp_(max=V - 1)
    dst = torch.randint(0, V, (E,), generator=g, device=dev)
    community = torch.arange(V, device=dev) % args.classes
    same = torch.rand(E, generator=g, device=dev) < 0.7        # homophily: most edges stay inside a community
    peer = (torch.randint(0, max(V // args.classes, 1), (E,), generator=g, device=dev) * args.classes + community[src]).clamp_(max=V - 1)
    dst = torch.where(same, peer, dst)
    x = torch.randn((V, args.features), generator=g, device=dev)
    x[torch.arange(V, device=dev), community % args.features] += 1.0   # weak per-node signal, strong after aggregation
    graph_store, feature_store = GraphStore(), FeatureStore()
    graph_store[("node", "to", "node"), "coo", False, (V, V)] = torch.stack([src, dst])
    feature_store["node", "x", None] = x
    feature_store["node", "y", None] = community
    train_ids = torch.randperm(V, generator=g, device=dev)[: V // 2]
    loader = NeighborLoader((feature_store, graph_store), num_neighbors=args.fanout, input_nodes=train_ids,
                            batch_size=args.batch_size, shuffle=True, local_seeds_per_call=16 * args.batch_size)
    model = SAGE(args.features, 128, args.classes, len(args.fanout)).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    for epoch in range(args.epochs):
        t0, total, correct, seen, edges = time.perf_counter(), 0.0, 0, 0, 0
        for batch in loader:
            out = model(batch.x, batch.edge_index, batch.num_sampled_nodes, batch.num_sampled_edges)[: batch.batch_size]
            y = batch.y[: batch.batch_size]
            loss = F.cross_entropy(out, y)
            opt.zero_grad()
            loss.backward()
            opt.step()
            total += float(loss.detach()) * batch.batch_size
            correct += int((out.argmax(1) == y).sum())
            seen += batch.batch_size
            edges += batch.edge_index.shape[1]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"epoch {epoch}: loss {total / seen:.4f}  train acc {correct / seen:.3f}  {edges / dt / 1e6:.1f} M sampled edges/s "
              f"(sampling + feature fetch + forward + backward + Adam), {dt:.2f} s")
    return total / seen, correct / seen


if __name__ == "__main__":
    main()
