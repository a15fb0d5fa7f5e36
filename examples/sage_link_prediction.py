#!/usr/bin/env python
"""Mini-batch link prediction with the cugraph_pyg-compatible stack of this repo: GraphStore / FeatureStore ->
LinkNeighborLoader (seed edges + binary negatives, call-group sampling on the HIP kernels) -> 2-layer GraphSAGE encoder
(HIP aggregation) -> dot-product decoder -> BCE loss, plain PyTorch-ROCm training loop.  The shape of the reference's
link-prediction examples (python/cugraph-pyg/cugraph_pyg/examples/mag_lp_mnmg.py, movielens_mnmg.py:
`LinkNeighborLoader((feature_store, graph_store), num_neighbors, edge_label_index, neg_sampling, batch_size)` feeding an
encoder / decoder pair); only the imports change.

    python examples/sage_link_prediction.py [--nodes 20000] [--epochs 3]
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
from cugraph_pyg_amd.loader import LinkNeighborLoader  # noqa: E402       (reference: cugraph_pyg.loader)
from wholegraph_amd.nn import SAGEConv  # noqa: E402                      (reference: torch_geometric.nn.SAGEConv)


class Encoder(torch.nn.Module):
    def __init__(self, in_dim, hidden):
        super().__init__()
        self.conv1, self.conv2 = SAGEConv(in_dim, hidden), SAGEConv(hidden, hidden)

    def forward(self, x, edge_index):
        # full-subgraph message passing (every sampled vertex keeps an embedding: the decoder indexes any of them)
        h = F.relu(self.conv1(x, edge_index))
        return self.conv2(h, edge_index)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=20_000)
    ap.add_argument("--avg-degree", type=int, default=12)
    ap.add_argument("--communities", type=int, default=20)
    ap.add_argument("--batch-size", type=int, default=512)
    ap.add_argument("--fanout", type=int, nargs="+", default=[10, 5])
    ap.add_argument("--epochs", type=int, default=3)
    args = ap.parse_args()
    assert torch.cuda.is_available(), "needs an MI355X (there is no CPU fallback)"
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    V, E, C = args.nodes, args.nodes * args.avg_degree, args.communities
    # planted partition: 90 % of the edges stay inside a community, features are a noisy one-hot of the community
    community = torch.arange(V, device=dev) % C
    src = torch.randint(0, V, (E,), generator=g, device=dev)
    inside = torch.rand(E, generator=g, device=dev) < 0.9
    peer = (torch.randint(0, max(V // C, 1), (E,), generator=g, device=dev) * C + community[src]).clamp_(max=V - 1)
    dst = torch.where(inside, peer, torch.randint(0, V, (E,), generator=g, device=dev))
    x = 0.5 * torch.randn((V, 32), generator=g, device=dev)
    x[torch.arange(V, device=dev), community % 32] += 1.0
    graph_store, feature_store = GraphStore(), FeatureStore()
    graph_store[("node", "to", "node"), "coo", False, (V, V)] = torch.stack([src, dst])
    feature_store["node", "x", None] = x
    train_edges = torch.randperm(E, generator=g, device=dev)[: E // 4]
    loader = LinkNeighborLoader((feature_store, graph_store), num_neighbors=args.fanout,
                                edge_label_index=torch.stack([src[train_edges], dst[train_edges]]), batch_size=args.batch_size,
                                neg_sampling=("binary", 1.0), shuffle=True, local_seeds_per_call=16 * args.batch_size)
    model = Encoder(32, 64).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    loss_avg = acc = 0.0
    for epoch in range(args.epochs):
        t0, total, correct, seen, edges = time.perf_counter(), 0.0, 0, 0, 0
        for batch in loader:
            h = model(batch.x, batch.edge_index)
            eli, label = batch.edge_label_index, (batch.edge_label > 0).float()
            score = (h[eli[0]] * h[eli[1]]).sum(-1)
            loss = F.binary_cross_entropy_with_logits(score, label)
            opt.zero_grad()
            loss.backward()
            opt.step()
            total += float(loss.detach()) * label.numel()
            correct += int(((score > 0).float() == label).sum())
            seen += label.numel()
            edges += batch.edge_index.shape[1]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        loss_avg, acc = total / seen, correct / seen
        print(f"epoch {epoch}: loss {loss_avg:.4f}  link accuracy {acc:.3f}  {edges / dt / 1e6:.1f} M sampled edges/s "
              f"(negatives + sampling + feature fetch + forward + backward + Adam), {dt:.2f} s")
    return loss_avg, acc


if __name__ == "__main__":
    main()
