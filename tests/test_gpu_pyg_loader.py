"""GPU: the cugraph_pyg-shaped GraphStore / FeatureStore / NeighborLoader on the HIP hot path.
Structural invariants are the reference tests' own
(/root/reference/python/cugraph-pyg/cugraph_pyg/tests/loader/test_neighbor_loader.py:20-133 and the
karate configuration of BASELINE.json configs[0]); the random part is checked bit-exactly against
the oracle composed the same way."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)


def oracle_neighbor_sample(oracle_mod, row_ptr, col, edge_id, seeds, fanout, random_state, weights=None):
    """PyG-style hop expansion (new vertices only) composed from the oracle's one-hop ops."""
    from cugraph_pyg_amd.sampler.sampler import hop_seed
    nodes = np.asarray(seeds, dtype=np.int64)
    frontier, f_start = nodes, 0
    rows, cols, edges, nn, ne = [], [], [], [len(nodes)], []
    for k, fan in enumerate(fanout):
        if len(frontier) == 0:
            nn.append(0)
            ne.append(0)
            continue
        if weights is None:
            off, nbr, lid, gid = oracle_mod.unweighted_sample(row_ptr, col, frontier, fan, hop_seed(random_state, k))
        else:
            off, nbr, lid, gid = oracle_mod.weighted_sample(row_ptr, col, weights, frontier, fan, hop_seed(random_state, k))
        new_nodes, mp = oracle_mod.append_unique(nodes, nbr.astype(np.int64))
        rows.append(mp.astype(np.int64))
        cols.append(lid.astype(np.int64) + f_start)
        edges.append(edge_id[gid])
        ne.append(len(nbr))
        nn.append(len(new_nodes) - len(nodes))
        f_start = len(nodes)
        frontier, nodes = new_nodes[f_start:], new_nodes
    z = np.zeros(0, np.int64)
    return nodes, np.concatenate(rows or [z]), np.concatenate(cols or [z]), np.concatenate(edges or [z]), nn, ne


def _karate():
    e = np.loadtxt(os.path.join(HERE, "golden", "karate.csv"), dtype=np.int64, usecols=(0, 1))
    return e[:, 0], e[:, 1]


def test_neighbor_loader_karate_e2e(oracle_mod, hiplib):
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    src, dst = _karate()
    ei = torch.stack([torch.from_numpy(dst), torch.from_numpy(src)]).cuda()     # as the reference test builds it
    graph_store = GraphStore()
    graph_store.put_edge_index(ei, ("person", "knows", "person"), "coo", False, (34, 34))
    feature_store = FeatureStore()
    feat = torch.randint(128, (34, 16), generator=torch.Generator().manual_seed(0))
    feature_store["person", "feat", None] = feat
    loader = NeighborLoader((feature_store, graph_store), [5, 5], input_nodes=torch.arange(34), batch_size=16,
                            random_state=62)
    assert len(loader) == 3
    g = graph_store._graph
    rp, col, eid = g.row_ptr.cpu().numpy(), g.col.cpu().numpy(), g.edge_id.cpu().numpy()
    n_batches = 0
    for b, batch in enumerate(loader):
        n_batches += 1
        # the reference test's assertion
        assert (feature_store["person", "feat", None][batch.n_id] == batch.feat).all()
        assert torch.equal(batch.feat.cpu(), feat[batch.n_id.cpu()])
        seeds = np.arange(34)[b * 16:(b + 1) * 16]
        assert batch.batch_size == len(seeds) and torch.equal(batch.input_id.cpu(), torch.from_numpy(seeds))
        assert np.array_equal(batch.n_id[: len(seeds)].cpu().numpy(), seeds)           # seeds first
        assert torch.equal(batch.batch, batch.n_id[: len(seeds)])
        # every sampled edge is an original edge: e_id indexes the ORIGINAL edge_index
        gsrc = batch.n_id[batch.edge_index[0]]
        gdst = batch.n_id[batch.edge_index[1]]
        assert torch.equal(ei[0][batch.e_id], gsrc) and torch.equal(ei[1][batch.e_id], gdst)
        assert int(batch.num_sampled_nodes.sum()) == batch.n_id.numel()
        assert int(batch.num_sampled_edges.sum()) == batch.edge_index.shape[1]
        assert batch.n_id.unique().numel() == batch.n_id.numel()
        # bit-exact vs the oracle composed the same way
        node, row, colv, edge, nn, ne = oracle_neighbor_sample(oracle_mod, rp, col, eid, seeds, [5, 5], 62 + b)
        assert np.array_equal(batch.n_id.cpu().numpy(), node)
        assert np.array_equal(batch.edge_index.cpu().numpy(), np.stack([row, colv]))
        assert np.array_equal(batch.e_id.cpu().numpy(), edge)
        assert batch.num_sampled_nodes.tolist() == nn and batch.num_sampled_edges.tolist() == ne
    assert n_batches == 3


def test_neighbor_loader_biased_exact_example(hiplib):
    # tests/loader/test_neighbor_loader.py:99-133 — exact output on the 3-edge graph
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    eix = torch.tensor([[3, 4, 5], [0, 1, 2]])
    graph_store = GraphStore()
    graph_store.put_edge_index(eix, ("person", "knows", "person"), "coo", False, (6, 6))
    feature_store = FeatureStore()
    feature_store["person", "feat", None] = torch.randint(128, (6, 12))
    feature_store[("person", "knows", "person"), "bias", None] = torch.tensor([0, 12, 14], dtype=torch.float32)
    loader = NeighborLoader((feature_store, graph_store), [1], input_nodes=torch.tensor([0, 1, 2], dtype=torch.int64),
                            batch_size=3, weight_attr="bias")
    out = list(iter(loader))
    assert len(out) == 1
    out = out[0]
    assert out.edge_index.shape[1] == 2
    assert (out.edge_index.cpu() == torch.tensor([[3, 4], [1, 2]])).all()
    assert out.bias.cpu().tolist() == [12.0, 14.0]          # edge attribute gathered at e_id


def test_neighbor_loader_fanout_all_and_hops(hiplib):
    # basic_pyg_graph_2 (tests/conftest.py:58-66): star around 0 and 9; fan-out -1 returns every in-edge
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    edge_index = torch.tensor([[0, 1, 0, 2, 3, 0, 4, 0, 5, 0, 6, 7, 0, 8, 9],
                               [1, 9, 2, 9, 9, 4, 9, 5, 9, 6, 9, 9, 8, 9, 0]])
    gs = GraphStore()
    gs.put_edge_index(edge_index, ("n", "e", "n"), "coo", False, (10, 10))
    fs = FeatureStore()
    fs["n", "x", None] = torch.arange(10, dtype=torch.float32).view(-1, 1).repeat(1, 4)
    loader = NeighborLoader((fs, gs), [-1, -1], input_nodes=torch.tensor([9]), batch_size=1, random_state=1)
    (b,) = list(loader)
    n_id = b.n_id.cpu()
    assert n_id[0] == 9 and sorted(n_id[1:].tolist()) == [0, 1, 2, 3, 4, 5, 6, 7, 8]  # hop 2 reaches 0 (0 -> 1,2,4,5,6,8)
    src = n_id[b.edge_index[0].cpu()]
    dst = n_id[b.edge_index[1].cpu()]
    hop1 = int(b.num_sampled_edges[0])
    # node 9 has 8 in-edges (from 1..8): fan-out -1 returns all of them
    assert hop1 == 8 and (dst[:hop1] == 9).all() and sorted(src[:hop1].tolist()) == [1, 2, 3, 4, 5, 6, 7, 8]
    # hop 2 expands only the NEW vertices (seed 9 is excluded as a prior source)
    assert not (dst[hop1:] == 9).any()
    pairs = set(zip(edge_index[0].tolist(), edge_index[1].tolist()))
    assert all((int(s), int(d)) in pairs for s, d in zip(src, dst))
    assert torch.equal(b.x.cpu(), n_id.float().view(-1, 1).repeat(1, 4))
    assert b.num_sampled_nodes.tolist()[0] == 1 and int(b.num_sampled_nodes.sum()) == n_id.numel()


def test_neighbor_loader_powerlaw_batches_vs_oracle(oracle_mod, hiplib):
    import torch
    from graphgen import powerlaw_csr
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    V = 5000
    rp, col = powerlaw_csr(V, 12, seed=21, max_deg=500)
    dst = np.repeat(np.arange(V), np.diff(rp))                     # CSR row = message target
    ei = torch.stack([torch.from_numpy(col), torch.from_numpy(dst)])
    gs = GraphStore()
    gs.put_edge_index(ei, ("paper", "cites", "paper"), "coo", False, (V, V))
    fs = FeatureStore()
    x = torch.randn(V, 100)
    fs["paper", "x", None] = x
    fs["paper", "y", None] = torch.arange(V)
    seeds = torch.randperm(V, generator=torch.Generator().manual_seed(3))[:700]
    loader = NeighborLoader((fs, gs), [25, 10], input_nodes=seeds, batch_size=256, random_state=1000, shuffle=False)
    g = gs._graph
    grp, gcol, geid = g.row_ptr.cpu().numpy(), g.col.cpu().numpy(), g.edge_id.cpu().numpy()
    for b, batch in enumerate(loader):
        s = seeds[b * 256:(b + 1) * 256].numpy()
        node, row, colv, edge, nn, ne = oracle_neighbor_sample(oracle_mod, grp, gcol, geid, s, [25, 10], 1000 + b)
        assert np.array_equal(batch.n_id.cpu().numpy(), node)
        assert np.array_equal(batch.edge_index.cpu().numpy(), np.stack([row, colv]))
        assert np.array_equal(batch.e_id.cpu().numpy(), edge)
        assert torch.equal(batch.x.cpu(), x[batch.n_id.cpu()]) and torch.equal(batch.y.cpu(), batch.n_id.cpu())
        assert batch.num_sampled_nodes.tolist() == nn and batch.num_sampled_edges.tolist() == ne
        # a GraphSAGE layer runs straight on the batch
        from wholegraph_amd import nn as wnn
        conv = wnn.SAGEConv(100, 32).cuda()
        out = conv(batch.x, batch.edge_index)
        assert out.shape == (batch.n_id.numel(), 32) and torch.isfinite(out).all()
