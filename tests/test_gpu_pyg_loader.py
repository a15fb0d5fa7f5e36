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


@pytest.mark.parametrize("location", ["cuda", "cpu"])   # "cpu": the reference's default, rows in pinned host memory
def test_neighbor_loader_karate_e2e(oracle_mod, hiplib, location):
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    src, dst = _karate()
    ei = torch.stack([torch.from_numpy(dst), torch.from_numpy(src)]).cuda()     # as the reference test builds it
    graph_store = GraphStore()
    graph_store.put_edge_index(ei, ("person", "knows", "person"), "coo", False, (34, 34))
    feature_store = FeatureStore(location=location)
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


# ------------------------------------------------------------------------------- heterogeneous
def oracle_hetero_sample(oracle_mod, graphs, seed_type, seeds, fanout, random_state):
    """Same composition as cugraph_pyg_amd.sampler.hetero_neighbor_sample, on the oracle."""
    from cugraph_pyg_amd.sampler.sampler import hop_seed
    etypes = sorted(graphs.keys())
    ntypes = sorted({t for et in etypes for t in (et[0], et[2])} | {seed_type})
    node = {t: np.zeros(0, np.int64) for t in ntypes}
    node[seed_type] = np.asarray(seeds, np.int64)
    fstart = {t: 0 for t in ntypes}
    n_hops = len(next(iter(fanout.values())))
    rows, cols, edges = {et: [] for et in etypes}, {et: [] for et in etypes}, {et: [] for et in etypes}
    for h in range(n_hops):
        begin = {t: len(node[t]) for t in ntypes}
        for ti, et in enumerate(etypes):
            src_t, _, dst_t = et
            fan = fanout.get(et, [0] * n_hops)[h]
            frontier = node[dst_t][fstart[dst_t]:begin[dst_t]]
            if fan == 0 or len(frontier) == 0:
                continue
            rp, col, eid = graphs[et]
            off, nbr, lid, gid = oracle_mod.unweighted_sample(rp, col, frontier, fan, hop_seed(random_state, h * len(etypes) + ti))
            node[src_t], mp = oracle_mod.append_unique(node[src_t], nbr.astype(np.int64))
            rows[et].append(mp.astype(np.int64))
            cols[et].append(lid.astype(np.int64) + fstart[dst_t])
            edges[et].append(eid[gid])
        for t in ntypes:
            fstart[t] = begin[t]
    z = np.zeros(0, np.int64)
    c = lambda xs: np.concatenate(xs) if xs else z  # noqa: E731
    return node, {et: c(rows[et]) for et in etypes}, {et: c(cols[et]) for et in etypes}, {et: c(edges[et]) for et in etypes}


def test_neighbor_loader_hetero_basic_reference_example(hiplib):
    # tests/loader/test_neighbor_loader.py:355-409, same assertions
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    src = torch.tensor([0, 1, 2, 4, 3, 4, 5, 5])
    dst = torch.tensor([4, 5, 4, 3, 2, 1, 0, 1])
    asrc = torch.tensor([0, 1, 2, 3, 3, 0])
    adst = torch.tensor([0, 1, 2, 3, 4, 5])
    graph_store, feature_store = GraphStore(), FeatureStore()
    graph_store[("paper", "cites", "paper"), "coo", False, (6, 6)] = [src, dst]
    graph_store[("author", "writes", "paper"), "coo", False, (4, 6)] = [asrc, adst]
    feature_store["paper", "x", None] = torch.arange(6, dtype=torch.float32).view(-1, 1).repeat(1, 8)
    loader = NeighborLoader((feature_store, graph_store),
                            num_neighbors={("paper", "cites", "paper"): [1, 1], ("author", "writes", "paper"): [1, 1]},
                            input_nodes=("paper", torch.tensor([0, 1])), batch_size=2)
    out = next(iter(loader))
    ei_out = out["paper"].n_id.cpu()[out["paper", "cites", "paper"].edge_index.cpu()]
    assert (src[out["paper", "cites", "paper"].e_id.cpu()] == ei_out[0]).all()
    assert (dst[out["paper", "cites", "paper"].e_id.cpu()] == ei_out[1]).all()
    ej_out = torch.stack([out["author"].n_id.cpu()[out["author", "writes", "paper"].edge_index[0].cpu()],
                          out["paper"].n_id.cpu()[out["author", "writes", "paper"].edge_index[1].cpu()]])
    assert (asrc[out["author", "writes", "paper"].e_id.cpu()] == ej_out[0]).all()
    assert (adst[out["author", "writes", "paper"].e_id.cpu()] == ej_out[1]).all()
    assert out["paper"].n_id[:2].tolist() == [0, 1] and out["paper"].batch_size == 2
    assert torch.equal(out["paper"].x.cpu()[:, 0], out["paper"].n_id.cpu().float())


def test_hetero_mag_like_sampling_vs_oracle_and_gat(oracle_mod, hiplib):
    """BASELINE configs[4] shape at test scale: 4 node types, 2-hop per-edge-type fan-out, then the GAT
    edge-softmax aggregation of one relation on the sampled bipartite subgraph vs the oracle."""
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    from wholegraph_amd import nn as wnn
    rng = np.random.default_rng(0)
    n = {"paper": 3000, "author": 4000, "institution": 100, "field": 300}
    rel = {("author", "writes", "paper"): 9000, ("paper", "cites", "paper"): 12000,
           ("paper", "has_topic", "field"): 7000, ("author", "affiliated_with", "institution"): 4000,
           ("paper", "rev_writes", "author"): 9000, ("field", "rev_has_topic", "paper"): 7000}
    gs, fs = GraphStore(), FeatureStore()
    coo = {}
    for (s, r, d), m in rel.items():
        ei = torch.from_numpy(np.stack([rng.integers(0, n[s], m), rng.integers(0, n[d], m)]))
        coo[(s, r, d)] = ei
        gs[(s, r, d), "coo", False, (n[s], n[d])] = ei
    xp = torch.randn(n["paper"], 128)
    fs["paper", "x", None] = xp
    fs["author", "x", None] = torch.randn(n["author"], 128)
    fanout = {et: [5, 3] for et in rel}
    seeds = torch.from_numpy(rng.permutation(n["paper"])[:200])
    loader = NeighborLoader((fs, gs), fanout, input_nodes=("paper", seeds), batch_size=128, random_state=5)
    hg = gs._hetero_graphs
    graphs_np = {et: (g.row_ptr.cpu().numpy(), g.col.cpu().numpy(), g.edge_id.cpu().numpy()) for et, g in hg.items()}
    nb = 0
    for b, batch in enumerate(loader):
        nb += 1
        s = seeds[b * 128:(b + 1) * 128].numpy()
        node, row, col, edge = oracle_hetero_sample(oracle_mod, graphs_np, "paper", s, fanout, 5 + b)
        for t in n:
            assert np.array_equal(batch[t].n_id.cpu().numpy(), node[t]), t
        for et in rel:
            assert np.array_equal(batch[et].edge_index.cpu().numpy(), np.stack([row[et], col[et]])), et
            assert np.array_equal(batch[et].e_id.cpu().numpy(), edge[et]), et
            # e_id indexes the ORIGINAL per-type edge list (hetero edge-id rule of the reference test)
            g_src = batch[et[0]].n_id.cpu()[batch[et].edge_index[0].cpu()]
            g_dst = batch[et[2]].n_id.cpu()[batch[et].edge_index[1].cpu()]
            assert torch.equal(coo[et][0][batch[et].e_id.cpu()], g_src) and torch.equal(coo[et][1][batch[et].e_id.cpu()], g_dst)
        assert torch.equal(batch["paper"].x.cpu(), xp[batch["paper"].n_id.cpu()])
        assert int(batch["paper"].num_sampled_nodes.sum()) == batch["paper"].n_id.numel()
        # GAT over (author -writes-> paper) on the sampled bipartite subgraph, 4 heads
        et = ("author", "writes", "paper")
        ei = batch[et].edge_index
        x_src, x_dst = batch["author"].x, batch["paper"].x
        gat = wnn.GATConv(128, 32, heads=4, add_self_loops=False).cuda()
        out = gat((x_src, x_dst), ei)
        assert out.shape == (x_dst.shape[0], 128)
        # oracle on the same CSR
        rp, cc = wnn._to_csr(ei, x_dst.shape[0])
        h_src, h_dst = gat.lin(x_src), gat.lin(x_dst)
        a_s = (h_src.view(-1, 4, 32) * gat.att_src).sum(-1)
        a_d = (h_dst.view(-1, 4, 32) * gat.att_dst).sum(-1)
        oref, _ = oracle_mod.gat_csr(rp.cpu().numpy(), cc.cpu().numpy(), h_src.detach().cpu().numpy().reshape(-1, 4, 32),
                                     a_s.detach().cpu().numpy(), a_d.detach().cpu().numpy(), 0.2)
        np.testing.assert_allclose((out - gat.bias).detach().cpu().numpy().reshape(-1, 4, 32), oref, rtol=1e-4, atol=1e-5)
    assert nb == 2


# ------------------------------------------------------------------------------- link loaders
def _link_fixture():
    import torch
    from graphgen import powerlaw_csr
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    V = 3000
    rp, col = powerlaw_csr(V, 10, seed=31, max_deg=300)
    dst = np.repeat(np.arange(V), np.diff(rp))
    ei = torch.stack([torch.from_numpy(col), torch.from_numpy(dst)])
    gs = GraphStore()
    gs.put_edge_index(ei, ("n", "e", "n"), "coo", False, (V, V))
    fs = FeatureStore()
    x = torch.randn(V, 16)
    fs["n", "x", None] = x
    return V, ei, gs, fs, x


@pytest.mark.parametrize("neg", [None, "binary", ("binary", 2.0), "triplet"])
def test_link_neighbor_loader(hiplib, neg):
    """Structure of the reference's link tests (tests/loader/test_neighbor_loader.py:138-350): the label
    index points at the seed edges' endpoints inside n_id; positives first; negatives labelled 0."""
    import torch
    from cugraph_pyg_amd.loader import LinkNeighborLoader
    V, ei, gs, fs, x = _link_fixture()
    seeds = ei[:, torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(1))[:333]]
    loader = LinkNeighborLoader((fs, gs), [5, 3], edge_label_index=seeds, batch_size=100, neg_sampling=neg,
                                random_state=9)
    assert len(loader) == 4
    seen = 0
    for b, batch in enumerate(loader):
        n_pos = min(100, 333 - b * 100)
        n_id = batch.n_id.cpu()
        assert batch.batch_size == n_pos and torch.equal(batch.input_id.cpu(), torch.arange(b * 100, b * 100 + n_pos))
        assert torch.equal(batch.x.cpu(), x[n_id]) and n_id.unique().numel() == n_id.numel()
        pos = seeds[:, b * 100:b * 100 + n_pos]
        if neg == "triplet":
            assert torch.equal(n_id[batch.src_index.cpu()], pos[0]) and torch.equal(n_id[batch.dst_pos_index.cpu()], pos[1])
            assert batch.dst_neg_index.numel() == n_pos and int(batch.dst_neg_index.max()) < n_id.numel()
        else:
            eli = batch.edge_label_index.cpu()
            assert torch.equal(n_id[eli[0, :n_pos]], pos[0]) and torch.equal(n_id[eli[1, :n_pos]], pos[1])
            if neg is None:
                assert eli.shape[1] == n_pos
            else:
                amount = 2.0 if isinstance(neg, tuple) else 1.0
                assert eli.shape[1] == n_pos + int(round(n_pos * amount))
                assert batch.edge_label.cpu().tolist() == [1.0] * n_pos + [0.0] * int(round(n_pos * amount))
        # the sampled subgraph is a valid neighbourhood: every edge exists, seeds' endpoints come first
        gsrc, gdst = n_id[batch.edge_index[0].cpu()], n_id[batch.edge_index[1].cpu()]
        assert torch.equal(ei[0][batch.e_id.cpu()], gsrc) and torch.equal(ei[1][batch.e_id.cpu()], gdst)
        seen += n_pos
    assert seen == 333


def test_link_loader_labels_and_determinism(hiplib):
    import torch
    from cugraph_pyg_amd.loader import LinkNeighborLoader
    V, ei, gs, fs, x = _link_fixture()
    seeds = ei[:, :64]
    labels = torch.arange(64) % 3
    a = [b for b in LinkNeighborLoader((fs, gs), [4], edge_label_index=seeds, edge_label=labels, batch_size=32,
                                       neg_sampling="binary", random_state=3)]
    b2 = [b for b in LinkNeighborLoader((fs, gs), [4], edge_label_index=seeds, edge_label=labels, batch_size=32,
                                        neg_sampling="binary", random_state=3)]
    for u, v in zip(a, b2):
        assert torch.equal(u.n_id, v.n_id) and torch.equal(u.edge_label_index, v.edge_label_index)
    assert a[0].edge_label.cpu()[:32].tolist() == (labels[:32] + 1).tolist()      # positives shifted by one, negatives 0
    plain = next(iter(LinkNeighborLoader((fs, gs), [4], edge_label_index=seeds, edge_label=labels, batch_size=32)))
    assert plain.edge_label.cpu().tolist() == labels[:32].tolist()


def _paper_author_stores():
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    src = torch.tensor([0, 1, 2, 4, 3, 4, 5, 5])
    dst = torch.tensor([4, 5, 4, 3, 2, 1, 0, 1])
    asrc = torch.tensor([0, 1, 2, 3, 3, 0])
    adst = torch.tensor([0, 1, 2, 3, 4, 5])
    graph_store, feature_store = GraphStore(), FeatureStore()
    graph_store[("paper", "cites", "paper"), "coo", False, (6, 6)] = [src, dst]
    graph_store[("author", "writes", "paper"), "coo", False, (4, 6)] = [asrc, adst]
    return feature_store, graph_store, asrc, adst


def test_link_neighbor_loader_hetero_linkpred_reference_example(hiplib):
    # tests/loader/test_neighbor_loader.py:455-523: exact outputs (fan-out >= degree => deterministic)
    import torch
    from cugraph_pyg_amd.loader import LinkNeighborLoader
    feature_store, graph_store, asrc, adst = _paper_author_stores()
    loader = LinkNeighborLoader((feature_store, graph_store),
                                num_neighbors={("paper", "cites", "paper"): [2, 2], ("author", "writes", "paper"): [2, 2]},
                                edge_label_index=(("author", "writes", "paper"), torch.stack([asrc, adst])), batch_size=5)
    out = next(iter(loader))
    assert out["paper"].n_id.tolist() == [0, 1, 2, 3, 4, 5]
    assert out["author"].n_id.tolist() == [0, 1, 2, 3]
    assert out["paper"].num_sampled_nodes.tolist() == [5, 1, 0]
    assert out["author"].num_sampled_nodes.tolist() == [4, 0, 0]
    assert out["paper", "cites", "paper"].edge_index.shape == torch.Size([2, 8])
    assert out["paper", "cites", "paper"].num_sampled_edges.tolist() == [7, 1]
    assert "edge_label_index" not in out["paper", "cites", "paper"]
    assert out["author", "writes", "paper"].edge_index.shape == torch.Size([2, 6])
    assert out["author", "writes", "paper"].num_sampled_edges.tolist() == [5, 1]
    eli = out["author", "writes", "paper"].edge_label_index
    assert list(eli.shape) == [2, 5]
    assert eli.tolist()[0] == [0, 1, 2, 3, 3] and eli.tolist()[1] == [0, 1, 2, 3, 4]
    assert len(loader) == 2


def test_link_neighbor_loader_hetero_bidirectional(hiplib):
    # tests/loader/test_neighbor_loader.py:529-583: nonexistent seed edges, two edge types that mirror each other
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import LinkNeighborLoader
    src = torch.tensor([1, 5, 5, 8, 1, 1])
    dst = torch.tensor([4, 2, 3, 1, 0, 4])
    feature_store, graph_store = FeatureStore(), GraphStore()
    graph_store[("user", "to", "merchant"), "coo", False, (9, 5)] = torch.stack([src, dst])
    graph_store[("merchant", "rev_to", "user"), "coo", False, (5, 9)] = torch.stack([dst, src])
    eli = torch.tensor([[0, 5, 8, 1, 7, 2], [4, 4, 2, 3, 1, 0]])
    loader = LinkNeighborLoader(data=(feature_store, graph_store),
                                num_neighbors={("user", "to", "merchant"): [2, 2], ("merchant", "rev_to", "user"): [2, 2]},
                                edge_label_index=(("user", "to", "merchant"), eli), edge_label=None, batch_size=2,
                                shuffle=False)
    n = 0
    for i, batch in enumerate(loader):
        li = batch["user", "to", "merchant"].edge_label_index.cpu()
        r_i = torch.stack([batch["user"].n_id.cpu()[li[0]], batch["merchant"].n_id.cpu()[li[1]]])
        assert (r_i == eli[:, i * 2:(i + 1) * 2]).all()
        # sampled edges are real edges of their type
        ei = batch["user", "to", "merchant"].edge_index.cpu()
        e = batch["user", "to", "merchant"].e_id.cpu()
        assert (src[e] == batch["user"].n_id.cpu()[ei[0]]).all() and (dst[e] == batch["merchant"].n_id.cpu()[ei[1]]).all()
        n += 1
    assert n == 3


@pytest.mark.parametrize("batch_size", [1, 3])
@pytest.mark.parametrize("mode,amount", [("binary", 1), ("binary", 2), ("triplet", 1), ("triplet", 3)])
def test_link_neighbor_loader_hetero_negative_sampling(hiplib, batch_size, mode, amount):
    # tests/loader/test_neighbor_loader.py:742-835, same assertions
    import torch
    from cugraph_pyg_amd.loader import LinkNeighborLoader
    feature_store, graph_store, asrc, adst = _paper_author_stores()
    et = ("author", "writes", "paper")
    loader = LinkNeighborLoader((feature_store, graph_store),
                                num_neighbors={("paper", "cites", "paper"): [2, 2], et: [2, 2]},
                                edge_label_index=(et, torch.stack([asrc, adst])), batch_size=batch_size,
                                neg_sampling=(mode, float(amount)), shuffle=False)
    seen = 0
    for batch in loader:
        assert [et] == list(batch.edge_label_index_dict.keys())
        assert [et] == list(batch.edge_label_dict.keys())
        labels = batch[et].edge_label
        assert torch.any(labels == 1.0) and torch.any(labels == 0.0)
        assert (labels == 0.0).sum() == amount * (labels == 1.0).sum()
        eli = batch[et].edge_label_index
        assert eli.shape[0] == 2 and eli.shape[1] == labels.shape[0] > 0
        assert int(eli[0].max()) < batch["author"].n_id.numel() and int(eli[1].max()) < batch["paper"].n_id.numel()
        n_pos = int((labels == 1.0).sum())
        pos = torch.stack([batch["author"].n_id[eli[0, :n_pos]], batch["paper"].n_id[eli[1, :n_pos]]]).cpu()
        assert torch.equal(pos, torch.stack([asrc, adst])[:, seen:seen + n_pos])
        seen += n_pos
    assert seen == asrc.numel()


def test_neighbor_loader_disjoint_reference_example(hiplib):
    # tests/loader/test_neighbor_loader.py:840-885: two seeds sharing their only neighbour
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    graph_store, feature_store = GraphStore(), FeatureStore()
    graph_store.put_edge_index(torch.stack([torch.tensor([2, 2]), torch.tensor([0, 1])]), ("node", "connects", "node"),
                               "coo", False, (3, 3))
    feature_store["node", "feat", None] = torch.randint(128, (3, 8))
    kw = dict(input_nodes=torch.tensor([0, 1]), batch_size=2)
    batch_nd = next(iter(NeighborLoader((feature_store, graph_store), [1], disjoint=False, **kw)))
    assert batch_nd.e_id.numel() == 2
    batch_d = next(iter(NeighborLoader((feature_store, graph_store), [1], disjoint=True, **kw)))
    assert batch_d.e_id.numel() == 1
    assert batch_d.input_id.min() >= 0 and batch_d.input_id.max() == batch_d.batch_size - 1
    assert sorted(batch_d.input_id.tolist()) == [0, 1]
    assert sorted(batch_d.n_id.tolist()) == [0, 1, 2]


@pytest.mark.parametrize("batch_size", [1, 2, 4, 8, 16])
def test_neighbor_loader_disjoint_batch_structure(hiplib, batch_size):
    # tests/loader/test_neighbor_loader.py:888-943 on karate: the per-seed trees of a batch never share a vertex
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    src, dst = (torch.from_numpy(a) for a in _karate())
    graph_store, feature_store = GraphStore(), FeatureStore()
    graph_store.put_edge_index(torch.stack([dst, src]), ("person", "knows", "person"), "coo", False, (34, 34))
    feature_store["person", "feat", None] = torch.randint(128, (34, 16))
    loader = NeighborLoader((feature_store, graph_store), [5, 5], input_nodes=torch.arange(34), batch_size=batch_size,
                            disjoint=True)
    n_batches = 0
    for batch in loader:
        ei = batch.edge_index.cpu()
        trees = {}
        for n_id in range(int(batch.num_sampled_nodes[0])):
            trees[n_id] = {n_id}
            off = 0
            for hop in range(len(batch.num_sampled_edges)):
                cnt = int(batch.num_sampled_edges[hop])
                e_h = ei[:, off:off + cnt]
                e_in = torch.isin(e_h[1], torch.tensor(sorted(trees[n_id])))
                trees[n_id].update(e_h[0][e_in].tolist())
                off += cnt
        sets = list(trees.values())
        for i in range(len(sets)):
            for j in range(i + 1, len(sets)):
                assert not (sets[i] & sets[j])
        # every non-seed vertex belongs to some tree, and sampled edges are real edges
        assert set().union(*sets) == set(range(batch.n_id.numel()))
        g = batch.n_id.cpu()[ei]
        e = batch.e_id.cpu()
        assert (dst[e] == g[0]).all() and (src[e] == g[1]).all()
        n_batches += 1
    assert n_batches == (34 + batch_size - 1) // batch_size


def test_link_neighbor_loader_disjoint_reference_example(hiplib):
    # tests/loader/test_neighbor_loader.py:138-190
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import LinkNeighborLoader
    graph_store, feature_store = GraphStore(), FeatureStore()
    graph_store[("node", "connects", "node"), "coo", False, (5, 5)] = torch.stack(
        [torch.tensor([4, 4, 4, 4]), torch.tensor([0, 1, 2, 3])])
    eli = torch.tensor([[0, 2], [1, 3]])
    kw = dict(num_neighbors=[1], edge_label_index=eli, batch_size=2, shuffle=False)
    assert next(iter(LinkNeighborLoader((feature_store, graph_store), disjoint=False, **kw))).e_id.numel() == 4
    batch_d = next(iter(LinkNeighborLoader((feature_store, graph_store), disjoint=True, **kw)))
    assert batch_d.e_id.numel() == 1
    li = batch_d.edge_label_index
    assert tuple(li.shape) == (2, 2) and li.min() >= 0 and li.max() < batch_d.n_id.numel()
    assert batch_d.n_id[li[0]].cpu().tolist() == [0, 2] and batch_d.n_id[li[1]].cpu().tolist() == [1, 3]


def _temporal_stores(hetero):
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    src_cite, dst_cite, tme_cite = torch.tensor([3, 2, 1, 2]), torch.tensor([2, 1, 0, 0]), torch.tensor([0, 1, 2, 0])
    graph_store, feature_store = GraphStore(), FeatureStore()
    graph_store[("paper", "cites", "paper"), "coo", False, (4, 4)] = [dst_cite, src_cite]
    feature_store[("paper", "cites", "paper"), "time", None] = tme_cite
    feature_store[("paper", "cites", "paper"), "bias", None] = torch.ones(4, device="cuda")
    if hetero:
        src_author = torch.tensor([3, 2, 2, 1, 3, 2, 0])
        dst_author = torch.tensor([0, 0, 1, 1, 2, 2, 2])
        graph_store[("author", "writes", "paper"), "coo", False, (3, 4)] = [dst_author, src_author]
        feature_store[("author", "writes", "paper"), "time", None] = torch.tensor([0, 0, 1, 0, 2, 1, 1])
        feature_store[("author", "writes", "paper"), "bias", None] = torch.ones(7, device="cuda")
    return feature_store, graph_store


@pytest.mark.parametrize("biased", [True, False])
def test_neighbor_loader_temporal_simple(hiplib, biased):
    # tests/loader/test_neighbor_loader.py:946-990: exact outputs of a strictly-increasing temporal walk
    import torch
    from cugraph_pyg_amd.loader import NeighborLoader
    feature_store, graph_store = _temporal_stores(False)
    loader = NeighborLoader((feature_store, graph_store), num_neighbors=[2, 2, 2], batch_size=1, input_nodes=torch.tensor([3]),
                            input_time=torch.tensor([-1]), time_attr="time", shuffle=False,
                            weight_attr="bias" if biased else None, temporal_comparison="strictly_increasing")
    out = next(iter(loader))
    assert out.n_id.tolist() == [3, 2, 1, 0]
    assert out.e_id.tolist() == [0, 1, 2]
    assert out.num_sampled_nodes.tolist() == [1, 1, 1, 1]
    assert out.num_sampled_edges.tolist() == [1, 1, 1]
    # the default comparison (monotonically_decreasing) from time 0 only follows edges with t <= previous
    loader = NeighborLoader((feature_store, graph_store), num_neighbors=[2, 2, 2], batch_size=1, input_nodes=torch.tensor([3]),
                            input_time=torch.tensor([0]), time_attr="time", shuffle=False)
    out = next(iter(loader))
    assert out.n_id.tolist() == [3, 2, 0] and out.e_id.tolist() == [0, 3]


@pytest.mark.parametrize("biased", [True, False])
def test_neighbor_loader_temporal_hetero(hiplib, biased):
    # tests/loader/test_neighbor_loader.py:993-1057
    import torch
    from cugraph_pyg_amd.loader import NeighborLoader
    feature_store, graph_store = _temporal_stores(True)
    loader = NeighborLoader((feature_store, graph_store),
                            num_neighbors={("paper", "cites", "paper"): [2, 2, 2], ("author", "writes", "paper"): [2, 2, 0]},
                            batch_size=1, input_nodes=("paper", torch.tensor([3])), input_time=torch.tensor([-1]),
                            time_attr="time", weight_attr="bias" if biased else None, shuffle=False,
                            temporal_comparison="strictly_increasing")
    out = next(iter(loader))
    assert sorted(out["author"].n_id.tolist()) == [0, 1, 2]
    assert out["paper"].n_id.tolist() == [3, 2, 1, 0]
    assert sorted(out["author", "writes", "paper"].e_id.tolist()) == [0, 2, 4, 5]
    assert out["author", "writes", "paper"].num_sampled_edges.tolist() == [2, 2, 0]


@pytest.mark.parametrize("biased", [True, False])
def test_link_neighbor_loader_temporal(hiplib, biased):
    # tests/loader/test_neighbor_loader.py:1061-1176 (homogeneous and heterogeneous edge seeds)
    import torch
    from cugraph_pyg_amd.loader import LinkNeighborLoader
    feature_store, graph_store = _temporal_stores(False)
    loader = LinkNeighborLoader((feature_store, graph_store), num_neighbors=[2, 2, 2], batch_size=1,
                                edge_label_index=torch.tensor([[3], [3]]), edge_label_time=torch.tensor([-1]),
                                time_attr="time", weight_attr="bias" if biased else None, shuffle=False,
                                temporal_comparison="strictly_increasing")
    assert next(iter(loader)).n_id.tolist() == [3, 2, 1, 0]
    feature_store, graph_store = _temporal_stores(True)
    loader = LinkNeighborLoader((feature_store, graph_store),
                                num_neighbors={("paper", "cites", "paper"): [2, 2, 2], ("author", "writes", "paper"): [2, 2, 0]},
                                batch_size=1, edge_label_index=(("author", "writes", "paper"), torch.tensor([[0], [3]])),
                                edge_label_time=torch.tensor([-1]), time_attr="time",
                                weight_attr="bias" if biased else None, shuffle=False,
                                temporal_comparison="strictly_increasing")
    out = next(iter(loader))
    assert sorted(out["author"].n_id.tolist()) == [0, 1, 2]
    assert out["paper"].n_id.tolist() == [3, 2, 1, 0]
    assert sorted(out["author", "writes", "paper"].e_id.tolist()) == [0, 2, 4, 5]
    assert out["author", "writes", "paper"].num_sampled_edges.tolist() == [2, 2, 0]


def test_temporal_sampling_respects_fanout_and_order_on_random_graph(hiplib):
    """Property check at a larger size: every sampled edge passes the comparison against the time its expanded vertex
    was reached at, and no vertex takes more than fan-out edges per hop."""
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    g = torch.Generator().manual_seed(0)
    V, E = 500, 6000
    ei = torch.randint(0, V, (2, E), generator=g)
    tm = torch.randint(0, 100, (E,), generator=g)
    graph_store, feature_store = GraphStore(), FeatureStore()
    graph_store[("n", "e", "n"), "coo", False, (V, V)] = ei
    feature_store[("n", "e", "n"), "time", None] = tm
    seeds = torch.arange(0, 64)
    loader = NeighborLoader((feature_store, graph_store), num_neighbors=[4, 3], batch_size=8, input_nodes=seeds,
                            input_time=torch.full((64,), 60), time_attr="time", shuffle=False, disjoint=True)
    for batch in loader:
        e = batch.e_id.cpu()
        loc = batch.edge_index.cpu()
        n_id = batch.n_id.cpu()
        assert (ei[0][e] == n_id[loc[0]]).all() and (ei[1][e] == n_id[loc[1]]).all()
        reached = torch.full((n_id.numel(),), -1)
        reached[: batch.batch_size] = 60
        off = 0
        for hop, cnt in enumerate(batch.num_sampled_edges.tolist()):
            rows, cols, ts = loc[0, off:off + cnt], loc[1, off:off + cnt], tm[e[off:off + cnt]]
            assert (reached[cols] >= 0).all() and (ts <= reached[cols]).all()          # monotonically decreasing
            assert torch.bincount(cols).max() <= [4, 3][hop]
            for r, t in zip(rows.tolist(), ts.tolist()):                              # first edge decides the time
                if reached[r] < 0:
                    reached[r] = t
            off += cnt


@pytest.mark.parametrize("G,biased", [(1, False), (3, False), (8, False), (4, True)])
def test_hetero_call_group_walk_equals_single_batch_path(hiplib, G, biased):
    """HeteroPygWalk (one batched no-sync launch sequence per hop and edge type for G mini-batches) returns, batch by
    batch, exactly what hetero_neighbor_sample computes for that batch alone through the C-ABI ops."""
    import torch
    from cugraph_pyg_amd.data import GraphStore
    from cugraph_pyg_amd.sampler.sampler import HeteroNeighborSampler, hetero_neighbor_sample
    rng = np.random.default_rng(G)
    n = {"paper": 2000, "author": 1500, "institution": 50, "field": 200}
    rel = {("author", "writes", "paper"): 8000, ("paper", "cites", "paper"): 9000,
           ("paper", "has_topic", "field"): 3000, ("author", "affiliated_with", "institution"): 2000,
           ("paper", "rev_writes", "author"): 8000, ("field", "rev_has_topic", "paper"): 3000}
    gs = GraphStore()
    from cugraph_pyg_amd.data import FeatureStore
    fs = FeatureStore()
    for (s, r, d), m in rel.items():
        gs[(s, r, d), "coo", False, (n[s], n[d])] = torch.from_numpy(np.stack([rng.integers(0, n[s], m), rng.integers(0, n[d], m)]))
        fs[(s, r, d), "w", None] = torch.from_numpy(rng.random(m).astype(np.float32) + 0.1)
    if biased:
        gs._set_weight_attr((fs, "w"))
    fanout = {et: [4, 3, 2] for et in rel}
    fanout[("field", "rev_has_topic", "paper")] = [2, 0, 1]          # a zero fan-out in the middle
    fanout[("author", "affiliated_with", "institution")] = [3, 3, 3]  # destination type never reached from papers
    B = 32
    seeds = torch.from_numpy(rng.permutation(n["paper"])[:B * 8 + 5]).cuda()
    smp = HeteroNeighborSampler(gs._hetero_graphs, fanout, biased=biased, local_seeds_per_call=G * B,
                                num_nodes=n if G != 3 else None)   # with and without the packed renumber table
    got = dict(smp.sample_batches("paper", seeds, B, 1234))
    assert len(got) == 9 and smp._walks
    for b in range(9):
        ref = hetero_neighbor_sample(gs._hetero_graphs, "paper", seeds[b * B:(b + 1) * B], smp.fanout, 1234 + b, biased)
        node, row, col, edge, nn, ne = got[b]
        for t in n:
            assert torch.equal(node[t], ref[0][t]), (b, t)
            assert list(nn[t]) == list(ref[4][t]), (b, t, nn[t], ref[4][t])
        for et in rel:
            assert torch.equal(row[et], ref[1][et]) and torch.equal(col[et], ref[2][et]), (b, et)
            assert torch.equal(edge[et], ref[3][et]), (b, et)
            assert list(ne[et]) == list(ref[5][et]), (b, et)


@pytest.mark.parametrize("G,wdtype", [(1, "float32"), (4, "float32"), (6, "float64")])
def test_biased_call_group_walk_equals_single_batch_path(hiplib, G, wdtype):
    """The biased no-sync hop (A-Res keys inside wgamd_sample_hop_pyg_nosync) gives every mini-batch of a call group what
    the one-batch path draws through wholegraph_csr_weighted_sample_without_replacement + graph_append_unique — incl. rows
    on both sides of every kernel boundary (one-wave kernel, long-row workgroups with LDS / slab keys)."""
    import torch
    from cugraph_pyg_amd.data.graph_store import CSRGraph
    from cugraph_pyg_amd.sampler.sampler import NeighborSampler, neighbor_sample
    rng = np.random.default_rng(G)
    V = 4000
    deg = rng.integers(0, 60, V)
    deg[:6] = [0, 1, 1024, 1025, 13000, 3000]         # hubs: every path of the biased kernels
    row_ptr = np.zeros(V + 1, np.int64)
    row_ptr[1:] = np.cumsum(deg)
    E = int(row_ptr[-1])
    col = rng.integers(0, V, E)
    col[rng.integers(0, E, E // 10)] = rng.integers(0, 6, E // 10)   # make the hubs popular neighbours
    w = (rng.random(E) + 0.05).astype(wdtype)
    graph = CSRGraph(row_ptr=torch.from_numpy(row_ptr).cuda(), col=torch.from_numpy(col).cuda(),
                     edge_id=torch.arange(E, device="cuda"), edge_type=None, weight=torch.from_numpy(w).cuda(), num_vertices=V)
    B = 48
    seeds = torch.from_numpy(np.concatenate([np.arange(6), rng.permutation(V)[:B * 7 + 5 - 6]])).cuda()
    smp = NeighborSampler(graph, fanout=[8, 5], biased=True, local_seeds_per_call=G * B)
    got = dict(smp.sample_batches(seeds, B, 77))
    assert len(got) == 8 and smp._positive_weights is True and smp._walks
    for b in range(8):
        ref = neighbor_sample(graph, seeds[b * B:(b + 1) * B], [8, 5], 77 + b, biased=True)
        for a_, r_ in zip(got[b][:4], ref[:4]):
            assert torch.equal(a_, r_), b
        assert list(got[b][4]) == list(ref[4]) and list(got[b][5]) == list(ref[5])
    # a zero weight anywhere sends the sampler to the one-batch path (libcugraph never returns zero-weight edges)
    w0 = torch.from_numpy(w).cuda()
    w0[5] = 0
    smp0 = NeighborSampler(CSRGraph(row_ptr=graph.row_ptr, col=graph.col, edge_id=graph.edge_id, edge_type=None, weight=w0,
                                    num_vertices=V), fanout=[8, 5], biased=True, local_seeds_per_call=G * B)
    list(smp0.sample_batches(seeds[:B], B, 77))
    assert smp0._positive_weights is False and not smp0._walks


def test_call_group_feature_fetch_equals_per_batch_fetch(hiplib):
    """Node and edge attributes are gathered once per call group and handed out as views: every batch must carry exactly
    what the one-batch-at-a-time loader (filter_store per batch) gives it."""
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    torch.manual_seed(5)
    n, m, B = 5000, 60000, 64
    ei = torch.stack([torch.randint(0, n, (m,)), torch.randint(0, n, (m,))])
    gs, fs = GraphStore(), FeatureStore()
    gs[("n", "e", "n"), "coo", False, (n, n)] = ei
    fs["n", "x", None] = torch.randn(n, 33)
    fs["n", "y", None] = torch.arange(n)
    fs[("n", "e", "n"), "w", None] = torch.randn(m, 3)
    seeds = torch.randperm(n)[:B * 11 + 7].cuda()          # 11 full batches + a ragged one
    def run(per_call):
        return list(NeighborLoader((fs, gs), [6, 4], input_nodes=seeds, batch_size=B, local_seeds_per_call=per_call,
                                   shuffle=False, random_state=3))
    one, grouped = run(B), run(B * 4)
    assert len(one) == len(grouped) == 12
    for a, b in zip(one, grouped):
        for key in ("x", "y", "w", "edge_index", "n_id", "e_id", "batch", "input_id"):
            assert torch.equal(getattr(a, key), getattr(b, key)), key
        assert a.num_nodes == b.num_nodes and a.batch_size == b.batch_size
        assert a.num_sampled_nodes.tolist() == b.num_sampled_nodes.tolist()
        assert torch.equal(b.x, fs["n", "x", None][b.n_id]) and torch.equal(b.w, fs[("n", "e", "n"), "w", None][b.e_id])


def test_hetero_call_group_feature_fetch_equals_per_batch_fetch(hiplib):
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    torch.manual_seed(9)
    n_p, n_a, B = 3000, 1200, 32
    cites = torch.stack([torch.randint(0, n_p, (20000,)), torch.randint(0, n_p, (20000,))])
    writes = torch.stack([torch.randint(0, n_a, (9000,)), torch.randint(0, n_p, (9000,))])
    gs, fs = GraphStore(), FeatureStore()
    gs[("paper", "cites", "paper"), "coo", False, (n_p, n_p)] = cites
    gs[("author", "writes", "paper"), "coo", False, (n_a, n_p)] = writes
    gs[("paper", "rev_writes", "author"), "coo", False, (n_p, n_a)] = writes.flip(0)
    fs["paper", "x", None] = torch.randn(n_p, 17)
    fs["author", "x", None] = torch.randn(n_a, 5)
    fs["paper", "year", None] = torch.arange(n_p)
    fs[("author", "writes", "paper"), "w", None] = torch.randn(9000, 2)
    seeds = torch.randperm(n_p)[:B * 7 + 3].cuda()
    fan = {("paper", "cites", "paper"): [4, 3], ("author", "writes", "paper"): [3, 2], ("paper", "rev_writes", "author"): [2, 2]}
    def run(per_call):
        return list(NeighborLoader((fs, gs), fan, input_nodes=("paper", seeds), batch_size=B, local_seeds_per_call=per_call,
                                   shuffle=False, random_state=11))
    one, grouped = run(B), run(B * 4)
    assert len(one) == len(grouped) == 8
    for a, b in zip(one, grouped):
        for nt in ("paper", "author"):
            assert torch.equal(a[nt].n_id, b[nt].n_id) and torch.equal(a[nt].x, b[nt].x)
            assert torch.equal(b[nt].x, fs[nt, "x", None][b[nt].n_id])
        assert torch.equal(a["paper"].year, b["paper"].year)
        for et in fan:
            assert torch.equal(a[et].edge_index, b[et].edge_index) and torch.equal(a[et].e_id, b[et].e_id)
        et = ("author", "writes", "paper")
        assert torch.equal(a[et].w, b[et].w) and torch.equal(b[et].w, fs[et, "w", None][b[et].e_id])


@pytest.mark.parametrize("mode,amount", [(None, 0), ("binary", 1.0), ("binary", 0.3), ("triplet", 2.0)])
@pytest.mark.parametrize("biased", [False, True])
def test_link_loader_call_groups_equal_one_batch_path(hiplib, mode, amount, biased):
    """LinkNeighborLoader in call groups (row-wise endpoint de-duplication for the whole group + one walk over ragged seed
    lists) must hand out, batch by batch, exactly what the one-batch-at-a-time path does (same negatives, same samples)."""
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import LinkNeighborLoader
    torch.manual_seed(21)
    n, m, B = 3000, 50000, 48
    ei = torch.stack([torch.randint(0, n, (m,)), torch.randint(0, n, (m,))])
    gs, fs = GraphStore(), FeatureStore()
    gs[("n", "e", "n"), "coo", False, (n, n)] = ei
    fs["n", "x", None] = torch.randn(n, 9)
    fs[("n", "e", "n"), "w", None] = torch.rand(m) + 0.05
    eli = ei[:, torch.randperm(m)[:B * 9 + 5]]            # 9 full batches + a ragged one
    label = None if mode == "triplet" else torch.randint(0, 3, (eli.shape[1],))
    def run(groups, per_call=None):
        return list(LinkNeighborLoader((fs, gs), num_neighbors=[5, 3], edge_label_index=eli, edge_label=label, batch_size=B,
                                       neg_sampling=None if mode is None else (mode, amount), shuffle=False, random_state=17,
                                       weight_attr="w" if biased else None, call_groups=groups, local_seeds_per_call=per_call))
    slow, fast, fast1 = run(False), run(True, B * 4), run(True, B)
    assert len(slow) == len(fast) == len(fast1) == 10
    for a, b, c in zip(slow, fast, fast1):
        for other in (b, c):
            for key in ("n_id", "e_id", "edge_index", "edge_label_index", "x", "w", "input_id"):
                assert torch.equal(getattr(a, key), getattr(other, key)), key
            assert a.num_sampled_nodes.tolist() == other.num_sampled_nodes.tolist()
            assert a.num_sampled_edges.tolist() == other.num_sampled_edges.tolist()
            if mode is not None or label is not None:
                assert torch.equal(a.edge_label, other.edge_label)
            if mode == "triplet":
                assert torch.equal(a.dst_neg_index, other.dst_neg_index) and torch.equal(a.src_index, other.src_index)


@pytest.mark.parametrize("mode", [None, "binary"])
def test_sampler_protocol_sample_from_edges(hiplib, mode):
    """BaseSampler.sample_from_edges(EdgeSamplerInput, neg_sampling) -> SamplerOutput with the reference's 4-slot metadata
    (sampler.py:799-896, :621-628); through SampleIterator it yields the same Data as LinkNeighborLoader."""
    import torch
    from cugraph_pyg_amd._compat import EdgeSamplerInput, SamplerOutput
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import LinkNeighborLoader
    from cugraph_pyg_amd.sampler.sampler import BaseSampler, NeighborSampler, SampleIterator
    torch.manual_seed(4)
    n, m, B = 2000, 30000, 32
    ei = torch.stack([torch.randint(0, n, (m,)), torch.randint(0, n, (m,))])
    gs, fs = GraphStore(), FeatureStore()
    gs[("n", "e", "n"), "coo", False, (n, n)] = ei
    fs["n", "x", None] = torch.randn(n, 7)
    eli = ei[:, torch.randperm(m)[:B * 5 + 3]].cuda()
    label = torch.randint(0, 2, (eli.shape[1],)).cuda()
    neg = None if mode is None else (mode, 1.0)
    want = list(LinkNeighborLoader((fs, gs), num_neighbors=[4, 3], edge_label_index=eli, edge_label=label, batch_size=B,
                                   neg_sampling=neg, shuffle=False, random_state=62))
    sampler = BaseSampler(NeighborSampler(gs._graph, fanout=[4, 3]), (fs, gs), batch_size=B)
    index = EdgeSamplerInput(input_id=torch.arange(eli.shape[1], device="cuda"), row=eli[0], col=eli[1], label=label)
    outs = list(sampler.sample_from_edges(index, neg_sampling=neg, random_state=62))
    assert len(outs) == len(want) == 6 and all(isinstance(o, SamplerOutput) and len(o.metadata) == 4 for o in outs)
    got = list(SampleIterator((fs, gs), iter(sampler.sample_from_edges(index, neg_sampling=neg, random_state=62))))
    for a, b, o in zip(want, got, outs):
        for key in ("n_id", "e_id", "edge_index", "edge_label_index", "edge_label", "x", "input_id"):
            assert torch.equal(getattr(a, key), getattr(b, key)), key
        assert torch.equal(o.node, a.n_id) and torch.equal(o.metadata[1], a.edge_label_index)
        assert a.batch_size == b.batch_size


@pytest.mark.parametrize("mode,amount", [(None, 0), ("binary", 1.0), ("triplet", 2.0)])
@pytest.mark.parametrize("etype", [("author", "writes", "paper"), ("paper", "cites", "paper")])
def test_hetero_link_loader_call_groups_equal_one_batch_path(hiplib, mode, amount, etype):
    """Typed edge seeds in call groups (both endpoint types seed the walk with ragged per-batch lists) against the
    one-batch-at-a-time path: identical HeteroData, batch by batch."""
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import LinkNeighborLoader
    torch.manual_seed(33)
    n_p, n_a, B = 2500, 900, 40
    cites = torch.stack([torch.randint(0, n_p, (15000,)), torch.randint(0, n_p, (15000,))])
    writes = torch.stack([torch.randint(0, n_a, (7000,)), torch.randint(0, n_p, (7000,))])
    gs, fs = GraphStore(), FeatureStore()
    gs[("paper", "cites", "paper"), "coo", False, (n_p, n_p)] = cites
    gs[("author", "writes", "paper"), "coo", False, (n_a, n_p)] = writes
    gs[("paper", "rev_writes", "author"), "coo", False, (n_p, n_a)] = writes.flip(0)
    fs["paper", "x", None] = torch.randn(n_p, 6)
    fs["author", "x", None] = torch.randn(n_a, 4)
    fs[("author", "writes", "paper"), "w", None] = torch.randn(7000, 2)
    src_edges = writes if etype[0] == "author" else cites
    eli = src_edges[:, torch.randperm(src_edges.shape[1])[:B * 6 + 9]]
    fan = {("paper", "cites", "paper"): [3, 2], ("author", "writes", "paper"): [2, 2], ("paper", "rev_writes", "author"): [2, 1]}
    def run(groups, per_call=None):
        return list(LinkNeighborLoader((fs, gs), num_neighbors=fan, edge_label_index=(etype, eli), batch_size=B,
                                       neg_sampling=None if mode is None else (mode, amount), shuffle=False, random_state=5,
                                       call_groups=groups, local_seeds_per_call=per_call))
    slow, fast = run(False), run(True, B * 4)
    assert len(slow) == len(fast) == 7
    for a, b in zip(slow, fast):
        for nt in ("paper", "author"):
            assert torch.equal(a[nt].n_id, b[nt].n_id), nt
            if a[nt].n_id.numel():
                assert torch.equal(a[nt].x, b[nt].x)
        for et in fan:
            assert torch.equal(a[et].edge_index, b[et].edge_index) and torch.equal(a[et].e_id, b[et].e_id), et
            assert a[et].num_sampled_edges.tolist() == b[et].num_sampled_edges.tolist()
        assert torch.equal(a[("author", "writes", "paper")].w, b[("author", "writes", "paper")].w)
        assert torch.equal(a[etype].edge_label_index, b[etype].edge_label_index)
        assert torch.equal(a[etype].input_id, b[etype].input_id)
        if mode is not None:
            assert torch.equal(a[etype].edge_label, b[etype].edge_label)


def test_neighbor_loader_replace_true_and_hetero_disjoint(hiplib):
    """`replace=True` (reference: neighbor_loader.py:118-120 -> libcugraph with_replacement) and heterogeneous `disjoint`
    sampling (distributed_sampler.py:808-824), both refused in round 1."""
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    from cugraph_pyg_amd.sampler.sampler import hetero_neighbor_sample
    # homogeneous, replace=True: a vertex with 2 in-neighbours and fan-out 6 yields 6 edges from {its 2 neighbours}
    gs, fs = GraphStore(), FeatureStore()
    src = torch.tensor([1, 2, 0, 3, 4])
    dst = torch.tensor([0, 0, 1, 1, 1])
    gs[("n", "e", "n"), "coo", False, (5, 5)] = [src, dst]
    fs["n", "x", None] = torch.arange(5, dtype=torch.float32).view(-1, 1)
    loader = NeighborLoader((fs, gs), num_neighbors=[6], input_nodes=torch.tensor([0, 1]), batch_size=2, replace=True)
    b = next(iter(loader))
    ei, n_id = b.edge_index.cpu(), b.n_id.cpu()
    assert ei.shape[1] == 12 and b.num_sampled_edges.tolist() == [12]
    for s_, d_ in zip(n_id[ei[0]].tolist(), n_id[ei[1]].tolist()):
        assert (s_, d_) in {(1, 0), (2, 0), (0, 1), (3, 1), (4, 1)}
    assert src[b.e_id.cpu()].tolist() == n_id[ei[0]].tolist() and dst[b.e_id.cpu()].tolist() == n_id[ei[1]].tolist()
    # heterogeneous disjoint: papers 0 and 1 are both written by author 0 and by one author of their own; with vertex-
    # disjoint trees author 0 joins exactly ONE tree and the other paper's edge to it is dropped
    gs2 = GraphStore()
    gs2[("author", "writes", "paper"), "coo", False, (3, 2)] = [torch.tensor([0, 0, 1, 2]), torch.tensor([0, 1, 0, 1])]
    gs2[("paper", "rev_writes", "author"), "coo", False, (2, 3)] = [torch.tensor([0, 1, 0, 1]), torch.tensor([0, 0, 1, 2])]
    graphs = gs2._hetero_graphs
    fan = {et: [5, 5] for et in graphs}
    seeds = torch.tensor([0, 1], device="cuda")
    node, row, col, edge, nn, ne = hetero_neighbor_sample(graphs, "paper", seeds, fan, 7, disjoint=True)
    et = ("author", "writes", "paper")
    a_of = node["author"][row[et]].cpu().tolist()
    p_of = node["paper"][col[et]].cpu().tolist()
    hop1 = list(zip(a_of[:ne[et][0]], p_of[:ne[et][0]]))
    assert sorted(hop1) == [(0, 0), (1, 0), (2, 1)]          # (author 0 -> paper 1) dropped: author 0 sits in paper 0's tree
    assert sorted(node["author"].cpu().tolist()) == [0, 1, 2] and node["paper"].cpu().tolist() == [0, 1]
    # the trees never share a vertex: walking on from the authors reaches no paper of the other tree
    et2 = ("paper", "rev_writes", "author")
    assert ne[et2][1] == 0 or set(node["paper"][row[et2]].cpu().tolist()) <= {0, 1}
    plain = hetero_neighbor_sample(graphs, "paper", seeds, fan, 7)
    assert plain[5][et][0] == 4                               # without `disjoint` all four authorships are sampled
    # and through the loader
    fs2 = FeatureStore()
    fs2["paper", "x", None] = torch.zeros(2, 1)
    out = next(iter(NeighborLoader((fs2, gs2), num_neighbors=fan, input_nodes=("paper", torch.tensor([0, 1])), batch_size=2,
                                   disjoint=True)))
    assert out[et].edge_index.shape[1] == 3
