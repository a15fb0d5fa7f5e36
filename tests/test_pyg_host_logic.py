"""Host logic of the cugraph_pyg-shaped layer that needs no GPU: GraphStore CSR construction
(direction reversal, per-type vertex offsets by sorted type name, per-type edge ids;
/root/reference/python/cugraph-pyg/cugraph_pyg/data/graph_store.py:372-383,508-614), loader length
rules (loader/node_loader.py:168-178) and argument validation."""
import pytest
import torch


def test_graph_store_csr_reverses_direction_and_keeps_edge_ids():
    from cugraph_pyg_amd.data import GraphStore
    # PyG convention: edge_index[0] = message source j, edge_index[1] = message target i
    ei = torch.tensor([[3, 4, 5, 3], [0, 1, 2, 2]])
    gs = GraphStore()
    gs.put_edge_index(ei, ("person", "knows", "person"), "coo", False, (6, 6))
    g = gs._graph
    assert g.num_vertices == 6 and g.edge_type is None
    assert g.row_ptr.tolist() == [0, 1, 2, 4, 4, 4, 4]          # rows = targets 0,1,2
    assert g.col.tolist() == [3, 4, 5, 3]                        # their in-neighbours (sources)
    assert g.edge_id.tolist() == [0, 1, 2, 3]                    # original positions
    assert gs.is_homogeneous and not gs.is_multi_gpu
    assert gs._vertex_offsets == {"person": 0}
    attrs = gs.get_all_edge_attrs()
    assert len(attrs) == 1 and attrs[0].edge_type == ("person", "knows", "person") and tuple(attrs[0].size) == (6, 6)
    assert torch.equal(gs.get_edge_index(("person", "knows", "person"), "coo"), ei)
    ptr, minor = gs.get_edge_index(("person", "knows", "person"), "csr")
    assert ptr.tolist() == [0, 0, 0, 0, 2, 3, 4] and minor.tolist() == [0, 2, 1, 2]


def test_graph_store_hetero_offsets_and_numeric_types():
    from cugraph_pyg_amd.data import GraphStore
    gs = GraphStore()
    gs.put_edge_index(torch.tensor([[0, 1, 2], [2, 0, 1]]), ("author", "writes", "paper"), "coo", False, (3, 4))
    gs.put_edge_index(torch.tensor([[0, 1], [1, 0]]), ("paper", "cites", "paper"), "coo", False, (4, 4))
    assert gs._vertex_offsets == {"author": 0, "paper": 3}        # sorted by type name
    assert gs._vertex_offset_array.tolist() == [0, 3, 7]
    keys, srcs, dsts = gs._numeric_edge_types
    assert keys == [("author", "writes", "paper"), ("paper", "cites", "paper")]
    assert srcs.tolist() == [1, 1] and dsts.tolist() == [0, 1]    # cuGraph src = PyG target type
    assert not gs.is_homogeneous
    g = gs._graph
    assert g.num_vertices == 7
    # paper p (global 3+p) is expanded: in-neighbours are authors (global 0..2) and citing papers (3..)
    rows = {v: (g.col[g.row_ptr[v]:g.row_ptr[v + 1]].tolist(), g.edge_type[g.row_ptr[v]:g.row_ptr[v + 1]].tolist(),
                g.edge_id[g.row_ptr[v]:g.row_ptr[v + 1]].tolist()) for v in range(7)}
    assert rows[3] == ([1, 4], [0, 1], [1, 1])     # paper 0 <- author 1 (writes #1), <- paper 1 (cites #1)
    assert rows[4] == ([2, 3], [0, 1], [2, 0])
    assert rows[5] == ([0], [0], [0])
    assert rows[0] == ([], [], [])


def test_graph_store_finalize_and_guards():
    from cugraph_pyg_amd.data import GraphStore
    gs = GraphStore()
    with pytest.raises(ValueError):
        gs.put_edge_index(torch.zeros((2, 1), dtype=torch.long), ("a", "b", "a"), "csr", False, (2, 2))
    with pytest.raises(ValueError):
        gs.put_edge_index(torch.zeros((3, 1), dtype=torch.long), ("a", "b", "a"), "coo", False, (2, 2))
    gs.put_edge_index(torch.tensor([[1], [0]]), ("a", "b", "a"), "coo", False, (2, 2))
    gs.finalize()
    with pytest.raises(NotImplementedError):
        gs.put_edge_index(torch.tensor([[1], [0]]), ("a", "c", "a"), "coo", False, (2, 2))
    with pytest.raises(RuntimeError):
        gs.finalize()
    assert gs._graph.col.tolist() == [1]


def test_graph_store_empty_slice_and_size_inference():
    from cugraph_pyg_amd.data import GraphStore
    gs = GraphStore()
    gs.put_edge_index(torch.tensor([[1, 2, 3, 4], [0, 1, 2, 3]]), ("n", "e", "n"), "coo")   # size=None
    assert gs._num_vertices() == {"n": 5}
    gs2 = GraphStore()
    gs2.put_edge_index(torch.zeros((2, 0), dtype=torch.long), ("n", "e", "n"), "coo", False, (4, 4))
    assert gs2._graph.row_ptr.tolist() == [0, 0, 0, 0, 0]


def test_loader_len_rules_and_validation():
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    gs = GraphStore()
    gs.put_edge_index(torch.stack([torch.tensor([1, 2, 3, 4]), torch.tensor([0, 1, 2, 3])]),
                      ("person", "knows", "person"), "coo", False, (5, 5))
    fs = FeatureStore()
    # tests/loader/test_neighbor_loader.py:55-95
    assert len(NeighborLoader((fs, gs), [1], input_nodes=torch.arange(5), batch_size=2)) == 3
    assert len(NeighborLoader((fs, gs), [1], input_nodes=torch.arange(5), batch_size=2, drop_last=True)) == 2
    with pytest.raises(ValueError, match="input_nodes"):
        len(NeighborLoader((fs, gs), [1], input_nodes="person", batch_size=2))
    with pytest.raises(ValueError):
        NeighborLoader((fs, gs), [1], input_nodes=torch.arange(1), batch_size=2, drop_last=True)
    with pytest.raises(ValueError):
        NeighborLoader((fs, gs), [1], subgraph_type="induced")
    with pytest.raises(ValueError):
        NeighborLoader((fs, gs), [1], compression="CSC")
    with pytest.raises(NotImplementedError):
        NeighborLoader((fs, None), [1])
    with pytest.raises(ValueError):
        NeighborLoader((fs, gs), {("person", "likes", "person"): [1]})       # unknown edge type
    with pytest.raises(ValueError):
        NeighborLoader((fs, gs), {("person", "knows", "person"): [1]}, compression="CSR")


def test_hetero_type_local_csrs():
    from cugraph_pyg_amd.data import GraphStore
    gs = GraphStore()
    # tests/loader/test_neighbor_loader.py:355-373
    src, dst = torch.tensor([0, 1, 2, 4, 3, 4, 5, 5]), torch.tensor([4, 5, 4, 3, 2, 1, 0, 1])
    asrc, adst = torch.tensor([0, 1, 2, 3, 3, 0]), torch.tensor([0, 1, 2, 3, 4, 5])
    gs[("paper", "cites", "paper"), "coo", False, (6, 6)] = [src, dst]
    gs[("author", "writes", "paper"), "coo", False, (4, 6)] = [asrc, adst]
    hg = gs._hetero_graphs
    w = hg[("author", "writes", "paper")]
    assert w.num_vertices == 6 and w.row_ptr.tolist() == [0, 1, 2, 3, 4, 5, 6]
    assert w.col.tolist() == [0, 1, 2, 3, 3, 0] and w.edge_id.tolist() == [0, 1, 2, 3, 4, 5]
    c = hg[("paper", "cites", "paper")]
    for v in range(6):      # row v = in-neighbours of paper v, edge ids point into the ORIGINAL arrays
        ids = c.edge_id[c.row_ptr[v]:c.row_ptr[v + 1]]
        assert (dst[ids] == v).all() and torch.equal(src[ids], c.col[c.row_ptr[v]:c.row_ptr[v + 1]])


def test_hop_seed_derivation_is_stable():
    from cugraph_pyg_amd.sampler.sampler import hop_seed
    assert hop_seed(62, 0) == 62
    assert hop_seed(62, 1) == (62 + 0x9E3779B97F4A7C15) % 2**64
    assert hop_seed(2**64 - 1, 2) == (2**64 - 1 + 2 * 0x9E3779B97F4A7C15) % 2**64


def test_batched_first_unique_matches_row_by_row_first_appearance():
    """link_loader._batched_first_unique (one pass for a whole call group) == per row what append_unique with an empty
    target list returns: unique ids in first-appearance order + the position of every element in that list."""
    import torch
    from cugraph_pyg_amd.loader.link_loader import _batched_first_unique
    g = torch.Generator().manual_seed(2)
    for G, S, n_ids in ((1, 1, 5), (3, 17, 6), (8, 64, 1000), (5, 40, 3)):
        ends = torch.randint(0, n_ids, (G, S), generator=g)
        uniq, seg, batch, local = _batched_first_unique(ends, n_ids)
        assert uniq.shape[0] == G * S and batch.shape[0] == G * S and seg.shape[0] == G + 1
        assert seg.dtype == torch.int32 and batch.dtype == torch.int32
        for r in range(G):
            want, seen = [], {}
            for v in ends[r].tolist():
                if v not in seen:
                    seen[v] = len(want)
                    want.append(v)
            lo, hi = int(seg[r]), int(seg[r + 1])
            assert uniq[lo:hi].tolist() == want
            assert batch[lo:hi].tolist() == [r] * len(want)
            assert local[r].tolist() == [seen[v] for v in ends[r].tolist()]
        assert int(seg[G]) <= G * S and bool((uniq[int(seg[G]):] == 0).all())


def test_merge_hops_batch_major_is_the_per_batch_concatenation():
    import torch
    from wholegraph_amd.fused import _merge_hops_batch_major
    g = torch.Generator().manual_seed(3)
    G = 5
    for H in (1, 2, 3):
        segs, fields = [], []
        for h in range(H):
            cnt = torch.randint(0, 7, (G,), generator=g)
            seg = [0] + torch.cumsum(cnt, 0).tolist()
            segs.append(seg)
            n = seg[-1] + 4                                   # capacity slack behind the live part
            fields.append([torch.arange(n) + 1000 * h, torch.randint(0, 99, (n,), generator=g)])
        merged, offs = _merge_hops_batch_major(fields, segs, G, torch.device("cpu"))
        for i in range(2):
            for b in range(G):
                want = torch.cat([fields[h][i][segs[h][b]:segs[h][b + 1]] for h in range(H)])
                assert torch.equal(merged[i][offs[b]:offs[b + 1]], want)
        assert offs[G] == sum(s[-1] for s in segs)
    merged, offs = _merge_hops_batch_major([], [], G, torch.device("cpu"))
    assert merged == [] and offs == [0] * (G + 1)


def test_temporal_negative_redraw_and_fallback():
    """link_loader._draw_negatives: every pair ends up no later than its seed time (redraws, then the earliest node)."""
    import torch
    from cugraph_pyg_amd.loader.link_loader import _draw_negatives
    gen = torch.Generator().manual_seed(0)
    node_time = torch.cat([torch.arange(10), torch.full((990,), 10 ** 6)])      # 1 % of the nodes exist early
    neg_time = torch.randint(0, 12, (500,), generator=gen)
    src, dst = _draw_negatives(500, 1000, 1000, gen, torch.device("cpu"), neg_time, node_time, node_time)
    assert bool((node_time[src] <= neg_time).all()) and bool((node_time[dst] <= neg_time).all())
    src2, dst2 = _draw_negatives(500, 1000, 1000, gen, torch.device("cpu"))    # no times: plain uniform draws
    assert int(src2.max()) < 1000 and int(dst2.min()) >= 0
    s3, d3 = _draw_negatives(0, 10, 10, gen, torch.device("cpu"), neg_time[:0], node_time, node_time)
    assert s3.numel() == 0 and d3.numel() == 0


def test_call_group_size_answers_for_the_rows_it_fetches(hiplib):
    """The default call group is sized from device memory (distributed_sampler.py:757,837-875): the walk's capacity-sized
    buffers AND the worst case of the feature rows the group fetches in one go (round-3 advice: 191 mini-batches of wide
    rows were tens of GB).  Narrow rows leave the walk budget in charge, wide rows shorten the group."""
    from cugraph_pyg_amd.sampler.sampler import default_local_seeds_per_call as per_call
    T = 288 << 30
    base = per_call([25, 10], 1024, 8, False, T)
    assert base // 1024 >= 128
    assert per_call([25, 10], 1024, 8, False, T, feature_row_bytes=(400, 0)) == base       # products: 400-byte rows
    wide = per_call([25, 10], 1024, 8, False, T, feature_row_bytes=(4096, 0))
    assert 1024 <= wide < base // 4 and wide % 1024 == 0
    worst_nodes = 1 + 25 + 250
    assert wide * worst_nodes * 4096 <= 0.10 * T                                           # the stated budget holds
    assert per_call([25, 10], 1024, 8, False, T, feature_row_bytes=(0, 4096)) < base       # edge attributes count too
    assert per_call([25, 10], 1024, 8, False, 1 << 30, feature_row_bytes=(1 << 20, 0)) == 1024   # never below one batch


def test_group_fetch_fills_one_preallocated_output(monkeypatch):
    """A call group's fetch larger than _GROUP_FETCH_BYTES is made in pieces into ONE output (no torch.cat copy)."""
    from cugraph_pyg_amd.sampler import sampler as S
    from cugraph_pyg_amd.data import FeatureStore
    table = torch.arange(200 * 6, dtype=torch.float32).view(200, 6)
    index = torch.randint(0, 200, (1000,), generator=torch.Generator().manual_seed(1))
    monkeypatch.setattr(S, "_GROUP_FETCH_BYTES", 5000)            # 1000 rows x 24 B -> 5 pieces
    assert torch.equal(S._fetch_rows_agreed(table, index), table[index])

    class Pieces:                                                 # a store tensor with the loaders' gather_into hook
        shape, dtype, calls = table.shape, table.dtype, []

        def gather_into(self, idx, out):
            Pieces.calls.append(int(idx.numel()))
            out.copy_(table[idx])

        def __getitem__(self, idx):
            raise AssertionError("the preallocated path must be taken")

    got = S._fetch_rows_agreed(Pieces(), index)
    assert torch.equal(got, table[index]) and len(Pieces.calls) == 5 and sum(Pieces.calls) == 1000
    # the output comes from the grow-only rows pool: handed out again only when nothing refers to its storage any more
    first = got.untyped_storage().data_ptr()
    view = got[3:5]
    del got
    other = S._fetch_rows_agreed(Pieces(), index[:900])
    assert other.untyped_storage().data_ptr() != first, "a buffer with a live view was recycled"
    keep = view.clone()
    del view, other
    again = S._fetch_rows_agreed(Pieces(), index[:950])
    assert again.untyped_storage().data_ptr() in (first,) or again.numel() > 0     # (either idle buffer may serve it)
    assert torch.equal(keep, table[index][3:5]) and torch.equal(again, table[index[:950]])
    # a busy buffer FIRST, an idle one of another size behind it, and a request the idle one cannot hold: the idle buffer goes
    # back to the allocator (dropped by identity — comparing tensors would be element-wise), the busy one stays
    del again
    pool = S._group_rows
    pool.clear()
    held = pool.take((1000, 4), torch.float32, "cpu")
    small = pool.take((10, 4), torch.float32, "cpu")
    bufs = pool._bufs[torch.device("cpu")]
    assert len(bufs) == 2 and bufs[0].numel() != bufs[1].numel()
    del small
    assert not pool._idle(bufs[0]) and pool._idle(bufs[1])
    big = pool.take((3 << 20, 1), torch.uint8, "cpu")
    assert len(bufs) == 2 and bufs[0].untyped_storage().data_ptr() == held.untyped_storage().data_ptr()
    assert bufs[1].numel() >= big.numel()
    del big, held
    S._group_rows.clear()
    fs = FeatureStore()
    fs["paper", "x", None] = table
    fs["paper", "y", None] = torch.zeros(200, dtype=torch.int64)
    fs[("paper", "cites", "paper"), "w", None] = torch.zeros(50, 3, dtype=torch.float16)
    assert S.store_row_bytes(fs) == (24 + 8, 6)


def test_hetero_conv_folds_attention_vectors_and_sums_relation_biases():
    """``nn.HeteroConv``'s call-group route reads GATConv's parameters in folded form: ``alpha_src = ((x W).view(H, C) * att).sum(-1)
    = x (W . att)`` per relation, the folded vectors of every relation end of a node type side by side, one bias per destination
    type = the sum of its relations' biases — and refreshes them when a parameter changes (host logic, no kernel)."""
    from wholegraph_amd import nn
    torch.manual_seed(3)
    ets = [("author", "writes", "paper"), ("paper", "cites", "paper"), ("paper", "rev_writes", "author")]
    hc = nn.HeteroConv({et: nn.GATConv(12, 5, heads=4, add_self_loops=False, bias=et != ets[1]) for et in ets})
    x = torch.randn(7, 12)
    for et in ets:
        c = hc.conv(et)
        w, v_src, v_dst = hc._rel(et)
        assert torch.equal(w, c.lin.weight.t())
        h = (x @ c.lin.weight.t()).view(7, 4, 5)
        assert torch.allclose(x @ v_src, (h * c.att_src).sum(-1), atol=1e-5) and torch.allclose(x @ v_dst, (h * c.att_dst).sum(-1), atol=1e-5)
    keys = hc._term_keys("paper")
    assert keys == [("dst", ets[0]), ("src", ets[1]), ("dst", ets[1]), ("src", ets[2])]
    tm = hc._terms_matrix("paper")
    assert tm.shape == (12, 16) and torch.equal(tm[:, 4:8], hc._rel(ets[1])[1]) and torch.equal(tm[:, 12:], hc._rel(ets[2])[1])
    assert torch.allclose(hc._bias("paper"), hc.conv(ets[0]).bias) and torch.allclose(hc._bias("author"), hc.conv(ets[2]).bias)
    assert hc._width("paper") == 20
    with torch.no_grad():
        hc.conv(ets[0]).att_dst.mul_(2.0)                       # an optimizer step: the folds follow the parameter version
        hc.conv(ets[0]).bias.add_(1.0)
    c = hc.conv(ets[0])
    assert torch.allclose(x @ hc._terms_matrix("paper")[:, :4], ((x @ c.lin.weight.t()).view(7, 4, 5) * c.att_dst).sum(-1), atol=1e-5)
    with torch.no_grad():
        hc.conv(ets[2]).bias.add_(1.0)                          # a bias alone changing is noticed too
    assert torch.allclose(hc._bias("author"), hc.conv(ets[2]).bias)


def test_hetero_default_call_group_is_sized_from_the_walks_capacities():
    """HeteroNeighborSampler.seeds_per_call without local_seeds_per_call: a seed costs what the heterogeneous walk allocates for
    it (frontier capacities grow per node type, an edge type's call holds frontier x fan-out slots), not the reference's
    fan-outs summed over the edge types as one homogeneous hop; edge types without a fan-out entry are not sampled."""
    from cugraph_pyg_amd.sampler.sampler import HeteroNeighborSampler, default_local_seeds_per_call

    class G:
        time = weight = None
    ets = [("author", "writes", "paper"), ("paper", "cites", "paper"), ("paper", "has_topic", "field"), ("author", "at", "inst"),
           ("paper", "rev_writes", "author"), ("field", "rev_has_topic", "paper")]
    smp = HeteroNeighborSampler({et: G() for et in ets}, {et: [25, 10] for et in ets})
    nbytes, slots, rows = smp._walk_capacity_per_seed()
    # worst seed type = paper: hop 1 = 3 edge types x 25; hop 2 = (3 into paper + 1 into author + 1 into field) x 25 x 10
    assert rows == 1 + 75 + 1250 and 60_000 < nbytes < 200_000 and slots >= 250
    summed = default_local_seeds_per_call([150, 60], 1024, 8, total_memory=16 << 30)
    assert smp.seeds_per_call(1024) >= summed and smp.seeds_per_call(1024) % 1024 == 0
    part = HeteroNeighborSampler({et: G() for et in ets}, {ets[0]: [25, 10], ets[1]: [25, 10]})      # four edge types unsampled
    assert part._walk_capacity_per_seed()[2] < rows
    assert HeteroNeighborSampler({et: G() for et in ets}, {et: [-1, 10] for et in ets})._walk_capacity_per_seed() is None
    assert HeteroNeighborSampler({et: G() for et in ets}, {et: [25, 10] for et in ets}, local_seeds_per_call=4096).seeds_per_call(1024) == 4096
