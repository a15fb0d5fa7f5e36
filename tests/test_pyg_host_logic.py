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
