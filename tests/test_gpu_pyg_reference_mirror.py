"""More of the reference's own loader / store tests, restated against cugraph_pyg_amd on the HIP path
(/root/reference/python/cugraph-pyg/cugraph_pyg/tests/loader/test_neighbor_loader.py, tests/data/test_feature_store.py,
tests/data/test_graph_store.py): same graphs, same assertions.  ``torch_geometric`` is not installed here: its
``NegativeSampling(mode, amount)`` is passed as the tuple the loader also accepts, ``HeteroData()`` as an empty feature
store."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)


@pytest.mark.parametrize("batch_size", [1, 3])
def test_link_neighbor_loader_negative_sampling_uneven(hiplib, batch_size):
    # test_neighbor_loader.py:317-350: 0.1 negatives per positive still puts a positive first in every batch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import LinkNeighborLoader
    torch.manual_seed(0)
    num_edges, num_nodes, select_edges = 62, 19, 17
    graph_store, feature_store = GraphStore(), FeatureStore()
    eix = torch.randperm(num_edges)[:select_edges]
    graph_store[("n", "e", "n"), "coo", False, (num_nodes, num_nodes)] = torch.stack(
        [torch.randint(0, num_nodes, (num_edges,)), torch.randint(0, num_nodes, (num_edges,))])
    elx = graph_store[("n", "e", "n"), "coo"][:, eix]
    loader = LinkNeighborLoader((feature_store, graph_store), num_neighbors=[3, 3, 3], edge_label_index=elx,
                                batch_size=batch_size, neg_sampling=("binary", 0.1), shuffle=False)
    n = 0
    for batch in loader:
        assert batch.edge_label[0] == 1.0
        n += 1
    assert n == -(-select_edges // batch_size)


def _paper_author():
    src = torch.tensor([0, 1, 2, 4, 3, 4, 5, 5])   # paper
    dst = torch.tensor([4, 5, 4, 3, 2, 1, 0, 1])   # paper
    asrc = torch.tensor([0, 1, 2, 3, 3, 0])        # author
    adst = torch.tensor([0, 1, 2, 3, 4, 5])        # paper
    return src, dst, asrc, adst


def test_neighbor_loader_hetero_single_etype(hiplib):
    # test_neighbor_loader.py:414-450: an edge type without a fan-out contributes nothing, but is present
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    src, dst, asrc, adst = _paper_author()
    graph_store, feature_store = GraphStore(), FeatureStore()
    graph_store[("paper", "cites", "paper"), "coo", False, (6, 6)] = [src, dst]
    graph_store[("author", "writes", "paper"), "coo", False, (4, 6)] = [asrc, adst]
    loader = NeighborLoader((feature_store, graph_store), num_neighbors={("paper", "cites", "paper"): [1, 1]},
                            input_nodes=("paper", torch.tensor([0, 1])), batch_size=2)
    out = next(iter(loader))
    assert out["author"].n_id.numel() == 0
    assert out["author", "writes", "paper"].edge_index.numel() == 0
    assert out["author", "writes", "paper"].num_sampled_edges.tolist() == [0, 0]


EI_12 = torch.tensor([[14, 14, 0, 7, 8, 7, 13, 13, 3, 13, 14, 6, 3, 14, 3, 1, 11, 11, 13, 4],
                      [7, 0, 3, 1, 0, 0, 0, 4, 2, 3, 3, 1, 4, 3, 0, 6, 5, 1, 4, 4]])
ELI = torch.tensor([[3, 14, 4, 0, 14, 13, 8, 13, 6, 11, 14, 13, 13, 1, 11, 7],
                    [2, 0, 4, 3, 3, 4, 0, 0, 1, 1, 3, 4, 3, 6, 5, 0]])


@pytest.mark.parametrize("three_types", [False, True])
def test_neighbor_loader_hetero_linkpred_bidirectional_v2_and_three_types(hiplib, three_types):
    # test_neighbor_loader.py:586-735: the seed edges come back, in order, through n_id[edge_label_index]
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import LinkNeighborLoader
    feature_store, graph_store = FeatureStore(), GraphStore()
    graph_store[("n1", "e", "n2"), "coo", False, (15, 8)] = EI_12
    graph_store[("n2", "f", "n1"), "coo", False, (8, 15)] = EI_12.flip(0)
    fan = {("n1", "e", "n2"): [2, 2], ("n2", "f", "n1"): [2, 2]}
    if three_types:
        ei_13 = torch.tensor([[1, 3, 5, 6, 8, 14, 14], [2, 4, 6, 8, 9, 0, 1]])
        ei_23 = torch.tensor([[7, 0, 3, 2, 2, 1, 1, 5, 4, 2], [9, 8, 1, 2, 3, 9, 8, 4, 6, 5]])
        graph_store[("n1", "g", "n3"), "coo", False, (15, 10)] = ei_13
        graph_store[("n2", "h", "n3"), "coo", False, (8, 10)] = ei_23
        graph_store[("n3", "i", "n1"), "coo", False, (10, 15)] = ei_13.flip(0)
        graph_store[("n3", "j", "n2"), "coo", False, (10, 8)] = ei_23.flip(0)
        fan.update({("n1", "g", "n3"): [2, 2], ("n2", "h", "n3"): [2, 2], ("n3", "i", "n1"): [2, 2],
                    ("n3", "j", "n2"): [2, 2]})
    loader = LinkNeighborLoader(data=(feature_store, graph_store), num_neighbors=fan,
                                edge_label_index=(("n1", "e", "n2"), ELI), edge_label=None, batch_size=2, shuffle=False)
    i = -1
    for i, batch in enumerate(loader):
        eli_i = ELI[:, i * 2:(i + 1) * 2]
        r_i = torch.stack([batch["n1"].n_id[batch["n1", "e", "n2"].edge_label_index[0].cpu()].cpu(),
                           batch["n2"].n_id[batch["n1", "e", "n2"].edge_label_index[1].cpu()].cpu()])
        assert (r_i == eli_i).all()
    assert i == 7


@pytest.mark.parametrize("batch_size", [1, 2])
@pytest.mark.parametrize("neg_sampling_mode", ["binary", "triplet"])
def test_link_neighbor_loader_temporal_negative_sampling_homogeneous(hiplib, batch_size, neg_sampling_mode):
    # test_neighbor_loader.py:1180-1290: negatives only among nodes that exist at the seed edge's time, 2 per positive
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import LinkNeighborLoader
    src_cite = torch.tensor([3, 2, 1, 2, 3, 4, 0])
    dst_cite = torch.tensor([2, 1, 0, 0, 1, 2, 1])
    tme_cite = torch.tensor([5, 6, 7, 3, 4, 8, 2])
    node_time = torch.tensor([0, 1, 2, 3, 4])
    graph_store, feature_store = GraphStore(), FeatureStore()
    graph_store[("paper", "cites", "paper"), "coo", False, (5, 5)] = [dst_cite, src_cite]
    feature_store[("paper", "cites", "paper"), "time", None] = tme_cite
    feature_store["paper", "time", None] = node_time
    edge_label_index = torch.tensor([[3, 2], [2, 1]])
    edge_label_time = torch.tensor([10, 10])
    loader = LinkNeighborLoader((feature_store, graph_store), num_neighbors=[2, 2], batch_size=batch_size,
                                edge_label_index=edge_label_index, edge_label_time=edge_label_time, time_attr="time",
                                neg_sampling=(neg_sampling_mode, 2.0), shuffle=False)
    total_pos = total_neg = 0
    i = -1
    for i, batch in enumerate(loader):
        assert hasattr(batch, "edge_label") and hasattr(batch, "edge_label_index")
        labels = batch.edge_label
        assert torch.any(labels == 1.0) and torch.any(labels == 0.0)
        total_pos += int((labels == 1.0).sum())
        total_neg += int((labels == 0.0).sum())
        eli = batch.edge_label_index
        assert eli.shape[0] == 2 and eli.shape[1] == len(labels)
        neg = labels == 0.0
        for ids in (batch.n_id[eli[0, neg].cpu()].cpu(), batch.n_id[eli[1, neg].cpu()].cpu()):
            assert bool((node_time[ids] <= edge_label_time[i * batch_size]).all())
    assert total_neg == 2 * total_pos and i >= 0


@pytest.mark.parametrize("batch_size", [1, 2])
@pytest.mark.parametrize("neg_sampling_mode", ["binary", "triplet"])
def test_link_neighbor_loader_temporal_negative_sampling_heterogeneous(hiplib, batch_size, neg_sampling_mode):
    # test_neighbor_loader.py:1293-1420
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import LinkNeighborLoader
    src_cite, dst_cite, tme_cite = torch.tensor([3, 2, 1, 2]), torch.tensor([2, 1, 0, 0]), torch.tensor([5, 6, 7, 3])
    src_author = torch.tensor([3, 2, 2, 1, 3, 2, 0])
    dst_author = torch.tensor([0, 0, 1, 1, 2, 2, 2])
    tme_author = torch.tensor([4, 3, 5, 2, 7, 6, 1])
    paper_time, author_time = torch.tensor([0, 1, 2, 3]), torch.tensor([0, 1, 2])
    graph_store, feature_store = GraphStore(), FeatureStore()
    graph_store[("paper", "cites", "paper"), "coo", False, (4, 4)] = [dst_cite, src_cite]
    graph_store[("author", "writes", "paper"), "coo", False, (3, 4)] = [dst_author, src_author]
    feature_store[("paper", "cites", "paper"), "time", None] = tme_cite
    feature_store[("author", "writes", "paper"), "time", None] = tme_author
    feature_store["paper", "time", None] = paper_time
    feature_store["author", "time", None] = author_time
    et = ("author", "writes", "paper")
    edge_label_index = torch.stack([torch.tensor([0, 1, 2]), torch.tensor([3, 2, 1])])
    edge_label_time = torch.tensor([8, 8, 8])
    loader = LinkNeighborLoader((feature_store, graph_store),
                                num_neighbors={("paper", "cites", "paper"): [2, 2], et: [2, 2]}, batch_size=batch_size,
                                edge_label_index=(et, edge_label_index), edge_label_time=edge_label_time,
                                time_attr="time", neg_sampling=(neg_sampling_mode, 2.0), shuffle=False)
    total_pos = total_neg = 0
    for i, batch in enumerate(loader):
        assert "author" in batch.node_types and "paper" in batch.node_types
        assert [et] == list(batch.edge_label_index_dict.keys()) and [et] == list(batch.edge_label_dict.keys())
        labels = batch[et].edge_label
        assert torch.any(labels == 1.0) and torch.any(labels == 0.0)
        total_pos += int((labels == 1.0).sum())
        total_neg += int((labels == 0.0).sum())
        eli = batch[et].edge_label_index
        assert eli.shape[0] == 2 and eli.shape[1] == len(labels)
        neg = labels == 0.0
        a_ids = batch["author"].n_id[eli[0, neg].cpu()].cpu()
        p_ids = batch["paper"].n_id[eli[1, neg].cpu()].cpu()
        assert bool((author_time[a_ids] <= edge_label_time[i * batch_size]).all())
        assert bool((paper_time[p_ids] <= edge_label_time[i * batch_size]).all())
        assert batch["author"].n_id.numel() > 0 and batch["paper"].n_id.numel() > 0
    assert total_neg == 2 * total_pos


def test_feature_store_basic_api(hiplib):
    # tests/data/test_feature_store.py:16-47
    from cugraph_pyg_amd.data import FeatureStore
    feature_store = FeatureStore()
    node_features_0 = torch.randint(128, (100, 1000))
    node_features_1 = torch.randint(256, (100, 10))
    other_features = torch.randint(1024, (10, 5))
    feature_store["node", "feat0", None] = node_features_0
    feature_store["node", "feat1", None] = node_features_1
    feature_store["other", "feat", None] = other_features
    assert (feature_store["node", "feat0", None].get_local_tensor().cpu() == node_features_0).all()
    assert (feature_store["node", "feat1", None].get_local_tensor().cpu() == node_features_1).all()
    assert (feature_store["other", "feat", None].get_local_tensor().cpu() == other_features).all()
    ixr = torch.randperm(node_features_0.shape[0])
    assert (feature_store["node", "feat0", None][ixr].cpu() == node_features_0[ixr]).all()
    assert len(feature_store.get_all_tensor_attrs()) == 3
    del feature_store["node", "feat0", None]
    assert len(feature_store.get_all_tensor_attrs()) == 2


@pytest.mark.parametrize("dtype_name", ["float32", "float16", "int8", "int16", "int32", "int64", "float64", "bfloat16"])
def test_feature_store_basic_api_types(hiplib, dtype_name):
    # tests/data/test_feature_store.py:50-85
    from cugraph_pyg_amd.data import FeatureStore
    dtype = getattr(torch, dtype_name)
    features = torch.arange(0, 2000)
    features = features.reshape((features.numel() // 100, 100)).to(dtype)
    whole_store = FeatureStore()
    whole_store["node", "fea", None] = features
    ix = torch.arange(features.shape[0])
    assert (whole_store["node", "fea", None][ix].cpu() == features[ix]).all()
    ix = torch.randperm(features.shape[0])
    label = torch.arange(0, features.shape[0]).reshape((features.shape[0], 1))
    whole_store["node", "label", None] = label
    assert (whole_store["node", "label", None][ix].cpu() == label[ix]).all()


@pytest.mark.parametrize("location", ["cpu", "cuda"])
def test_graph_store_basic_api_and_finalize(hiplib, location):
    # tests/data/test_graph_store.py:18-82 on the karate edge list (tests/golden/karate.csv)
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    e = np.loadtxt(os.path.join(HERE, "golden", "karate.csv"), dtype=np.int64, usecols=(0, 1))
    src, dst = torch.as_tensor(e[:, 0], device="cuda"), torch.as_tensor(e[:, 1], device="cuda")
    ei = torch.stack([dst, src])
    num_nodes = 34
    et = ("person", "knows", "person")
    graph_store = GraphStore(location=location)
    graph_store.put_edge_index(ei, et, "coo", False, (num_nodes, num_nodes))
    rei = graph_store.get_edge_index(et, "coo")
    assert (ei == rei.to(ei.device)).all()
    assert len(graph_store.get_all_edge_attrs()) == 1
    graph_store.remove_edge_index(et, "coo")
    assert len(graph_store.get_all_edge_attrs()) == 0
    graph_store = GraphStore()
    graph_store.put_edge_index(ei, et, "coo", False, (num_nodes, num_nodes))
    graph_store.finalize()
    with pytest.raises(NotImplementedError, match="Adding edges"):
        graph_store.put_edge_index(ei, et, "coo", False, (num_nodes, num_nodes))
    with pytest.raises(NotImplementedError, match="Removing edges"):
        graph_store.remove_edge_index(et, "coo")
    feature_store = FeatureStore()
    feature_store["person", "feat", None] = torch.arange(num_nodes).reshape(-1, 1)
    loader = NeighborLoader((feature_store, graph_store), [5, 5], input_nodes=torch.arange(num_nodes), batch_size=num_nodes)
    batch = next(iter(loader))
    assert (feature_store["person", "feat", None][batch.n_id] == batch.feat).all()


@pytest.mark.parametrize("neg_sampling_mode", ["binary", "triplet"])
def test_temporal_negatives_avoid_nodes_from_the_future(hiplib, neg_sampling_mode):
    """The reference's temporal negative-sampling tests only use node times that every seed time admits; here half of
    the nodes appear AFTER the seed edges, so the redraw / earliest-node fallback of sampler_utils.py:243-311 must act."""
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import LinkNeighborLoader
    torch.manual_seed(1)
    n = 40
    src, dst = torch.randint(0, n, (300,)), torch.randint(0, n, (300,))
    node_time = torch.cat([torch.arange(20), torch.full((20,), 1000)])       # nodes 20.. only exist from t = 1000
    graph_store, feature_store = GraphStore(), FeatureStore()
    graph_store[("paper", "cites", "paper"), "coo", False, (n, n)] = [src, dst]
    feature_store[("paper", "cites", "paper"), "time", None] = torch.randint(0, 50, (300,))
    feature_store["paper", "time", None] = node_time
    eli = torch.stack([torch.randint(0, 20, (64,)), torch.randint(0, 20, (64,))])
    eli_time = torch.randint(5, 60, (64,))                                   # some seeds admit only nodes 0..5
    loader = LinkNeighborLoader((feature_store, graph_store), num_neighbors=[3, 3], batch_size=8, edge_label_index=eli,
                                edge_label_time=eli_time, time_attr="time", neg_sampling=(neg_sampling_mode, 3.0),
                                shuffle=False)
    seen_neg = 0
    for i, batch in enumerate(loader):
        labels, idx = batch.edge_label, batch.edge_label_index
        n_pos = int((labels == 1.0).sum())
        assert n_pos == 8 and int((labels == 0.0).sum()) == 24
        t_pos = eli_time[i * 8:(i + 1) * 8]
        t_neg = t_pos[torch.arange(24) % 8]                                  # negative k belongs to positive k mod n_pos
        neg_src = batch.n_id[idx[0, n_pos:].cpu()].cpu()
        neg_dst = batch.n_id[idx[1, n_pos:].cpu()].cpu()
        assert bool((node_time[neg_dst] <= t_neg).all())
        if neg_sampling_mode == "triplet":                                   # negative k starts where positive k mod 8 starts
            assert torch.equal(neg_src, eli[0, i * 8:(i + 1) * 8][torch.arange(24) % 8])
            assert torch.equal(batch.n_id[batch.dst_neg_index.cpu()].cpu(), neg_dst.view(3, 8).t())
            assert torch.equal(batch.n_id[batch.src_index.cpu()].cpu(), eli[0, i * 8:(i + 1) * 8])
        else:
            assert bool((node_time[neg_src] <= t_neg).all())
        seen_neg += 24
    assert seen_neg == 24 * 8
