"""The reference's only exact heterogeneous sampler fixture, mirrored at the HeteroSamplerOutput level
(/root/reference/python/cugraph-pyg/cugraph_pyg/tests/sampler/test_distributed_sampler.py:20-150):
14-edge multigraph, two vertex types, two edge types, every hop takes ALL neighbours, so the sample is deterministic and
the expected per-type / per-hop edge ids and endpoints are reference-held values (tests/golden/hetero_distributed_sampler.json).

cuGraph samples along src -> dst; cugraph_pyg stores PyG edges reversed (graph_store.py:508-539), so a cuGraph edge
(s -> d) is the PyG edge (d -> s) whose destination s is the vertex being expanded.  Ids are type-local in our stores."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu
FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hetero_distributed_sampler.json")


def _fixture():
    import torch
    from cugraph_pyg_amd.data import GraphStore
    with open(FIX) as f:
        fx = json.load(f)
    off = fx["vertex_type_offsets"]
    names = ["a", "b"]                                           # sorted type names = cuGraph's type order

    def vtype(v):
        return 0 if v < off[1] else 1

    gs = GraphStore()
    et_of = {}
    for t in sorted(set(fx["etps"])):
        sel = [i for i, e in enumerate(fx["etps"]) if e == t]
        s, d = [fx["srcs"][i] for i in sel], [fx["dsts"][i] for i in sel]
        assert [fx["eids"][i] for i in sel] == list(range(len(sel)))   # edge ids = position inside the type
        st, dt = vtype(s[0]), vtype(d[0])
        # PyG edge = (cuGraph dst) -> (cuGraph src); the relation name keeps the cuGraph edge-type order when sorted
        et = (names[dt], "r%d" % t, names[st])
        pyg_src = torch.tensor([v - off[dt] for v in d])
        pyg_dst = torch.tensor([v - off[st] for v in s])
        gs[et, "coo", False, (off[dt + 1] - off[dt], off[st + 1] - off[st])] = [pyg_src, pyg_dst]
        et_of[str(t)] = (et, st, dt)
    return fx, gs, et_of, off, names


def _check(fx, et_of, off, names, node, row, col, edge, num_edges):
    for t, (et, st, dt) in et_of.items():
        lo = 0
        for h in range(fx["hops"]):
            want = fx["expected"][t][h]
            n = num_edges[et][h]
            assert n == len(want["edge_id"]), (t, h, n)
            r, c, e = row[et][lo:lo + n], col[et][lo:lo + n], edge[et][lo:lo + n]
            lo += n
            assert sorted(e.tolist()) == want["edge_id"]
            # row = neighbour end (cuGraph dst side), col = expanded vertex (cuGraph src side)
            assert sorted((node[names[st]][c.long()] + off[st]).tolist()) == want["src"]
            assert sorted((node[names[dt]][r.long()] + off[dt]).tolist()) == want["dst"]
            # and edge by edge: the stored edge with that id really joins the two endpoints
            sel = [i for i, x in enumerate(fx["etps"]) if str(x) == t]
            for k in range(n):
                i = sel[int(e[k])]
                assert fx["srcs"][i] == int(node[names[st]][int(c[k])]) + off[st]
                assert fx["dsts"][i] == int(node[names[dt]][int(r[k])]) + off[dt]
        assert lo == row[et].shape[0]


@pytest.mark.parametrize("fan", [-1, 8])
def test_reference_hetero_fixture_one_batch_path(hiplib, fan):
    """fan-out -1 (take all) and a fan-out above every degree both give the reference's exact sample."""
    import torch
    from cugraph_pyg_amd.sampler.sampler import hetero_neighbor_sample
    fx, gs, et_of, off, names = _fixture()
    graphs = gs._hetero_graphs
    fanout = {et: [fan] * fx["hops"] for et, _, _ in et_of.values()}
    seeds = torch.tensor([v - off[1] for v in fx["seeds"]], device="cuda")
    node, row, col, edge, num_nodes, num_edges = hetero_neighbor_sample(graphs, names[1], seeds, fanout, 62)
    cpu = lambda d: {k: v.cpu() for k, v in d.items()}  # noqa: E731
    _check(fx, et_of, off, names, cpu(node), cpu(row), cpu(col), cpu(edge), num_edges)
    assert node[names[1]][:2].cpu().tolist() == [0, 1]           # seeds first (retain_seeds)
    assert num_nodes[names[1]][0] == 2 and num_nodes[names[0]][0] == 0


def test_reference_hetero_fixture_call_group_path(hiplib):
    """The same fixture through the call-group (no-sync) heterogeneous walk that the loaders run on."""
    import torch
    from cugraph_pyg_amd.sampler.sampler import HeteroNeighborSampler
    fx, gs, et_of, off, names = _fixture()
    fanout = {et: [8] * fx["hops"] for et, _, _ in et_of.values()}
    nv = {names[0]: off[1] - off[0], names[1]: off[2] - off[1]}
    smp = HeteroNeighborSampler(gs._hetero_graphs, fanout, local_seeds_per_call=2, num_nodes=nv)
    seeds = torch.tensor([v - off[1] for v in fx["seeds"]], device="cuda")
    outs = list(smp.sample_batches(names[1], seeds, 2, 62))
    assert len(outs) == 1
    node, row, col, edge, num_nodes, num_edges = outs[0][1]
    cpu = lambda d: {k: v.cpu() for k, v in d.items()}  # noqa: E731
    _check(fx, et_of, off, names, cpu(node), cpu(row), cpu(col), cpu(edge), num_edges)
