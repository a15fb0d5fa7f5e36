"""The cugraph_pyg-shaped stack with TWO ranks on one GPU (gloo backend, both ranks on cuda:0 — RCCL refuses two ranks on
one device; the row kernels, samplers and call groups are the HIP ones): every rank contributes a slice of the edges and a
slice of the features (graph_store.py / feature_store.py multi-GPU contract), iterates its own seed shard through
NeighborLoader in call groups, and every batch must carry the features and edges of the GLOBAL graph — the per-group
attribute fetch is one all-to-all exchange per call group here."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent(r"""
    import os, sys
    sys.path[:0] = [sys.argv[1], sys.argv[1] + "/cugraph-gnn_amd"]
    import torch, torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    g = torch.Generator().manual_seed(0)
    n, m, B = 4000, 60000, 64
    ei = torch.stack([torch.randint(0, n, (m,), generator=g), torch.randint(0, n, (m,), generator=g)])
    x = torch.randn(n, 12, generator=g)
    y = torch.arange(n)
    cut_e, cut_n = (0, 25000, m), (0, 1500, n)                       # uneven slices, rank order = global id order
    gs, fs = GraphStore(), FeatureStore()
    gs.put_edge_index(ei[:, cut_e[rank]:cut_e[rank + 1]].cuda(), ("n", "e", "n"), "coo", False, (n, n))
    fs["n", "x", None] = x[cut_n[rank]:cut_n[rank + 1]].cuda()
    fs["n", "y", None] = y[cut_n[rank]:cut_n[rank + 1]].cuda()
    assert gs.is_multi_gpu
    seeds = torch.randperm(n, generator=g)[:B * 8].view(world, -1)[rank].cuda()     # 4 batches per rank
    loader = NeighborLoader((fs, gs), [5, 3], input_nodes=seeds, batch_size=B, local_seeds_per_call=B * 2, shuffle=False,
                            random_state=9)
    edge_set = set((ei[0] * n + ei[1]).tolist())
    nb = 0
    for batch in loader:
        nid = batch.n_id.cpu()
        assert torch.equal(batch.x.cpu(), x[nid]) and torch.equal(batch.y.cpu(), y[nid])          # global features
        assert torch.equal(nid[:B], seeds[nb * B:(nb + 1) * B].cpu())                             # seeds first
        src, dst = nid[batch.edge_index[0].cpu()], nid[batch.edge_index[1].cpu()]
        assert all(int(k) in edge_set for k in (src * n + dst).tolist())                          # real edges, all slices
        eid = batch.e_id.cpu()
        assert torch.equal(ei[0][eid], src) and torch.equal(ei[1][eid], dst)                      # global edge ids
        nb += 1
    assert nb == 4
    dist.barrier()
    print("RANK_OK", rank)
    dist.destroy_process_group()
""")


def test_neighbor_loader_two_ranks_one_gpu(hiplib, tmp_path):
    worker = tmp_path / "loader_worker.py"
    worker.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29537", str(worker), ROOT]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0 and p.stdout.count("RANK_OK") == 2, p.stdout[-2000:] + p.stderr[-4000:]


UNEVEN_WORKER = textwrap.dedent(r"""
    import os, sys, warnings
    sys.path[:0] = [sys.argv[1], sys.argv[1] + "/cugraph-gnn_amd"]
    import torch, torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import LinkNeighborLoader, NeighborLoader
    g = torch.Generator().manual_seed(0)
    n, m, B = 4000, 60000, 64
    ei = torch.stack([torch.randint(0, n, (m,), generator=g), torch.randint(0, n, (m,), generator=g)])
    x = torch.randn(n, 12, generator=g)
    cut_e, cut_n = (0, 25000, m), (0, 1500, n)
    gs, fs = GraphStore(), FeatureStore()
    gs.put_edge_index(ei[:, cut_e[rank]:cut_e[rank + 1]].cuda(), ("n", "e", "n"), "coo", False, (n, n))
    fs["n", "x", None] = x[cut_n[rank]:cut_n[rank + 1]].cuda()
    # UNEVEN seed shards: rank 0 gets 3 batches (2 call groups of <= 2), rank 1 gets 5 full batches + a ragged sixth
    # (3 call groups + 1 batch outside a group).  Every feature fetch is an all-to-all over both ranks: without the
    # cross-rank padding rank 1's third group fetch and its ragged batch's fetch would wait for rank 0 forever.
    perm = torch.randperm(n, generator=g)
    seeds = (perm[:3 * B] if rank == 0 else perm[1000:1000 + 5 * B + 17]).cuda()
    want = 3 if rank == 0 else 6
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        loader = NeighborLoader((fs, gs), [5, 3], input_nodes=seeds, batch_size=B, local_seeds_per_call=B * 2, shuffle=False,
                                random_state=9)
        nb = 0
        for batch in loader:
            nid = batch.n_id.cpu()
            assert torch.equal(batch.x.cpu(), x[nid])
            nb += 1
        assert nb == want, (rank, nb)
        if rank == 0:   # the reference's warning about uneven inputs (distributed_sampler.py:205-214), rank 0 only
            assert any("same number of batches" in str(x_.message) for x_ in w), [str(x_.message) for x_ in w]
    dist.barrier()
    # the same with edge seeds (link prediction, negatives): 2 vs 4 batches + ragged
    eli = ei[:, (torch.arange(2 * B) if rank == 0 else torch.arange(3000, 3000 + 4 * B + 9))].cuda()
    link = LinkNeighborLoader((fs, gs), [4, 2], edge_label_index=eli, batch_size=B, neg_sampling="binary", shuffle=False,
                              random_state=3, local_seeds_per_call=4 * B * 2)
    nb = 0
    for batch in link:
        assert torch.equal(batch.x.cpu(), x[batch.n_id.cpu()])
        nb += 1
    assert nb == (2 if rank == 0 else 5), (rank, nb)
    # heterogeneous store (two node types, two relations), seeds of type "a": 2 vs 4 batches + ragged
    hs, hf = GraphStore(), FeatureStore()
    na, nb_ = 3000, 2000
    e_ab = torch.stack([torch.randint(0, na, (30000,), generator=g), torch.randint(0, nb_, (30000,), generator=g)])
    e_ba = torch.stack([torch.randint(0, nb_, (30000,), generator=g), torch.randint(0, na, (30000,), generator=g)])
    half = 15000
    sl = slice(0, half) if rank == 0 else slice(half, 30000)
    hs.put_edge_index(e_ab[:, sl].cuda(), ("a", "to", "b"), "coo", False, (na, nb_))
    hs.put_edge_index(e_ba[:, sl].cuda(), ("b", "back", "a"), "coo", False, (nb_, na))
    xa, xb = torch.randn(na, 6, generator=g), torch.randn(nb_, 5, generator=g)
    ca, cb = (0, 1200, na), (0, 900, nb_)
    hf["a", "x", None] = xa[ca[rank]:ca[rank + 1]].cuda()
    hf["b", "x", None] = xb[cb[rank]:cb[rank + 1]].cuda()
    hseeds = (torch.arange(2 * B) if rank == 0 else torch.arange(500, 500 + 4 * B + 5)).cuda()
    hl = NeighborLoader((hf, hs), {("a", "to", "b"): [3, 2], ("b", "back", "a"): [3, 2]}, input_nodes=("a", hseeds), batch_size=B,
                        shuffle=False, random_state=5, local_seeds_per_call=2 * B)
    nb = 0
    for batch in hl:
        assert torch.equal(batch["a"].x.cpu(), xa[batch["a"].n_id.cpu()]) and torch.equal(batch["b"].x.cpu(), xb[batch["b"].n_id.cpu()])
        nb += 1
    assert nb == (2 if rank == 0 else 5), (rank, nb)
    # defaults: the call group is sized from device memory (>= 64 mini-batches of 1024 seeds for fan-out [25, 10] on 288 GB)
    from cugraph_pyg_amd.sampler.sampler import default_local_seeds_per_call
    assert default_local_seeds_per_call([25, 10], 1024) >= 64 * 1024
    dist.barrier()
    print("RANK_OK", rank)
    dist.destroy_process_group()
""")


def test_uneven_seed_shards_are_padded_across_ranks(hiplib, tmp_path):
    """3 vs 5(+1 ragged) batches per rank over a PARTITIONED FeatureStore: the loaders MAX-reduce their fetch plan and the
    rank that runs out first makes empty fetches (reference: call-group padding + uneven-batch warning,
    sampler/distributed_sampler.py:200-214,305-329)."""
    worker = tmp_path / "uneven_worker.py"
    worker.write_text(UNEVEN_WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29539", str(worker), ROOT]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0 and p.stdout.count("RANK_OK") == 2, p.stdout[-2000:] + p.stderr[-4000:]
