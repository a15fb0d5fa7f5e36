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
