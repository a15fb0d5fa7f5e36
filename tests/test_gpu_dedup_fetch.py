"""GPU: the call group's feature fetch through its DISTINCT rows (bench.py's headline fetch since round 6):
``wgamd_unique_bounded_live`` over the capacity-sized node list of a no-sync walk (live length on the device, no host sync),
one gather of the distinct rows, layer 1 reading them through the inverse index — bit for bit the layer over ``x = feat[n_id]``
gathered row for row (the same rows are summed in the same order)."""
import numpy as np
import pytest

from graphgen import powerlaw_csr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["int64", "int32"])
def test_unique_bounded_live_matches_the_synchronous_form(hiplib, dtype):
    import torch
    from wholegraph_amd.tensor import unique_bounded, unique_bounded_nosync
    g = torch.Generator(device="cuda").manual_seed(3)
    bound, cap, live = 5000, 40000, 23456
    ids = torch.randint(0, bound, (cap,), generator=g, device="cuda").to(getattr(torch, dtype))
    ids[:live:97] = -1                                       # rows to skip
    ids[live:] = bound + 7                                   # capacity slack: garbage that must never be looked at
    n_live = torch.tensor([live], dtype=torch.int32, device="cuda")
    distinct, inverse, info = unique_bounded_nosync(ids, n_live, bound)
    n_d, bad = info.tolist()
    want_d, want_inv = unique_bounded(ids[:live].contiguous(), bound)
    assert bad == 0 and n_d == want_d.shape[0]
    assert torch.equal(distinct[:n_d], want_d) and torch.equal(inverse[:live], want_inv)
    ok = ids[:live] >= 0
    assert torch.equal(distinct[inverse[:live][ok].long()], ids[:live][ok].long())


def test_layer_through_the_distinct_rows_equals_the_layer_over_gathered_rows(hiplib):
    import torch
    from wholegraph_amd import nn
    from wholegraph_amd.fused import NoSyncWalk
    from wholegraph_amd.tensor import local_gather, unique_bounded_nosync
    V, F, G, B = 30000, 100, 6, 128
    row_ptr, col = powerlaw_csr(V, 16, seed=2, col_dtype=np.int64, max_deg=2000)
    feat = torch.from_numpy(np.random.default_rng(1).standard_normal((V, F)).astype(np.float32)).cuda()
    walk = NoSyncWalk(torch.from_numpy(row_ptr).cuda(), torch.from_numpy(col).cuda(), B, [10, 5], torch.int64, G, pad_unique=False)
    seeds = torch.randint(0, V, (G * B,), generator=torch.Generator(device="cuda").manual_seed(4), device="cuda")
    res = walk.run(seeds, [[7 + b for b in range(G)], [99 + b for b in range(G)]])
    distinct, inverse, info = unique_bounded_nosync(res.unique[1], res.counts[1][1:2], V)
    n_d, bad = info.tolist()
    (e0, u0), (e1, u1) = res.counts.tolist()
    n_id = res.unique[1][:u1]
    assert bad == 0 and n_d == int(torch.unique(n_id).numel()) and n_d < u1          # batches overlap: repeats exist
    x_full = local_gather(feat, n_id, torch.empty((u1, F), device="cuda"))
    x_dist = local_gather(feat, distinct[:n_d], torch.empty((n_d, F), device="cuda"))
    assert torch.equal(x_dist[inverse[:u1].long()], x_full)
    conv = nn.SAGEConv(F, 256).cuda()
    w_t = conv._weight_t()
    rows = res.target_rows_in_unique(1, u0)
    ptr, nbr = res.offsets[1][:u0 + 1], res.neighbor_row[1][:e1]
    a = nn.sage_layer_fused_forward(ptr, nbr, x_full, rows, w_t, conv.lin_l.bias, relu=True, mean=True)
    b = nn.sage_layer_fused_forward(ptr, nbr, x_dist, rows, w_t, conv.lin_l.bias, relu=True, mean=True, src_ids=inverse[:u1])
    assert torch.equal(a, b)
    c = nn.sage_aggregate_forward(ptr, nbr, x_full, rows, True)
    d = nn.sage_aggregate_fetch_forward(ptr, nbr, x_dist, inverse[:u1], rows, True)
    assert torch.equal(c, d)
