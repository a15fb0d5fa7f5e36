"""BASELINE configs[4] pipeline (bench.py --workload mag -> bench_mag.py) on a small heterogeneous graph: the call-group GPU
path — HeteroPygWalk, per-type feature gather, two trimmed aggregate-first HeteroConv(GATConv 4x64) layers through
wgamd_gat_aggregate_heads_f32 — against
the SAME composition on the C oracle + torch CPU GEMMs, mini-batch by mini-batch (`cpu_port_batch`, which is also what
`cpu_baseline` times).  Sampling is bit-exact, so both sides aggregate identical subgraphs; outputs agree to fp32 accuracy."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_gat_rows_kernel_matches_oracle_with_indirection_and_accumulation(oracle_mod, hiplib):
    import torch
    from wholegraph_amd import nn
    rng = np.random.default_rng(3)
    H, C, n_src, n_all, n_rows = 4, 64, 3000, 5000, 1200
    deg = np.minimum(rng.zipf(1.5, n_rows), 70)
    deg[::9] = 0
    rp = np.zeros(n_rows + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    col = rng.integers(0, n_src, rp[-1]).astype(np.int32)
    x = rng.standard_normal((n_src, H * C)).astype(np.float32)
    a_s = rng.standard_normal((n_src, H)).astype(np.float32)
    a_d = rng.standard_normal((n_all, H)).astype(np.float32)
    rows = rng.permutation(n_all)[:n_rows].astype(np.int64)          # the launch's rows inside a larger destination list
    base = rng.standard_normal((n_all, H * C)).astype(np.float32)
    ref, _ = oracle_mod.gat_csr(rp, col, x.reshape(-1, H, C), a_s, np.ascontiguousarray(a_d[rows]))
    cu = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    for accumulate in (False, True):
        out = cu(base.copy())
        nn.gat_forward_rows(cu(rp), cu(col), cu(x), cu(a_s), cu(a_d), H, out, cu(rows), accumulate=accumulate)
        want = base.copy()
        want[rows] = (want[rows] if accumulate else 0) + ref.reshape(n_rows, H * C)
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=2e-6)
    # without indirection it is wgamd_gat_csr_f32
    out = torch.empty((n_rows, H * C), device="cuda")
    nn.gat_forward_rows(cu(rp), cu(col), cu(x), cu(a_s), cu(np.ascontiguousarray(a_d[rows])), H, out)
    np.testing.assert_allclose(out.cpu().numpy(), ref.reshape(n_rows, H * C), rtol=1e-5, atol=2e-6)
    got2, _ = nn.gat_forward(cu(rp), cu(col), cu(x), cu(a_s), cu(np.ascontiguousarray(a_d[rows])), H, 0.2, need_alpha=False)
    assert torch.equal(got2, out)


@pytest.mark.parametrize("F,H", [(128, 4), (256, 4), (64, 1), (100, 2), (32, 8)])
def test_gat_aggregate_heads_matches_oracle_and_transform_first(oracle_mod, hiplib, F, H):
    """wgamd_gat_aggregate_heads_f32 vs the oracle's restatement, and aggregate-then-transform == GATConv's
    transform-then-aggregate (wgo_gat_csr on x @ W) to fp32 accuracy."""
    import torch
    from wholegraph_amd import nn
    rng = np.random.default_rng(F + H)
    C, n_src, n_all, n_rows = 16, 4000, 2500, 900
    deg = np.minimum(rng.zipf(1.4, n_rows), 150)
    deg[::7] = 0
    rp = np.zeros(n_rows + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    col = rng.integers(0, n_src, rp[-1]).astype(np.int32)
    x = rng.standard_normal((n_src, F)).astype(np.float32)
    a_s = rng.standard_normal((n_src, H)).astype(np.float32) * 2
    a_d = rng.standard_normal((n_all, H)).astype(np.float32) * 2
    rows = rng.permutation(n_all)[:n_rows].astype(np.int64)
    cu = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    ref = oracle_mod.gat_aggregate_heads(rp, col, x, a_s, a_d, dst_rows=rows)
    got = nn.gat_aggregate_heads(cu(rp), cu(col), cu(x), cu(a_s), cu(a_d), H, dst_rows=cu(rows))
    np.testing.assert_allclose(got.cpu().numpy().reshape(n_rows, H, F), ref, rtol=1e-5, atol=2e-6)
    got_plain = nn.gat_aggregate_heads(cu(rp), cu(col), cu(x), cu(a_s), cu(np.ascontiguousarray(a_d[rows])), H)
    assert torch.equal(got, got_plain)
    # aggregate-first == transform-first: GATConv's own order (lin over every source row, then the attention-weighted sum of
    # the TRANSFORMED rows) evaluated in float64 is the reference; the device's aggregate-then-transform must match it to
    # 1e-5 of the dot products' scale and, wherever the result is not a cancellation, to 1e-5 relative, element by element
    w = (rng.standard_normal((F, H * C)) / np.sqrt(F)).astype(np.float32)
    out = nn.gat_transform_heads(got, cu(w), H).cpu().numpy()
    xw = (x.astype(np.float64) @ w.astype(np.float64)).reshape(n_src, H, C)
    first = np.zeros((n_rows, H, C))
    for h in range(H):       # head h: softmax weights of head h applied to the head's own C transformed channels
        first[:, h, :] = oracle_mod.gat_aggregate_heads_f64(rp, col, xw[:, h, :], a_s[:, h:h + 1], a_d[:, h:h + 1], dst_rows=rows)[:, 0, :]
    first = first.reshape(n_rows, H * C)
    agg64 = oracle_mod.gat_aggregate_heads_f64(rp, col, np.abs(x), a_s, a_d, dst_rows=rows)          # sum alpha |x|: the scale
    scale = np.einsum("nhf,fhc->nhc", agg64, np.abs(w.astype(np.float64)).reshape(F, H, C)).reshape(n_rows, H * C)
    assert np.all(np.abs(out - first) <= 1e-5 * scale + 1e-7), np.abs(out - first).max()
    big = (np.abs(first) >= 0.1 * scale) & (scale > 0)        # (rows without edges are exact zeros on both sides)
    assert big.any() and (np.abs(out - first)[big] / np.abs(first)[big]).max() <= 1e-5
    acc = torch.ones((n_rows, H * C), device="cuda")
    nn.gat_transform_heads(got, cu(w), H, out=acc)
    np.testing.assert_allclose(acc.cpu().numpy(), out + 1.0, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("fanout", [10, 25])
def test_gat_kernels_read_the_table_through_an_id_list(hiplib, fanout):
    """Fetch in the layer for GAT relations: wgamd_gat_aggregate_heads_ids_f32 / wgamd_gat_layer_fused_ids_bf16x3 over
    (feature table, node list) == the same kernels over the gathered rows, bit for bit (same loads, same order), and the
    attention terms of the listed rows without a row output == the terms the gather produces."""
    import torch
    from wholegraph_amd import nn
    g = torch.Generator(device="cuda").manual_seed(fanout)
    F, H, C, n_table, n_src, n_rows = 128, 4, 64, 50000, 7000, 1500
    table = torch.randn((n_table, F), generator=g, device="cuda")
    ids = torch.randint(0, n_table, (n_src,), generator=g, device="cuda")
    ids[:2] = torch.tensor([0, n_table - 1], device="cuda")
    deg = torch.randint(0, fanout + 1, (n_rows,), generator=g, device="cuda")
    deg[::11] = 0
    deg[5] = 3 * fanout                      # past the register window of the one-kernel relation
    rp = torch.zeros(n_rows + 1, dtype=torch.int32, device="cuda")
    rp[1:] = torch.cumsum(deg, 0)
    col = torch.randint(0, n_src, (int(rp[-1]),), generator=g, device="cuda", dtype=torch.int32)
    v = torch.randn((F, 8), generator=g, device="cuda") * 0.3
    dst_rows = torch.randperm(n_src, generator=g, device="cuda")[:n_rows].contiguous()
    x, slabs = nn.gather_with_terms(table, ids, v, heads=4)
    lazy_slabs = nn.lazy_rows_terms(table, ids, v, heads=4)
    assert torch.equal(x, table[ids]) and torch.equal(slabs, lazy_slabs)
    # ... and == the terms of the TABLE's rows gathered through the list (what a call group longer than the table takes)
    table_slabs = nn.rows_terms(table, v, heads=4)
    assert torch.equal(nn.gather_term_slabs(table_slabs, ids), slabs) and torch.equal(nn.gather_term_slabs(table_slabs, ids.int()), slabs)
    holes = ids.clone()
    holes[::5] = -1
    assert torch.equal(nn.gather_term_slabs(table_slabs, holes), nn.lazy_rows_terms(table, holes, v, heads=4))
    a_src, a_dst = slabs[0], slabs[1]
    agg = nn.gat_aggregate_heads(rp, col, x, a_src, a_dst, H, dst_rows=dst_rows)
    agg_ids = nn.gat_aggregate_heads(rp, col, table, a_src, a_dst, H, dst_rows=dst_rows, src_ids=ids)
    assert torch.equal(agg, agg_ids)
    w = torch.randn((F, H * C), generator=g, device="cuda") / F ** 0.5
    bias = torch.randn(H * C, generator=g, device="cuda")
    acc = torch.randn((n_rows, H * C), generator=g, device="cuda")
    out = nn.gat_layer_fused(rp, col, x, a_src, a_dst, w, H, dst_rows=dst_rows, acc_in=acc, bias=bias, relu=True)
    out_ids = nn.gat_layer_fused(rp, col, table, a_src, a_dst, w, H, dst_rows=dst_rows, acc_in=acc, bias=bias, relu=True, src_ids=ids)
    assert torch.equal(out, out_ids)
    # the attention terms of the TABLE's rows, read through the lists (either end, both): the per-list terms are never made
    ts, td = table_slabs[0], table_slabs[1]
    for kw in (dict(src_terms_by_id=True), dict(dst_ids=ids, dst_terms_by_id=True), dict(src_terms_by_id=True, dst_ids=ids, dst_terms_by_id=True)):
        s_, d_ = (ts if kw.get("src_terms_by_id") else a_src), (td if kw.get("dst_terms_by_id") else a_dst)
        assert torch.equal(nn.gat_aggregate_heads(rp, col, table, s_, d_, H, dst_rows=dst_rows, src_ids=ids, **kw), agg)
        assert torch.equal(nn.gat_layer_fused(rp, col, table, s_, d_, w, H, dst_rows=dst_rows, acc_in=acc, bias=bias, relu=True,
                                              src_ids=ids, **kw), out)
    with pytest.raises(AssertionError):
        nn.gat_aggregate_heads(rp, col, table, a_src, a_dst, H, dst_rows=dst_rows, src_ids=ids.int())


def _agg_heads_reference(x, a_src, a_dst, rp, col, H, dst_rows, slope=0.2):
    """agg[i, h, :] = sum_e softmax_e(leaky(a_src[col e, h] + a_dst[dst i, h])) x[col e, :] with torch index ops (autograd)."""
    import torch
    n = rp.shape[0] - 1
    deg = (rp[1:] - rp[:-1]).long()
    row = torch.repeat_interleave(torch.arange(n, device=x.device), deg)
    c = col.long()
    sc = torch.nn.functional.leaky_relu(a_src[c] + a_dst[dst_rows][row], slope)                      # [E, H]
    mx = torch.full((n, H), -float("inf"), dtype=sc.dtype, device=x.device).index_reduce_(0, row, sc, "amax", include_self=True)
    ex = torch.exp(sc - mx[row])
    den = torch.zeros((n, H), dtype=sc.dtype, device=x.device).index_add_(0, row, ex)
    alpha = ex / den[row]
    msg = alpha.unsqueeze(-1) * x[c].unsqueeze(1)                                                    # [E, H, F]
    return torch.zeros((n, H, x.shape[1]), dtype=x.dtype, device=x.device).index_add_(0, row, msg).reshape(n, -1)


@pytest.mark.parametrize("F,H,mode", [(128, 4, "plain"), (256, 4, "plain"), (128, 4, "ids"), (128, 4, "by_id"), (64, 2, "by_id_src"),
                                      (100, 1, "plain")])
def test_gat_aggregate_heads_backward_matches_float64_autograd(hiplib, F, H, mode):
    """wgamd_gat_aggregate_heads_bwd_f32 through nn._GatAggregateHeads: gradients of the attention terms (per listed row, or per
    TABLE row — a table row collects from every place the list names it) and of the source rows against float64 autograd of
    the same formula, held to the SAGE side's contract: |err| <= 1e-5 x the magnitude sum of the terms of an entry, and 1e-5
    relative on the entries that are not cancellations (round 6; rounds 4-5 used 2e-5 of the gradient's maximum)."""
    import torch
    from wholegraph_amd import nn
    g = torch.Generator(device="cuda").manual_seed(F + H)
    n_table, n_src, n_dst_list, n_rows = 3000, 5000, 2500, 900
    deg = torch.randint(0, 26, (n_rows,), generator=g, device="cuda"); deg[::7] = 0
    rp = torch.zeros(n_rows + 1, dtype=torch.int32, device="cuda"); rp[1:] = torch.cumsum(deg, 0)
    col = torch.randint(0, n_src, (int(rp[-1]),), generator=g, device="cuda", dtype=torch.int32)
    dst_rows = torch.randperm(n_dst_list, generator=g, device="cuda")[:n_rows].contiguous()
    gout = torch.randn((n_rows, H * F), generator=g, device="cuda")
    lazy = mode != "plain"
    table = torch.randn((n_table if lazy else n_src, F), generator=g, device="cuda")
    ids = torch.randint(0, n_table, (n_src,), generator=g, device="cuda") if lazy else None            # (with repeats)
    dids = torch.randint(0, n_table, (n_dst_list,), generator=g, device="cuda") if lazy else None
    src_by_id, dst_by_id = mode in ("by_id", "by_id_src"), mode == "by_id"
    a_src = (torch.randn((n_table if src_by_id else n_src, H), generator=g, device="cuda") * 2).requires_grad_(True)
    a_dst = (torch.randn((n_table if dst_by_id else n_dst_list, H), generator=g, device="cuda") * 2).requires_grad_(True)
    x_grad = (not lazy) or mode == "ids"     # plain: the transposed-hop kernel; "ids": a TABLE that asks for its gradient (atomics)
    x = table.clone().requires_grad_(x_grad)
    out = nn._GatAggregateHeads.apply(x, a_src, a_dst, rp, col, H, dst_rows, ids, dids if dst_by_id else None, src_by_id, dst_by_id, 0.2)
    out.backward(gout)
    # float64 reference: the list-level quantities spelled out, gradients flow back to the table-level leaves through indexing
    x64 = table.double().requires_grad_(x_grad)
    s64, d64 = a_src.detach().double().requires_grad_(True), a_dst.detach().double().requires_grad_(True)
    x_list = x64[ids] if lazy else x64
    s_list = s64[ids] if src_by_id else s64
    d_list = d64[dids] if dst_by_id else d64
    ref = _agg_heads_reference(x_list, s_list, d_list, rp, col, H, dst_rows)
    ref.backward(gout.double())
    assert float((out.double() - ref.detach()).abs().max()) <= 1e-5 * float(ref.detach().abs().max())
    # The contract of tests/test_gpu_sage_train.py (north_star's 1e-5): |err| <= 1e-5 x the MAGNITUDE SUM of the terms a gradient
    # entry is made of, and 1e-5 relative on the entries that are not cancellations.  The magnitude sums, in float64:
    #   p_e^h = <g_i^h, x_e> has magnitude P = sum_f |g||x|;  ds = alpha (p - sum_k alpha_k p_k) has alpha (P + sum alpha P);
    #   ga_src / ga_dst add |de| = |ds| slope' over the edges of a term row;  gx adds sum_h alpha |g_i^h| over a row's edges.
    with torch.no_grad():
        n = rp.shape[0] - 1
        deg = (rp[1:] - rp[:-1]).long()
        row = torch.repeat_interleave(torch.arange(n, device="cuda"), deg)
        c = col.long()
        xl, sl, dl = x_list.detach(), s_list.detach(), d_list.detach()
        raw = sl[c] + dl[dst_rows][row]
        sc = torch.nn.functional.leaky_relu(raw, 0.2)
        mx = torch.full((n, H), -float("inf"), dtype=torch.float64, device="cuda").index_reduce_(0, row, sc, "amax", include_self=True)
        ex = torch.exp(sc - mx[row])
        alpha = ex / torch.zeros((n, H), dtype=torch.float64, device="cuda").index_add_(0, row, ex)[row]
        g3 = gout.double().view(n, H, F)
        P = (g3[row].abs() * xl[c].abs().unsqueeze(1)).sum(-1)                                              # [E, H]
        de_mag = alpha * (P + torch.zeros((n, H), dtype=torch.float64, device="cuda").index_add_(0, row, alpha * P)[row])
        src_rows = (ids[c] if src_by_id else c)
        dst_term_rows = (dids[dst_rows] if dst_by_id else dst_rows)[row]
        mag_src = torch.zeros_like(s64).index_add_(0, src_rows, de_mag)
        mag_dst = torch.zeros_like(d64).index_add_(0, dst_term_rows, de_mag)
        mag_x = torch.zeros_like(x64).index_add_(0, ids[c] if lazy else c, (alpha.unsqueeze(-1) * g3[row].abs()).sum(1)) if x_grad else None

    def close(got, want, mag, what):
        err = (got.double() - want).abs()
        assert bool((err <= 1e-5 * mag + 1e-9).all()), (what, float((err - 1e-5 * mag).max()))
        big = want.abs() >= 0.1 * mag
        big &= mag > 0
        if int(big.sum()) > 0:
            assert float((err[big] / want.abs()[big]).max()) <= 1e-5, what
    close(a_src.grad, s64.grad, mag_src, "ga_src")
    close(a_dst.grad, d64.grad, mag_dst, "ga_dst")
    if x_grad:
        close(x.grad, x64.grad, mag_x, "gx")


@pytest.mark.parametrize("n,F,T,x_grad", [(50000, 128, 12, False), (7777, 100, 8, False), (3001, 256, 32, True), (999, 64, 1, True),
                                            (70000, 40, 20, False)])
def test_narrow_terms_function_matches_float64(hiplib, n, F, T, x_grad):
    """nn._NarrowTerms: terms = x @ v for a long x and a narrow v, dv = x^T @ dterms by wgamd_rows_terms_bwd_f32 (float atomics),
    dx by a library product — against float64."""
    import torch
    from wholegraph_amd import nn
    g = torch.Generator(device="cuda").manual_seed(n + T)
    x = torch.randn((n, F), generator=g, device="cuda").requires_grad_(x_grad)
    v = (torch.randn((F, T), generator=g, device="cuda") * 0.3).requires_grad_(True)
    gout = torch.randn((n, T), generator=g, device="cuda")
    out = nn._NarrowTerms.apply(x, v)
    out.backward(gout)
    x64, v64 = x.detach().double().requires_grad_(x_grad), v.detach().double().requires_grad_(True)
    ref = x64 @ v64
    ref.backward(gout.double())
    assert float((out.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    # (a sum of n products per element: compare at the scale of the sum of their magnitudes)
    scale = (x.detach().double().abs().t() @ gout.double().abs())
    assert bool(((v.grad.double() - v64.grad).abs() <= 1e-5 * scale + 1e-9).all()), float((v.grad.double() - v64.grad).abs().max())
    if x_grad:
        assert float((x.grad.double() - x64.grad).abs().max()) <= 1e-5 * float(x64.grad.abs().max())


@pytest.mark.parametrize("tails", ["library", "split_k"])
def test_hetero_conv_trains_aggregate_first_like_relation_by_relation(hiplib, tails, monkeypatch):
    """nn.HeteroConv over a heterogeneous call group under autograd: the aggregate-first route (_forward_layer_train: lazy x read
    through the node lists, terms of the tables' rows) gives the outputs and the parameter gradients of PyG's relation-by-relation
    formulation on GATConv (_forward_relations).  ``split_k``: the dense tails on nn._HeadsTransform whatever the row count — the
    products per head accumulated into the relations' running sum in place, the weight gradients on the split-K kernel (the route
    call groups of production size take)."""
    import torch
    import bench_mag as bm
    from wholegraph_amd import nn as wnn
    monkeypatch.setattr(wnn, "_HEADS_WGRAD_MIN_ROWS", 0 if tails == "split_k" else 1 << 60)
    dev = torch.device("cuda", 0)
    nodes = {"paper": 3000, "author": 4000, "institution": 200, "field_of_study": 500}
    rels = {k: max(v // 400, 1500) for k, v in bm.MAG_RELS.items()}
    graphs, num_nodes = bm.build_mag_like(dev, nodes, rels, seed=9)
    etypes, ntypes = sorted(graphs), sorted(num_nodes)
    g = torch.Generator(device=dev).manual_seed(2)
    tables = {t: torch.rand((num_nodes[t], bm.F_IN), generator=g, device=dev) * 2 - 1 for t in ntypes}
    model = bm.build_model(bm.make_params(etypes, ntypes, dev), etypes, ntypes, dev)
    params = [p for m in model for p in m.parameters()]
    for p in params:
        p.requires_grad_(True)
    B, G = 128, 4
    seeds = torch.randperm(num_nodes["paper"], generator=g, device=dev)[:B * G]
    grp = next(iter(bm.make_loader(bm.build_mag_like.graph_store, tables, seeds, B, G).call_groups()))
    gout = torch.randn((B * G, bm.HC), generator=g, device=dev)
    results = []
    for agg_first in (True, False):
        for m in model:
            m.train_aggregate_first = agg_first
        for p in params:
            p.grad = None
        h = grp.x_dict          # (no ReLU between the layers: a pre-activation within rounding of zero would flip between two
        for j, layer in enumerate(model):      # fp32 formulations and move whole rows in and out of the gradient sums)
            h = layer(h, grp.layer_graph(j), act=None)
        out = h["paper"]
        out.backward(gout)
        results.append((out.detach().clone(), [None if p.grad is None else p.grad.clone() for p in params]))
    (o1, g1), (o2, g2) = results
    assert float((o1 - o2).abs().max()) <= 2e-5 * float(o2.abs().max())
    # mixed inputs under autograd: one node type as gathered rows, the others lazy — the aggregate-first route gives the same bits
    for m in model:
        m.train_aggregate_first = True
    for resident in ("author", "paper"):
        h = {t: (v.materialize() if t == resident else v) for t, v in grp.x_dict.items()}
        for j, layer in enumerate(model):
            h = layer(h, grp.layer_graph(j), act=None)
        assert h["paper"].requires_grad and torch.equal(h["paper"].detach(), o1), resident
    assert sum(a is not None for a in g1) >= 20
    # ---- float64 end to end (round 6): both layers spelled out with torch index ops in PyG's transform-first order, autograd
    # in float64 from `gout` back to every parameter; the aggregate-first fp32 route is held to it (this replaces relying on the
    # fp32 relation-by-relation route below as the only witness)
    p64 = {id(p): p.detach().double().requires_grad_(True) for p in params}

    def layer64(layer, xs, graph):
        out = {t: torch.zeros((n, bm.HC), dtype=torch.float64, device=dev) for t, n in graph.n_out.items() if n > 0}
        biased = set()
        for r in graph.relations:
            if r.n_rows == 0:
                continue
            c = layer.conv(r.edge_type)
            st, dt = r.edge_type[0], r.edge_type[2]
            W = p64[id(c.lin.weight)].t()
            hs, hd = xs[st] @ W, xs[dt][r.dst_rows] @ W
            a_s = (hs.view(-1, bm.HEADS, bm.CH) * p64[id(c.att_src)].view(1, bm.HEADS, bm.CH)).sum(-1)
            a_d = (hd.view(-1, bm.HEADS, bm.CH) * p64[id(c.att_dst)].view(1, bm.HEADS, bm.CH)).sum(-1)
            deg = (r.row_ptr[1:] - r.row_ptr[:-1]).long()
            row = torch.repeat_interleave(torch.arange(r.n_rows, device=dev), deg)
            cl = r.col.long()
            sc = torch.nn.functional.leaky_relu(a_s[cl] + a_d[row], c.negative_slope)
            mx = torch.full((r.n_rows, bm.HEADS), -float("inf"), dtype=torch.float64, device=dev).index_reduce_(0, row, sc, "amax", include_self=True)
            ex = torch.exp(sc - mx[row])
            alpha = ex / torch.zeros((r.n_rows, bm.HEADS), dtype=torch.float64, device=dev).index_add_(0, row, ex)[row]
            y = torch.zeros((r.n_rows, bm.HEADS, bm.CH), dtype=torch.float64, device=dev).index_add_(
                0, row, alpha.unsqueeze(-1) * hs.view(-1, bm.HEADS, bm.CH)[cl]).reshape(r.n_rows, bm.HC)
            if c.bias is not None:
                y = y + p64[id(c.bias)]
            rows = r.out_rows if r.out_rows is not None else torch.arange(r.n_rows, device=dev)
            out[dt] = out[dt].index_add(0, rows, y)
            biased.add((r.hop, dt))
        return out
    h64 = {t: (v.materialize() if hasattr(v, "materialize") else v).double() for t, v in grp.x_dict.items()}
    for j, layer in enumerate(model):
        h64 = layer64(layer, h64, grp.layer_graph(j))
    h64["paper"].backward(gout.double())
    assert float((o1.double() - h64["paper"].detach()).abs().max()) <= 1e-5 * float(h64["paper"].detach().abs().max())
    worst = 0.0
    for p, a in zip(params, g1):
        w = p64[id(p)].grad
        if a is None or w is None:
            assert (a is None or float(a.abs().max()) == 0.0) and (w is None or float(w.abs().max()) == 0.0)
            continue
        scale = float(w.abs().max())
        worst = max(worst, float((a.double() - w).abs().max()) / scale)
        # north_star's 1e-5, here of the gradient's own maximum (measured on this graph: <= 1.3e-6); the per-kernel bars are the
        # magnitude-sum tests above
        assert float((a.double() - w).abs().max()) <= 1e-5 * scale + 1e-12, (float((a.double() - w).abs().max()), scale, tuple(a.shape))
    print("hetero training route vs float64: worst relative-to-max gradient error %.2e" % worst)
    for a, b in zip(g1, g2):
        if a is None or b is None:        # a relation the seeds cannot see through the remaining layers: no gradient, or zeros
            assert (a is None or float(a.abs().max()) == 0.0) and (b is None or float(b.abs().max()) == 0.0)
            continue
        scale = float(b.abs().max())
        # (two fp32 formulations of sums over thousands of rows with cancellation: the aggregate-first route is held to float64
        #  above; this comparison only says the relation-by-relation route computes the same thing)
        assert float((a - b).abs().max()) <= 2e-3 * scale + 1e-7, (float((a - b).abs().max()), scale, tuple(a.shape))


def test_mag_pipeline_matches_cpu_port(oracle_mod, hiplib):
    """The config-5 path through the PACKAGE — GraphStore + FeatureStore -> NeighborLoader.call_groups() (HeteroCallGroup) ->
    2 x nn.HeteroConv{GATConv(., 64, heads=4)} — against the float64 composition on the C oracle, mini-batch by mini-batch."""
    import torch
    import bench_mag as bm
    dev = torch.device("cuda", 0)
    nodes = {"paper": 6000, "author": 9000, "institution": 300, "field_of_study": 800}
    rels = {k: max(v // 200, 2000) for k, v in bm.MAG_RELS.items()}
    graphs, num_nodes = bm.build_mag_like(dev, nodes, rels, seed=5)
    etypes, ntypes = sorted(graphs), sorted(num_nodes)
    g = torch.Generator(device=dev).manual_seed(1)
    tables = {t: torch.rand((num_nodes[t], bm.F_IN), generator=g, device=dev) * 2 - 1 for t in ntypes}
    params = bm.make_params(etypes, ntypes, dev)
    model = bm.build_model(params, etypes, ntypes, dev)
    B, G = 64, 4
    seeds = torch.randperm(num_nodes["paper"], generator=g, device=dev)[:B * G]
    loader = bm.make_loader(bm.build_mag_like.graph_store, tables, seeds, B, G)
    groups = list(loader.call_groups())
    assert len(groups) == 1
    grp = groups[0]
    with torch.no_grad():
        out = bm.forward_group(model, grp)
        assert all(layer.fetch_in_layer for layer in model)
        # x stays lazy by default (the relation kernels read the tables through the node lists); gathering first is the same bits
        for layer in model:
            layer.fetch_in_layer = False
        assert torch.equal(bm.forward_group(model, grp), out)
        for layer in model:
            layer.fetch_in_layer = True
        # MIXED inputs: one node type handed over as gathered rows, the others lazy (a resident source next to destinations whose
        # terms are those of the table's rows, and the other way round) — still the same bits
        for resident in ("author", "paper"):
            h = {t: (v.materialize() if t == resident else v) for t, v in grp.x_dict.items()}
            for j, layer in enumerate(model):
                h = layer(h, grp.layer_graph(j), act="relu")
            assert torch.equal(h["paper"], out), resident
        # the relation-by-relation route (GATConv modules, transform-first: what trains) computes the same layer
        h = grp.node_attr("x", lazy=False)
        for j, layer in enumerate(model):
            h = layer._forward_relations(h, grp.layer_graph(j), act="relu")
    torch.cuda.synchronize()
    edges = grp.num_edges
    launches = [r for j in range(2) for r in grp.layer_graph(j).relations]
    assert out.shape == (B * G, bm.HC) and edges > 0 and len(launches) >= 8
    assert float((h["paper"] - out).abs().max()) <= 2e-5 * float(out.abs().max())
    assert sum(sum(v) for v in grp.num_sampled_edges.values()) == edges
    fanout, hops = {et: [25, 10] for et in etypes}, 2

    class pipe:      # (names the checks below were written against)
        pass
    pipe.fanout, pipe.hops = fanout, hops
    hg = {et: (gr.row_ptr.cpu().numpy(), gr.col.cpu().numpy()) for et, gr in graphs.items()}
    tables_h = {t: v.cpu().numpy() for t, v in tables.items()}
    params_h = [dict(rel={et: {k: v.cpu() for k, v in w.items()} for et, w in p["rel"].items()},
                     bias={t: b.cpu() for t, b in p["bias"].items()}) for p in params]
    total = 0
    worst = 0.0
    for b in range(G):
        # the reference: the SAME computation with every floating-point step in float64 (bench_mag.cpu_port_batch(fp64=True):
        # numpy restatement of GATConv's formulas, oracle.gat_aggregate_heads_f64); sampling is bit-exact, so both sides
        # aggregate identical subgraphs.  north_star: 1e-5 relative for fp32 aggregation.
        ref, e = bm.cpu_port_batch(hg, tables_h, params_h, seeds[b * B:(b + 1) * B].cpu().numpy(), pipe.fanout, pipe.hops, etypes,
                                   ntypes, 7 + b, fp64=True)
        total += e
        got = out[b * B:(b + 1) * B].double().cpu().numpy()
        assert ref.dtype == np.float64
        scale = np.abs(ref).max(axis=1, keepdims=True)           # a seed's output row: 256 ReLU'd sums over its neighbourhood
        err = np.abs(got - ref)
        assert np.all(err <= 1e-5 * scale + 1e-7), (b, err.max(), float(scale.max()))
        big = np.abs(ref) >= 0.1 * scale                         # entries that are not cancellations: 1e-5 relative, literally
        assert big.any()
        worst = max(worst, float((err[big] / np.abs(ref)[big]).max()))
        # the fp32 CPU port (what cpu_baseline times) computes the same thing
        ref32, e32 = bm.cpu_port_batch(hg, tables_h, params_h, seeds[b * B:(b + 1) * B].cpu().numpy(), pipe.fanout, pipe.hops,
                                       etypes, ntypes, 7 + b)
        assert e32 == e and np.all(np.abs(ref32 - ref) <= 1e-5 * scale + 1e-7)
    assert worst <= 1e-5, worst
    assert total == edges        # bit-exact sampling: the same edges on both sides


@pytest.mark.parametrize("formulation", ["aggregate_first", "transform_first"])
def test_gat_layer_matches_frozen_fp64_golden(hiplib, formulation):
    """tests/golden/gat_layer_golden.npz: float64 expectations (390 sampled rows) of a GATConv layer at the shape of one launch
    of the ogbn-mag-like pipeline (F = 128 -> 4 x 64, hub sources, empty rows, rows a subset of a larger destination list),
    frozen by tests/golden/make_golden.py in GATConv's own order of operations.  Both device formulations stay inside
    north_star's 1e-5: the pipeline's aggregate-first one (attention terms x @ fold(W, att), wgamd_gat_aggregate_heads_f32
    over the untransformed rows, per-head transform, bias + ReLU) and the transform-first one (lin GEMM, wgamd_gat_csr_rows_f32)."""
    import hashlib
    import torch
    from graphgen import gat_layer_case
    from wholegraph_amd import nn
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "gat_layer_golden.npz"))
    rp, col, dst_rows, x, x_dst, w, att_s, att_d, bias = gat_layer_case()
    h = hashlib.sha256()
    for a in (rp, col, dst_rows, x, x_dst, w, att_s, att_d, bias):
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == str(z["inputs_sha256"]), "the regenerated inputs are not the ones the golden was frozen for"
    F, (H, C) = x.shape[1], att_s.shape
    n_rows = rp.size - 1
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    if formulation == "aggregate_first":
        v_s = (w.reshape(F, H, C) * att_s).sum(-1)              # fold(W, att): alpha's inputs are x @ v   (bench_mag.make_params)
        v_d = (w.reshape(F, H, C) * att_d).sum(-1)
        a_s, a_d = cu(x) @ cu(v_s), cu(x_dst) @ cu(v_d)
        agg = nn.gat_aggregate_heads(cu(rp), cu(col), cu(x), a_s, a_d, H, dst_rows=cu(dst_rows))
        got = nn.bias_act_rows(nn.gat_transform_heads(agg, cu(w), H), cu(bias), True)
    else:
        hs, hd = cu(x) @ cu(w), cu(x_dst) @ cu(w)
        a_s = (hs.view(-1, H, C) * cu(att_s)).sum(-1)
        a_d = (hd.view(-1, H, C) * cu(att_d)).sum(-1)
        full = torch.zeros((x_dst.shape[0], H * C), device="cuda")
        nn.gat_forward_rows(cu(rp), cu(col), hs, a_s.contiguous(), a_d.contiguous(), H, full, cu(dst_rows))
        got = torch.relu(full[cu(dst_rows)] + cu(bias))
    got = got.cpu().numpy()[z["rows"]].astype(np.float64)
    ref, scale = np.maximum(z["pre_activation"], 0.0), z["scale"]
    # the lin GEMMs of the transform-first route run in the library's fp32 (K = 128): its error bound is the same dot-product one
    assert np.all(np.abs(got - ref) <= 1e-5 * scale + 1e-7), np.abs(got - ref).max()
    big = np.abs(ref) >= 0.1 * scale
    assert big.sum() > 1000 and (np.abs(got - ref)[big] / np.abs(ref)[big]).max() <= 1e-5


def test_call_group_hop_rows_matches_the_index_formulas(hiplib):
    """wgamd_call_group_hop_rows against the torch index arithmetic it replaces (full and compact numbering)."""
    import torch
    from wholegraph_amd import _lib as L
    g = torch.Generator(device="cuda").manual_seed(4)
    G = 7
    cnt = torch.randint(0, 40, (G,), generator=g, device="cuda")                       # frontier entries per batch
    f_seg = torch.zeros(G + 1, dtype=torch.int32, device="cuda"); f_seg[1:] = torch.cumsum(cnt, 0)
    n_f = int(f_seg[-1])
    f_batch = torch.repeat_interleave(torch.arange(G, device="cuda"), cnt).int()
    f_local0 = torch.randint(0, 50, (G,), generator=g, device="cuda").int()
    deg = torch.randint(0, 30, (n_f,), generator=g, device="cuda")
    off = torch.zeros(n_f + 1, dtype=torch.int32, device="cuda"); off[1:] = torch.cumsum(deg, 0)
    n_e = int(off[-1])
    row_l = torch.randint(0, 500, (n_e,), generator=g, device="cuda").int()
    seg_d = (torch.arange(G + 1, device="cuda") * 1000).int(); seg_s = (torch.arange(G + 1, device="cuda") * 3000).int()
    cseg_d = torch.arange(G + 1, device="cuda") * 300; cseg_s = torch.arange(G + 1, device="cuda") * 700
    dst_full, dst_c = torch.empty(n_f, dtype=torch.int64, device="cuda"), torch.empty(n_f, dtype=torch.int64, device="cuda")
    col_full, col_c = torch.empty(n_e, dtype=torch.int32, device="cuda"), torch.empty(n_e, dtype=torch.int32, device="cuda")
    rc = hiplib.wgamd_call_group_hop_rows(off.data_ptr(), f_batch.data_ptr(), f_seg.data_ptr(), f_local0.data_ptr(), row_l.data_ptr(),
                                          n_f, seg_d.data_ptr(), cseg_d.data_ptr(), seg_s.data_ptr(), cseg_s.data_ptr(),
                                          dst_full.data_ptr(), dst_c.data_ptr(), col_full.data_ptr(), col_c.data_ptr(), None)
    assert rc == L.WHOLEMEMORY_SUCCESS
    torch.cuda.synchronize()
    fb = f_batch.long()
    local = f_local0.long()[fb] + (torch.arange(n_f, device="cuda") - f_seg.long()[fb])
    eb = torch.repeat_interleave(fb, deg)
    assert torch.equal(dst_full, seg_d.long()[fb] + local) and torch.equal(dst_c, cseg_d[fb] + local)
    assert torch.equal(col_full.long(), row_l.long() + seg_s.long()[eb]) and torch.equal(col_c.long(), row_l.long() + cseg_s[eb])
    # the (batch, chunk) block form the loaders call: same numbers, with and without the compact numbering, many entries per batch
    for G2, hi in ((G, 40), (3, 4000)):
        cnt2 = torch.randint(0, hi, (G2,), generator=g, device="cuda"); cnt2[1] = 0
        fs = torch.zeros(G2 + 1, dtype=torch.int32, device="cuda"); fs[1:] = torch.cumsum(cnt2, 0)
        nf2 = int(fs[-1])
        fb2 = torch.repeat_interleave(torch.arange(G2, device="cuda"), cnt2)
        deg2 = torch.randint(0, 30, (nf2,), generator=g, device="cuda")
        off2 = torch.zeros(nf2 + 1, dtype=torch.int32, device="cuda"); off2[1:] = torch.cumsum(deg2, 0)
        ne2 = int(off2[-1])
        rl2 = torch.randint(0, 500, (ne2,), generator=g, device="cuda").int()
        for compact in (True, False):
            d_f, d_c = torch.empty(nf2, dtype=torch.int64, device="cuda"), torch.empty(nf2, dtype=torch.int64, device="cuda")
            c_f, c_c = torch.empty(ne2, dtype=torch.int32, device="cuda"), torch.empty(ne2, dtype=torch.int32, device="cuda")
            rc = hiplib.wgamd_call_group_hop_rows_batched(
                off2.data_ptr(), fs.data_ptr(), f_local0.data_ptr(), rl2.data_ptr(), nf2, G2, seg_d.data_ptr(),
                cseg_d.data_ptr() if compact else None, seg_s.data_ptr(), cseg_s.data_ptr() if compact else None, d_f.data_ptr(),
                d_c.data_ptr() if compact else None, c_f.data_ptr(), c_c.data_ptr() if compact else None, None)
            assert rc == L.WHOLEMEMORY_SUCCESS
            torch.cuda.synchronize()
            local2 = f_local0.long()[fb2] + (torch.arange(nf2, device="cuda") - fs.long()[fb2])
            eb2 = torch.repeat_interleave(fb2, deg2)
            assert torch.equal(d_f, seg_d.long()[fb2] + local2) and torch.equal(c_f.long(), rl2.long() + seg_s.long()[eb2])
            if compact:
                assert torch.equal(d_c, cseg_d[fb2] + local2) and torch.equal(c_c.long(), rl2.long() + cseg_s[eb2])


def test_bias_act_rows(hiplib):
    import torch
    from wholegraph_amd import nn
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn((1000, 256), generator=g, device="cuda")
    b = torch.randn(256, generator=g, device="cuda")
    assert torch.equal(nn.bias_act_rows(x, b, True), torch.relu(x + b)) and torch.equal(nn.bias_act_rows(x, None, False), x)
    rows = torch.randperm(5000, generator=g, device="cuda")[:1000]
    out = torch.full((5000, 256), 7.0, device="cuda")
    nn.bias_act_rows(x, b, True, rows, out)
    want = torch.full((5000, 256), 7.0, device="cuda")
    want[rows] = torch.relu(x + b)
    assert torch.equal(out, want)


@pytest.mark.parametrize("F,T,n,dtype", [(128, 20, 5000, "int64"), (128, 8, 33, "int32"), (64, 32, 1000, "int64"), (256, 17, 777, "int64"),
                                         (32, 1, 100, "int32"), (128, 16, 16, "int64"), (128, 12, 0, "int64")])
def test_gather_with_terms_matches_gather_and_fp64_product(oracle_mod, hiplib, F, T, n, dtype):
    """wgamd_gather_terms_f32: out_x is the plain row gather bit for bit (the oracle's gather), out_terms = x @ v within fp32
    round-off of the fp64 product (1e-5 x scale); negative ids skip their row and give zero terms."""
    import numpy as np
    import torch
    from wholegraph_amd import nn
    rng = np.random.default_rng(F * 100 + T)
    V_rows = 4000
    table = rng.standard_normal((V_rows, F)).astype(np.float32)
    v = (rng.standard_normal((F, T)) * 0.3).astype(np.float32)
    ids = rng.integers(0, V_rows, n).astype(dtype)
    if n > 20:
        ids[::9] = -1
    assert nn.gather_terms_supported(F, T) and not nn.gather_terms_supported(100, 8) and not nn.gather_terms_supported(128, 33)
    out = torch.full((n, F), 7.0, dtype=torch.float32, device="cuda")
    x, terms = nn.gather_with_terms(torch.from_numpy(table).cuda(), torch.from_numpy(ids).cuda(), torch.from_numpy(v).cuda(), out=out)
    want_x = oracle_mod.gather_rows(table, np.where(ids < 0, 0, ids)) if n else np.zeros((0, F), np.float32)
    want_x[ids < 0] = 7.0
    assert np.array_equal(x.cpu().numpy(), want_x)
    live = want_x.astype(np.float64) * (ids >= 0)[:, None]
    ref = live @ v.astype(np.float64)
    scale = np.abs(live) @ np.abs(v.astype(np.float64))
    assert terms.shape == (n, T)
    assert np.all(np.abs(terms.cpu().numpy() - ref) <= 1e-5 * scale + 1e-7)
    assert np.all(terms.cpu().numpy()[ids < 0] == 0)
    if T % 4 == 0:   # the slab layout [T / 4][n][4]: same numbers, one contiguous [n, 4] block per relation end
        _, slabs = nn.gather_with_terms(torch.from_numpy(table).cuda(), torch.from_numpy(ids).cuda(), torch.from_numpy(v).cuda(), heads=4)
        assert slabs.shape == (T // 4, n, 4)
        assert torch.equal(slabs.permute(1, 0, 2).reshape(n, T), terms)


@pytest.mark.parametrize("G,cap_extra", [(1, 0), (7, 13), (64, 0), (191, 1000), (1000, 5)])
def test_frontier_list_equals_the_framework_formulation(hiplib, G, cap_extra):
    """wgamd_frontier_list against the torch index arithmetic it replaces (HeteroPygWalk._frontier): ids, batch and f_seg are
    equal everywhere, padding slots included; batches that gained nothing and batches that gained everything."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(G)
    sizes = torch.randint(0, 50, (G,), generator=g, device="cuda", dtype=torch.int32)
    seg = torch.zeros(G + 1, dtype=torch.int32, device="cuda")
    seg[1:] = torch.cumsum(sizes, 0)
    begin = (torch.rand(G, generator=g, device="cuda") * (sizes + 1)).to(torch.int32).clamp_(max=sizes)
    begin[::5] = 0
    begin[1::7] = sizes[1::7]
    n_nodes = max(int(seg[-1]), 1)
    nodes = torch.randint(0, 10 ** 9, (n_nodes,), generator=g, device="cuda")
    total = int((sizes - begin).sum())
    cap = total + cap_extra
    ids = torch.full((max(cap, 1),), -7, dtype=torch.int64, device="cuda")
    batch = torch.full((max(cap, 1),), -7, dtype=torch.int32, device="cuda")
    f_seg = torch.empty(G + 1, dtype=torch.int32, device="cuda")
    rc = hiplib.wgamd_frontier_list(nodes.data_ptr(), n_nodes, seg.data_ptr(), begin.data_ptr(), G, cap, ids.data_ptr(),
                                    batch.data_ptr(), f_seg.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    cnt = (seg[1:] - seg[:-1] - begin).to(torch.int32)
    want_seg = torch.zeros(G + 1, dtype=torch.int32, device="cuda")
    want_seg[1:] = torch.cumsum(cnt, 0)
    p = torch.arange(cap, dtype=torch.int32, device="cuda")
    b = torch.searchsorted(want_seg[1:].contiguous(), p, right=True).clamp_(max=G - 1)
    src = seg[:-1][b].long() + begin[b].long() + (p - want_seg[:-1][b]).long()
    want_ids = nodes[src.clamp_(0, n_nodes - 1)]
    assert torch.equal(f_seg, want_seg)
    assert torch.equal(ids[:cap], want_ids) and torch.equal(batch[:cap], b.to(torch.int32))


@pytest.mark.parametrize("F,T,n", [(256, 24, 70001), (256, 8, 15), (128, 12, 4096), (64, 5, 1000), (256, 20, 0)])
def test_rows_terms_is_the_gathers_product_without_ids(hiplib, F, T, n):
    """nn.rows_terms (wgamd_gather_terms_f32 with ids = NULL, out_x = NULL): bit-identical to the terms the gather produces
    for ids = 0 .. n-1 (same kernel, same MFMA order), within 1e-5 x scale of the fp64 product; x is not written; a strided
    view of a wider matrix works."""
    import numpy as np
    import torch
    from wholegraph_amd import nn
    rng = np.random.default_rng(F + T + n)
    wide = torch.from_numpy(rng.standard_normal((max(n, 1), F + 8)).astype(np.float32)).cuda()[:n]   # (n = 0: an empty view)
    x = wide[:, 4:4 + F]                                   # row stride F + 8, 16-byte aligned start
    v = torch.from_numpy((rng.standard_normal((F, T)) * 0.3).astype(np.float32)).cuda()
    terms = nn.rows_terms(x, v)
    assert terms.shape == (n, T)
    _, via_gather = nn.gather_with_terms(x, torch.arange(n, device="cuda"), v)
    assert torch.equal(terms, via_gather)
    ref = x.double().cpu().numpy() @ v.double().cpu().numpy()
    scale = np.abs(x.double().cpu().numpy()) @ np.abs(v.double().cpu().numpy())
    assert np.all(np.abs(terms.cpu().numpy() - ref) <= 1e-5 * scale + 1e-7)
    if T % 4 == 0:
        slabs = nn.rows_terms(x, v, heads=4)
        assert slabs.shape == (T // 4, n, 4) and torch.equal(slabs.permute(1, 0, 2).reshape(n, T), terms)


def test_gather_with_terms_refuses_unsupported_widths(hiplib):
    import torch
    import wholegraph_amd._lib as L
    from wholegraph_amd import nn
    t = torch.zeros((10, 100), device="cuda")
    with pytest.raises(L.WholeMemoryError):
        nn.gather_with_terms(t, torch.zeros(4, dtype=torch.int64, device="cuda"), torch.zeros((100, 8), device="cuda"))
