"""GPU: ``NeighborLoader.call_groups()`` — an epoch handed out as block-diagonal call groups
(cugraph_pyg_amd/loader/call_group.py; the reference samples ``local_seeds_per_call`` seeds per library call and hands out one
``Data`` per mini-batch, python/cugraph-pyg/cugraph_pyg/loader/node_loader.py:16-178, sampler/sampler.py:51-165).

* every mini-batch inside a call group is bit for bit what ``for batch in loader`` yields (and what the oracle gives);
* the LAZY ``x`` (table + ``n_id``, never gathered) through ``nn.SAGEConv`` equals the gathered ``x`` bit for bit;
* the trimmed two- and three-layer forward over a call group equals, for every seed, the plain PyG formulation — each layer
  over ALL sampled edges of that seed's mini-batch, untrimmed — computed in float64 on the CPU (rtol 1e-5)."""
import numpy as np
import pytest

from graphgen import powerlaw_csr

pytestmark = pytest.mark.gpu


def _stores(V, deg, F, seed=3):
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    row_ptr, col = powerlaw_csr(V, deg, seed=seed, max_deg=600)
    dst = np.repeat(np.arange(V), np.diff(row_ptr))
    gs, fs = GraphStore(), FeatureStore()
    gs[("n", "e", "n"), "coo", False, (V, V)] = torch.stack([torch.from_numpy(col.astype(np.int64)), torch.from_numpy(dst)]).cuda()
    feat = torch.from_numpy(np.random.default_rng(seed).standard_normal((V, F)).astype(np.float32))
    fs["n", "x", None] = feat.cuda()
    fs["n", "y", None] = torch.arange(V, dtype=torch.int64).cuda()
    return gs, fs, feat


def _model(dims, dev, seed=5):
    import torch
    from wholegraph_amd import nn
    g = torch.Generator().manual_seed(seed)
    convs = []
    for a, b in zip(dims[:-1], dims[1:]):
        c = nn.SAGEConv(a, b)
        for p in c.parameters():
            p.data = (torch.rand(p.shape, generator=g) - 0.5) * 0.4
        convs.append(c.to(dev))
    return convs


def _forward(convs, group, x):
    import torch
    with torch.no_grad():
        h = x
        for j, c in enumerate(convs):
            h = c(h, group.layer_graph(j), act="relu" if j + 1 < len(convs) else None)
    return h


def _reference_seed_outputs(convs, data):
    """PyG's plain formulation on ONE mini-batch, float64 on the host: every layer over all sampled edges and all nodes,
    mean over a node's in-edges, ReLU between layers; the seeds' rows of the result."""
    import torch
    x = data.x.double().cpu()
    src, dst = data.edge_index[0].cpu(), data.edge_index[1].cpu()
    n = x.shape[0]
    deg = torch.zeros(n, dtype=torch.float64).index_add_(0, dst, torch.ones(dst.shape[0], dtype=torch.float64))
    h = x
    for j, c in enumerate(convs):
        agg = torch.zeros((n, h.shape[1]), dtype=torch.float64).index_add_(0, dst, h[src]) / deg.clamp(min=1).unsqueeze(1)
        h = agg @ c.lin_l.weight.double().cpu().t() + c.lin_l.bias.double().cpu() + h @ c.lin_r.weight.double().cpu().t()
        if j + 1 < len(convs):
            h = torch.relu(h)
    return h[:data.batch_size]


@pytest.mark.parametrize("fanout,dims", [([25, 10], [100, 256, 47]), ([10, 5, 3], [64, 128, 64, 16]), ([7], [32, 64])])
@pytest.mark.parametrize("per_call", [4, 1])
def test_call_groups_equal_the_per_batch_loader_and_the_dense_formulation(hiplib, fanout, dims, per_call):
    import torch
    from cugraph_pyg_amd.loader import NeighborLoader
    from wholegraph_amd.nn import LazyRows
    V, B = 6000, 96
    gs, fs, feat = _stores(V, 14, dims[0])
    seeds = torch.from_numpy(np.random.default_rng(1).permutation(V)[:B * 9 + 17])       # 9 full batches + a short one
    make = lambda: NeighborLoader((fs, gs), fanout, input_nodes=seeds, batch_size=B, local_seeds_per_call=B * per_call,  # noqa: E731
                                  random_state=77)
    per_batch = list(make())
    groups = list(make().call_groups())
    assert sum(g.n_batches for g in groups) == len(per_batch) == 10
    assert [g.n_batches for g in groups] == [per_call] * (9 // per_call) + ([9 % per_call] if 9 % per_call else []) + [1]
    convs = _model(dims, "cuda")
    at = 0
    for g in groups:
        datas = g.to_data_list()
        # ---- every mini-batch of the group == the loader's own Data ------------------------------------------------
        node_ptr, batch_ptr = g.node_ptr.tolist(), g.batch_ptr.tolist()
        for j, d in enumerate(datas):
            ref = per_batch[at + j]
            assert torch.equal(d.n_id, ref.n_id) and torch.equal(d.edge_index, ref.edge_index) and torch.equal(d.e_id, ref.e_id)
            assert torch.equal(d.x, ref.x) and torch.equal(d.y, ref.y) and torch.equal(d.input_id, ref.input_id)
            assert d.num_sampled_nodes.tolist() == ref.num_sampled_nodes.tolist()
            assert d.num_sampled_edges.tolist() == ref.num_sampled_edges.tolist()
            assert torch.equal(g.n_id[node_ptr[j]:node_ptr[j + 1]], ref.n_id)
        assert g.num_nodes == sum(d.n_id.numel() for d in datas) and g.num_edges == sum(d.edge_index.shape[1] for d in datas)
        assert g.num_sampled_edges == [sum(int(d.num_sampled_edges[k]) for d in datas) for k in range(len(fanout))]
        assert torch.equal(g.batch, torch.cat([d.batch for d in datas])) and torch.equal(g.input_id, torch.cat([d.input_id for d in datas]))
        # block-diagonal COO of the group == the batches' edge lists shifted by their node offsets (as multisets per hop)
        ei = g.edge_index
        assert ei.shape[1] == g.num_edges and int(ei.max()) < g.num_nodes
        want = torch.cat([d.edge_index + node_ptr[j] for j, d in enumerate(datas)], dim=1)
        key = lambda e: torch.sort(e[0] * g.num_nodes + e[1]).values    # noqa: E731
        assert torch.equal(key(ei), key(want))
        # ---- lazy x == gathered x, bit for bit, through the model ----------------------------------------------------
        x_lazy = g.x
        assert isinstance(x_lazy, LazyRows) and tuple(x_lazy.shape) == (g.num_nodes, dims[0])
        x_dense = g.node_attr("x", lazy=False)
        assert isinstance(x_dense, torch.Tensor) and torch.equal(x_dense, feat[g.n_id.cpu()].cuda())
        out_lazy, out_dense = _forward(convs, g, x_lazy), _forward(convs, g, x_dense)
        assert out_lazy.shape == (g.num_seeds, dims[-1]) and torch.equal(out_lazy, out_dense)
        assert torch.equal(x_lazy.materialize(), x_dense) and torch.equal(x_lazy[3:5], x_dense[3:5])
        assert torch.equal(torch.relu(x_lazy), torch.relu(x_dense))         # any torch function sees the gathered rows
        # ---- the trimmed group forward == the plain per-batch formulation in float64 ---------------------------------------
        for j, d in enumerate(datas):
            want = _reference_seed_outputs(convs, d)
            got = out_lazy[batch_ptr[j]:batch_ptr[j + 1]].double().cpu()
            scale = float(want.abs().max()) + 1e-30
            assert float((got - want).abs().max()) <= 1e-5 * scale, (j, float((got - want).abs().max()), scale)
        at += g.n_batches
    assert at == len(per_batch)


def test_call_groups_refuse_what_they_cannot_do(hiplib):
    import torch
    from cugraph_pyg_amd.loader import NeighborLoader
    gs, fs, _ = _stores(500, 6, 8)
    with pytest.raises(NotImplementedError):
        NeighborLoader((fs, gs), [-1, 3], input_nodes=torch.arange(64), batch_size=16).call_groups()
    with pytest.raises(NotImplementedError):
        NeighborLoader((fs, gs), [3, 3], input_nodes=torch.arange(64), batch_size=16, replace=True).call_groups()


def test_sageconv_layer_graph_falls_back_to_two_kernels_with_the_same_result(hiplib):
    """Shapes the one-kernel layer is not built for (F % 4 != 0), and training mode (autograd on): aggregation kernel +
    library GEMM — same call, same result up to fp32 reassociation."""
    import torch
    from cugraph_pyg_amd.loader import NeighborLoader
    gs, fs, feat = _stores(3000, 10, 30)        # F = 30: not a multiple of 4
    loader = NeighborLoader((fs, gs), [6, 4], input_nodes=torch.arange(3000)[:256], batch_size=64, local_seeds_per_call=128,
                            random_state=5)
    convs = _model([30, 50, 7], "cuda")
    for g in loader.call_groups():
        out = _forward(convs, g, g.x)
        datas = g.to_data_list()
        bp = g.batch_ptr.tolist()
        for j, d in enumerate(datas):
            want = _reference_seed_outputs(convs, d)
            got = out[bp[j]:bp[j + 1]].double().cpu()
            assert float((got - want).abs().max()) <= 2e-5 * (float(want.abs().max()) + 1e-30)
        with torch.enable_grad():               # training: the autograd-capable path, gradients reach the weights
            h = g.x
            for j, c in enumerate(convs):
                h = c(h, g.layer_graph(j), act="relu" if j == 0 else None)
            h.sum().backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for c in convs for p in c.parameters())
        for c in convs:
            c.zero_grad()


def test_per_batch_loop_takes_the_sorted_edge_list_and_the_one_kernel_layer(hiplib):
    """The classic loop — `for batch in loader: conv(batch.x, batch.edge_index)` — in inference: the loader's edge list is
    destination-major and says so (no radix sort to build the CSR), and the layer runs as one kernel; same numbers as the
    sort + aggregate + library-GEMM route (bit-equal CSR, outputs to fp32 reassociation) and as the float64 formulation."""
    import torch
    from cugraph_pyg_amd.loader import NeighborLoader
    from wholegraph_amd import nn
    gs, fs, feat = _stores(5000, 12, 100)
    loader = NeighborLoader((fs, gs), [10, 5], input_nodes=torch.arange(5000)[:640], batch_size=128, local_seeds_per_call=256,
                            random_state=9)
    convs = _model([100, 256, 47], "cuda")
    n = 0
    for batch in loader:
        ei = batch.edge_index
        assert getattr(ei, "_wgamd_dst_sorted", None) == ei._version and bool((ei[1][1:] >= ei[1][:-1]).all())
        fast = nn._to_csr(ei, batch.x.shape[0])
        slow = nn._to_csr(ei.clone(), batch.x.shape[0])              # a plain tensor: the radix-sort route (stable)
        assert torch.equal(fast[0], slow[0]) and torch.equal(fast[1], slow[1])
        if n == 0:      # an in-place edit of the edge list voids the loader's "destination-sorted" promise
            ej = ei.clone()
            ej._wgamd_dst_sorted = ej._version
            ej[:, :ej.shape[1] // 2] = ej[:, :ej.shape[1] // 2].flip(1)      # no longer sorted by destination
            assert ej._wgamd_dst_sorted != ej._version
            redo, ref = nn._to_csr(ej, batch.x.shape[0]), nn._to_csr(ej.clone(), batch.x.shape[0])
            assert torch.equal(redo[0], ref[0]) and torch.equal(redo[1], ref[1])
        with torch.no_grad():
            h = convs[0](batch.x, ei, act="relu")
            out = convs[1](h, ei)[:batch.batch_size]
        with torch.enable_grad():                                    # the autograd route: aggregation kernel + torch Linear
            h2 = torch.relu(convs[0](batch.x, ei))
            out2 = convs[1](h2, ei)[:batch.batch_size]
        want = _reference_seed_outputs(convs, batch)
        scale = float(want.abs().max())
        assert float((out.double().cpu() - want).abs().max()) <= 1e-5 * scale
        assert float((out2.detach().double().cpu() - want).abs().max()) <= 2e-5 * scale
        n += 1
    assert n == 5


def test_call_groups_over_host_pinned_features_equal_the_hbm_placement():
    """FeatureStore(location="cpu") — the reference's default placement: the call-group path (lazy x read through n_id inside
    the layer-1 kernel, and the gathered x) gives bit for bit what the HBM placement gives."""
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    from wholegraph_amd import nn
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    V, E, F = 20000, 300000, 100
    src = torch.randint(0, V, (E,), generator=g, device=dev)
    dst = torch.randint(0, V, (E,), generator=g, device=dev)
    feat = torch.randn(V, F)
    convs = [nn.SAGEConv(F, 64).to(dev), nn.SAGEConv(64, 16).to(dev)]
    for c in convs:
        for p in c.parameters():
            p.data = torch.randn(p.shape, generator=g, device=dev) * 0.1
            p.requires_grad_(False)
    outs = {}
    for loc in ("cuda", "cpu"):
        gs, fs = GraphStore(), FeatureStore(location=loc)
        gs[("n", "e", "n"), "coo", False, (V, V)] = torch.stack([src, dst])
        fs["n", "x", None] = feat
        assert fs["n", "x", None].get_local_tensor().device.type == loc
        loader = NeighborLoader((fs, gs), [10, 5], input_nodes=torch.arange(8192, device=dev), batch_size=1024, shuffle=False,
                                random_state=3)
        res = []
        with torch.no_grad():
            for grp in loader.call_groups():
                for lazy in (True, False):
                    h = grp.x if lazy else grp.node_attr("x", lazy=False)
                    for j, c in enumerate(convs):
                        h = c(h, grp.layer_graph(j), act="relu" if j == 0 else None)
                    res.append(h.clone())
        assert all(torch.equal(res[i], res[i + 1]) for i in range(0, len(res), 2)), "lazy x != gathered x (%s)" % loc
        outs[loc] = res
    assert len(outs["cpu"]) == len(outs["cuda"]) > 0
    assert all(torch.equal(a, b) for a, b in zip(outs["cuda"], outs["cpu"]))


def test_group_rows_pool_reuses_a_buffer_only_when_nothing_refers_to_it():
    """The materialised ``x = feat[n_id]`` of a call group comes from a grow-only pool (no hipMalloc per group): a buffer is
    handed out again only when neither the returned tensor, nor a view of it, nor anything autograd saved is alive."""
    import torch
    from cugraph_pyg_amd.loader import NeighborLoader
    from cugraph_pyg_amd.sampler import sampler as smp
    gs, fs, feat = _stores(5000, 12, 100)
    smp._group_rows.clear()
    loader = NeighborLoader((fs, gs), [10, 5], input_nodes=torch.arange(5000)[:1280], batch_size=128, local_seeds_per_call=256,
                            random_state=9)
    ptrs, kept = [], None
    for k, grp in enumerate(loader.call_groups()):
        x = grp.node_attr("x", lazy=False)
        assert torch.equal(x, feat.cuda()[grp.n_id])
        ptrs.append(x.untyped_storage().data_ptr())
        if k == 1:
            kept = x[5:7]             # a VIEW kept alive: its buffer must not come back
        del x
    assert len(ptrs) == 5 and ptrs[0] == ptrs[1], ptrs              # released -> reused
    assert ptrs[2] != ptrs[1] and ptrs[3] not in (ptrs[1],) and ptrs[4] not in (ptrs[1],), ptrs
    assert kept is not None and torch.equal(kept.cpu(), feat[list(loader.call_groups())[1].n_id.cpu()][5:7])


@pytest.mark.parametrize("hops", [2, 3])
def test_hetero_call_groups_equal_the_per_batch_loader(hiplib, hops):
    """``NeighborLoader.call_groups()`` on a heterogeneous graph: every mini-batch inside a ``HeteroCallGroup`` is the
    ``HeteroData`` ``for batch in loader`` yields (node lists, counts), and a stack of ``nn.HeteroConv{GATConv}`` layers over the
    group's TRIMMED per-layer relation hops (one-kernel relations, aggregate-first) gives the seeds the outputs of PyG's own
    formulation — every layer over all sampled edges of the mini-batch, transform-first, in float64 — for 2 and 3 hops, a ragged
    last mini-batch included."""
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import HeteroCallGroup, NeighborLoader
    from wholegraph_amd import nn
    torch.manual_seed(4 + hops)
    n = {"paper": 3000, "author": 1500, "venue": 40}
    rel = {("paper", "cites", "paper"): 20000, ("author", "writes", "paper"): 9000, ("paper", "rev_writes", "author"): 9000,
           ("venue", "publishes", "paper"): 3000, ("paper", "rev_publishes", "venue"): 3000}
    gs, fs = GraphStore(), FeatureStore()
    for (s, r, d), m in rel.items():
        gs[(s, r, d), "coo", False, (n[s], n[d])] = torch.stack([torch.randint(0, n[s], (m,)), torch.randint(0, n[d], (m,))])
    feat = {t: torch.randn(n[t], 128) for t in n}
    for t in n:
        fs[t, "x", None] = feat[t].cuda()
    B = 32
    seeds = torch.randperm(n["paper"])[:B * 5 + 7].cuda()
    fan = {et: [4, 3, 2][:hops] for et in rel}
    mk = lambda per_call: NeighborLoader((fs, gs), fan, input_nodes=("paper", seeds), batch_size=B, shuffle=False,      # noqa: E731
                                         random_state=3, local_seeds_per_call=per_call)
    batches = list(mk(B))
    layers = []
    for j in range(hops):
        layers.append(nn.HeteroConv({et: nn.GATConv(128 if j == 0 else 256, 64, heads=4, add_self_loops=False) for et in rel}).cuda())
    groups = list(mk(B * 2).call_groups())
    assert all(isinstance(g, HeteroCallGroup) for g in groups) and [g.n_batches for g in groups] == [2, 2, 1, 1]
    assert sum(g.num_edges for g in groups) == sum(int(b[et].edge_index.shape[1]) for b in batches for et in rel)
    b0 = 0
    for grp in groups:
        with torch.no_grad():
            h = grp.x_dict
            for j, layer in enumerate(layers):
                h = layer(h, grp.layer_graph(j), act="relu")
        out = h["paper"]
        assert out.shape == (grp.num_seeds, 256)
        ptr = {t: grp.node_ptr[t].tolist() for t in n}
        seed_ptr = grp.batch_ptr.tolist()
        for j in range(grp.n_batches):
            batch = batches[b0 + j]
            for t in n:
                assert torch.equal(grp.n_id[t][ptr[t][j]:ptr[t][j + 1]], batch[t].n_id), (b0 + j, t)
            # PyG's formulation on the mini-batch alone, float64: every layer over ALL its sampled edges
            x = {t: feat[t][batch[t].n_id.cpu()].double().cuda() for t in n}
            for layer in layers:
                y = {}
                for et in rel:
                    c = layer.conv(et)
                    ei = batch[et].edge_index
                    if ei.shape[1] == 0 and et[2] not in y:
                        y.setdefault(et[2], torch.zeros((x[et[2]].shape[0], 256), dtype=torch.float64, device="cuda"))
                        continue
                    w = c.lin.weight.double().t()
                    hs, hd = (x[et[0]] @ w).view(-1, 4, 64), (x[et[2]] @ w).view(-1, 4, 64)
                    a_s, a_d = (hs * c.att_src.double()).sum(-1), (hd * c.att_dst.double()).sum(-1)
                    e = torch.nn.functional.leaky_relu(a_s[ei[0]] + a_d[ei[1]], 0.2)
                    e = e - torch.zeros((x[et[2]].shape[0], 4), dtype=torch.float64, device="cuda").index_reduce_(
                        0, ei[1], e, "amax", include_self=False)[ei[1]]
                    p = e.exp()
                    den = torch.zeros((x[et[2]].shape[0], 4), dtype=torch.float64, device="cuda").index_add_(0, ei[1], p)
                    alpha = p / den[ei[1]]
                    msg = torch.zeros((x[et[2]].shape[0], 4, 64), dtype=torch.float64, device="cuda").index_add_(
                        0, ei[1], alpha.unsqueeze(-1) * hs[ei[0]])
                    res = msg.view(-1, 256) + (c.bias.double() if c.bias is not None else 0)
                    y[et[2]] = res if et[2] not in y else y[et[2]] + res
                x = {t: torch.relu(v) for t, v in y.items()}
            want = x["paper"][:batch["paper"].batch_size]
            got = out[seed_ptr[j]:seed_ptr[j + 1]].double()
            scale = want.abs().amax(dim=1, keepdim=True).clamp_(min=1e-30)
            assert bool(((got - want).abs() <= 1e-5 * scale + 1e-7).all()), (b0 + j, float(((got - want).abs() / scale).max()))
        b0 += grp.n_batches
    assert b0 == len(batches) == 6


def _gat_reference_seed_outputs(convs, data, self_loops):
    """PyG's GATConv on ONE mini-batch, float64 on the host, every layer over ALL sampled edges and all nodes of the batch
    (self loops as ``csr_add_self_loop`` makes them: one extra (i, i) edge per node); ReLU between layers; the seeds' rows."""
    import torch
    h = data.x.double().cpu()
    src, dst = data.edge_index[0].cpu(), data.edge_index[1].cpu()
    n = h.shape[0]
    if self_loops:
        loops = torch.arange(n)
        src, dst = torch.cat([loops, src]), torch.cat([loops, dst])
    for j, c in enumerate(convs):
        H, C = c.heads, c.out_channels
        hw = (h @ c.lin.weight.detach().double().cpu().t()).view(n, H, C)
        a_s = (hw * c.att_src.detach().double().cpu()).sum(-1)
        a_d = (hw * c.att_dst.detach().double().cpu()).sum(-1)
        e = torch.nn.functional.leaky_relu(a_s[src] + a_d[dst], c.negative_slope)
        mx = torch.full((n, H), -float("inf"), dtype=torch.float64).index_reduce_(0, dst, e, "amax", include_self=True)
        ex = torch.exp(e - mx[dst])
        den = torch.zeros((n, H), dtype=torch.float64).index_add_(0, dst, ex)
        alpha = ex / den[dst]
        out = torch.zeros((n, H, C), dtype=torch.float64).index_add_(0, dst, alpha.unsqueeze(-1) * hw[src])
        out = out.reshape(n, H * C) if c.concat else out.mean(1)
        if c.bias is not None:
            out = out + c.bias.detach().double().cpu()
        h = torch.relu(out) if j + 1 < len(convs) else out
    return h[:data.batch_size]


def test_homogeneous_gatconv_mean_over_heads_and_no_bias(hiplib):
    """concat=False (mean over the heads) and bias=False through the call-group route, forward against the float64 formulation."""
    import torch
    from cugraph_pyg_amd.loader import NeighborLoader
    from wholegraph_amd import nn
    gs, fs, feat = _stores(4000, 10, 32, seed=13)
    torch.manual_seed(8)
    convs = [nn.GATConv(32, 8, heads=8, bias=False).cuda(), nn.GATConv(64, 12, heads=2, concat=False).cuda()]
    seeds = torch.randperm(4000, generator=torch.Generator().manual_seed(3))[:3 * 64].cuda()
    loader = NeighborLoader((fs, gs), [5, 5], input_nodes=seeds, batch_size=64, shuffle=False, random_state=9, local_seeds_per_call=3 * 64)
    g = next(iter(loader.call_groups()))
    with torch.no_grad():
        h = g.x
        for j, c in enumerate(convs):
            h = c(h, g.layer_graph(j), act="relu" if j == 0 else None)
    want = torch.cat([_gat_reference_seed_outputs(convs, d, True) for d in g.to_data_list()])
    assert h.shape == want.shape == (3 * 64, 12)
    assert float((h.double().cpu() - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("self_loops,table_rows", [(True, 3000), (False, 3000), (True, 60000)])
def test_homogeneous_gatconv_over_call_groups(hiplib, self_loops, table_rows):
    """nn.GATConv over a homogeneous call group's trimmed layer graphs (aggregate-first, x lazy; with a short table the
    attention terms are those of the table's rows, with a long one those of the listed rows) equals, for every seed, PyG's
    formulation over ALL sampled edges of its mini-batch in float64 — and trains: parameter gradients of the call-group route
    against float64 autograd of the same reference."""
    import torch
    from cugraph_pyg_amd.loader import NeighborLoader
    from wholegraph_amd import nn
    from wholegraph_amd.nn import LazyRows
    V, F0 = table_rows, 64
    gs, fs, feat = _stores(V, 12, F0, seed=11)
    torch.manual_seed(4)
    convs = [nn.GATConv(F0, 16, heads=4, add_self_loops=self_loops).cuda(), nn.GATConv(64, 8, heads=2, add_self_loops=self_loops).cuda()]
    for c in convs:
        for p in c.parameters():
            p.data = (torch.rand(p.shape, device="cuda") - 0.5) * 0.6
    seeds = torch.randperm(V, generator=torch.Generator().manual_seed(1))[:4 * 96 + 17].cuda()
    loader = NeighborLoader((fs, gs), [6, 4], input_nodes=seeds, batch_size=96, shuffle=False, random_state=5, local_seeds_per_call=4 * 96)
    groups = list(loader.call_groups())
    assert len(groups) == 2
    for g in groups:
        x = g.x
        assert isinstance(x, LazyRows)
        h = x
        for j, c in enumerate(convs):
            h = c(h, g.layer_graph(j), act="relu" if j == 0 else None)
        got = h.detach().double().cpu()
        bp = g.batch_ptr.tolist()
        want = torch.cat([_gat_reference_seed_outputs(convs, d, self_loops) for d in g.to_data_list()])
        assert got.shape == want.shape == (bp[-1], 16)
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 1e-5 * scale, float((got - want).abs().max()) / scale
    # gradients (first group): the call-group route under autograd against float64 autograd of the reference
    g = groups[0]
    params = [p for c in convs for p in c.parameters()]
    gout = torch.randn((g.batch_ptr.tolist()[-1], 16), generator=torch.Generator().manual_seed(2)).cuda()
    h = g.x
    for j, c in enumerate(convs):
        h = c(h, g.layer_graph(j), act=None)
    h.backward(gout)
    got_grads = [p.grad.detach().double().cpu().clone() for p in params]

    class _Dbl:          # the same modules' parameters as float64 leaves on the host
        def __init__(self, c):
            self.heads, self.out_channels, self.concat, self.negative_slope = c.heads, c.out_channels, c.concat, c.negative_slope
            self.lin = type("L", (), {})()
            self.lin.weight = c.lin.weight.detach().double().cpu().requires_grad_(True)
            self.att_src = c.att_src.detach().double().cpu().requires_grad_(True)
            self.att_dst = c.att_dst.detach().double().cpu().requires_grad_(True)
            self.bias = None if c.bias is None else c.bias.detach().double().cpu().requires_grad_(True)

    def ref_forward(cs, data):
        hh = data.x.double().cpu()
        src, dst = data.edge_index[0].cpu(), data.edge_index[1].cpu()
        n = hh.shape[0]
        if self_loops:
            loops = torch.arange(n)
            src, dst = torch.cat([loops, src]), torch.cat([loops, dst])
        for c in cs:
            H, C = c.heads, c.out_channels
            hw = (hh @ c.lin.weight.t()).view(n, H, C)
            e = torch.nn.functional.leaky_relu((hw * c.att_src).sum(-1)[src] + (hw * c.att_dst).sum(-1)[dst], c.negative_slope)
            mx = torch.full((n, H), -float("inf"), dtype=torch.float64).index_reduce_(0, dst, e.detach(), "amax", include_self=True)
            ex = torch.exp(e - mx[dst])
            den = torch.zeros((n, H), dtype=torch.float64).index_add(0, dst, ex)
            out = torch.zeros((n, H, C), dtype=torch.float64).index_add(0, dst, (ex / den[dst]).unsqueeze(-1) * hw[src])
            hh = out.reshape(n, H * C) + c.bias
        return hh[:data.batch_size]

    dbl = [_Dbl(c) for c in convs]
    ref = torch.cat([ref_forward(dbl, d) for d in g.to_data_list()])
    ref.backward(gout.double().cpu())
    want_grads = [q.grad for c in dbl for q in (c.lin.weight, c.att_src, c.att_dst, c.bias)]
    names = [n for c in convs for n, _ in c.named_parameters()]
    order = {"lin.weight": 0, "att_src": 1, "att_dst": 2, "bias": 3}
    got_by = {}
    for (ci, c) in enumerate(convs):
        for n, p in c.named_parameters():
            got_by[(ci, order[n])] = p.grad.detach().double().cpu()
    for ci in range(2):
        for k in range(4):
            a, b = got_by[(ci, k)], want_grads[ci * 4 + k]
            scale = float(b.abs().max())
            # (north_star's 1e-5 of the gradient's own maximum; the aggregation's gradients are held entry by entry to 1e-5 x the
            #  magnitude sum of their terms in tests/test_gpu_mag_pipeline.py)
            assert float((a - b.reshape(a.shape)).abs().max()) <= 1e-5 * scale + 1e-9, (ci, k, float((a - b.reshape(a.shape)).abs().max()), scale)


@pytest.mark.parametrize("F,H,C", [(64, 4, 16), (100, 2, 8), (32, 8, 4), (256, 4, 64)])
def test_hetero_conv_routes_agree_for_other_shapes(hiplib, F, H, C):
    """nn.HeteroConv over a heterogeneous call group for shapes the one-kernel relation does not take: the inference route with
    x lazy == with x gathered (bit for bit), == the relation-by-relation route on GATConv and the aggregate-first training route
    to fp32 accuracy."""
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    from wholegraph_amd import nn
    torch.manual_seed(F + H)
    n = {"paper": 900, "author": 700}
    rel = {("paper", "cites", "paper"): 9000, ("author", "writes", "paper"): 6000, ("paper", "rev_writes", "author"): 6000}
    gs, fs = GraphStore(), FeatureStore()
    for (s, r, d), m in rel.items():
        gs[(s, r, d), "coo", False, (n[s], n[d])] = torch.stack([torch.randint(0, n[s], (m,)), torch.randint(0, n[d], (m,))])
    for t in n:
        fs[t, "x", None] = torch.randn(n[t], F).cuda()
    B = 48
    seeds = torch.randperm(n["paper"])[:B * 6].cuda()
    loader = NeighborLoader((fs, gs), {et: [6, 4] for et in rel}, input_nodes=("paper", seeds), batch_size=B, shuffle=False,
                            random_state=3, local_seeds_per_call=B * 6)
    grp = next(iter(loader.call_groups()))
    layers = [nn.HeteroConv({et: nn.GATConv(F if j == 0 else H * C, C, heads=H, add_self_loops=False) for et in rel}).cuda()
              for j in range(2)]

    def run(**flags):
        for layer in layers:
            for k, v in flags.items():
                setattr(layer, k, v)
        h = grp.x_dict
        for j, layer in enumerate(layers):
            h = layer(h, grp.layer_graph(j), act=None)
        return h["paper"]
    with torch.no_grad():
        lazy = run(fetch_in_layer=True)
        gathered = run(fetch_in_layer=False)
    assert lazy.shape == (B * 6, H * C) and torch.equal(lazy, gathered)
    trained = run(fetch_in_layer=True, train_aggregate_first=True)
    relations = run(train_aggregate_first=False)
    assert trained.requires_grad and relations.requires_grad
    scale = float(lazy.abs().max())
    assert float((trained.detach() - lazy).abs().max()) <= 2e-5 * scale and float((relations.detach() - lazy).abs().max()) <= 2e-5 * scale
