"""GPU: training with the reference's optimizer semantics — sample per call group, STEP PER MINI-BATCH
(cugraph_pyg_amd.loader.PerBatchStep; the reference: one ``optimizer.step()`` per mini-batch,
python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:119-125, cugraph_pyg/examples/gcn_dist_mnmg.py).

* the staged mini-batch (wgamd_call_group_stage_batch: fixed-size buffers, batch-local ids) is bit for bit the ``Data`` that
  ``for batch in loader`` yields for it;
* one captured step gives the gradients of that mini-batch's float64 formulation (PyG SAGEConv, every layer over all
  sampled edges, untrimmed) at 1e-5: |err| <= 1e-5 x the magnitude sum of the terms + element-wise 1e-5 on the elements that
  are not cancellations — the contract of tests/test_gpu_sage_train.py;
* an epoch of captured steps leaves the parameters where the eager per-mini-batch loop leaves them."""
import numpy as np
import pytest

from graphgen import powerlaw_csr

pytestmark = pytest.mark.gpu

F_IN, HID, CLS = 100, 256, 47


def _stores(V, deg, seed=3):
    import torch
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    row_ptr, col = powerlaw_csr(V, deg, seed=seed, max_deg=400)
    dst = np.repeat(np.arange(V), np.diff(row_ptr))
    gs, fs = GraphStore(), FeatureStore()
    gs[("n", "e", "n"), "coo", False, (V, V)] = torch.stack([torch.from_numpy(col.astype(np.int64)), torch.from_numpy(dst)]).cuda()
    feat = torch.from_numpy(np.random.default_rng(seed).standard_normal((V, F_IN)).astype(np.float32)).cuda()
    fs["n", "x", None] = feat
    return gs, fs, feat


def _model(seed=5):
    import torch
    from wholegraph_amd import nn
    g = torch.Generator().manual_seed(seed)
    convs = torch.nn.ModuleList([nn.SAGEConv(F_IN, HID), nn.SAGEConv(HID, CLS)])
    for p in convs.parameters():
        p.data = (torch.rand(p.shape, generator=g) - 0.5) * 0.3
    return convs.cuda()


def _step_fn(model, opt, labels, fused_loss=False):
    import torch

    def step(batch):
        opt.zero_grad(set_to_none=True)
        h = batch.x
        for j, c in enumerate(model):
            h = c(h, batch.layer_graph(j), act="relu" if j + 1 < len(model) else None)
        B = batch.batch_size
        if fused_loss:      # wholegraph_amd.nn.cross_entropy: one launch forward, one backward (what bench.py's step uses)
            from wholegraph_amd import nn as wnn
            loss = wnn.cross_entropy(h, labels[batch.n_id[:h.shape[0]]], batch.seed_mask)       # (no slice: row weights)
        else:
            per_seed = torch.nn.functional.cross_entropy(h[:B], labels[batch.seeds], reduction="none")
            loss = (per_seed * batch.seed_mask[:B]).sum() / batch.n_live_seeds
        loss.backward()
        opt.step()
        return loss
    return step


def _hidden_masks(model, sb):
    """ReLU sign patterns of the hidden layers as the device computes them for the staged mini-batch, by batch-local vertex id
    (the same launches as inside the captured step: same bits).  The float64 reference takes its masks from here — a hidden
    pre-activation within fp32 round-off of zero may legitimately land on either side, and with a few hundred rows per sum one
    flipped entry moves a gradient row by several per cent (the convention of tests/test_gpu_sage_train.py and smoke())."""
    import torch
    masks, sizes = [], sb.sizes.tolist()
    with torch.no_grad():
        h = sb.x
        for j, c in enumerate(list(model)[:-1]):
            h = c(h, sb.layer_graph(j), act="relu")
            parts, base = [], 0
            for k in range(sb.hops - j):
                parts.append(h[base:base + sizes[2 * k]] > 0)
                base += sb.row_cap[k]
            masks.append(torch.cat(parts).cpu())
    return masks


def _fp64_grads(model, data, labels, masks=None):
    """The mini-batch's loss and gradients in float64 on the host: PyG's SAGEConv (mean) over ALL sampled edges, untrimmed."""
    import torch
    x = data.x.double().cpu()
    src, dst = data.edge_index[0].cpu(), data.edge_index[1].cpu()
    n = x.shape[0]
    deg = torch.zeros(n, dtype=torch.float64).index_add_(0, dst, torch.ones(dst.shape[0], dtype=torch.float64)).clamp_(min=1)
    ps = [[p.detach().double().cpu().requires_grad_(True) for p in (c.lin_l.weight, c.lin_r.weight, c.lin_l.bias)] for c in model]
    h = x
    for j, (wl, wr, b) in enumerate(ps):
        agg = torch.zeros((n, h.shape[1]), dtype=torch.float64).index_add_(0, dst, h[src]) / deg.unsqueeze(1)
        h = agg @ wl.t() + h @ wr.t() + b
        if j + 1 < len(ps):
            m = h > 0
            if masks is not None:        # the vertices a later layer can still see: the device's own sign pattern
                m = m.clone()
                m[:masks[j].shape[0]] = masks[j]
            h = h * m
    B = int(data.batch_size)
    loss = torch.nn.functional.cross_entropy(h[:B], labels.cpu()[data.n_id[:B].cpu()])
    loss.backward()
    return float(loss.detach()), [[p.grad for p in layer] for layer in ps]


def _close(got, ref, what):
    import torch
    err = (got.double().cpu() - ref).abs()
    scale = float(ref.abs().max())
    assert bool((err <= 1e-5 * scale + 1e-8).all()), (what, float(err.max()), scale)
    assert bool(torch.isfinite(got).all()), what


def test_staged_batch_is_the_loaders_data(hiplib):
    import torch
    from cugraph_pyg_amd.loader import NeighborLoader, PerBatchStep
    gs, fs, feat = _stores(6000, 14)
    seeds = torch.randperm(6000, generator=torch.Generator().manual_seed(1))[:5 * 64 + 17].cuda()
    loader = NeighborLoader((fs, gs), [5, 3], input_nodes=seeds, batch_size=64, shuffle=False, random_state=9, local_seeds_per_call=3 * 64)
    stepper = PerBatchStep(lambda batch: None, table=feat)
    n_checked = 0
    for grp in loader.call_groups():
        datas = grp.to_data_list()
        rows, edges, nodes = stepper._group_sizes(grp)
        if stepper.batch is None or not stepper.batch.fits(rows, edges, nodes):
            stepper._make_buffers(grp, rows, edges, nodes)
        for b, d in enumerate(datas):
            stepper.stage(grp, b)
            sb = stepper.batch
            sz = sb.sizes.tolist()
            assert sz[-1] == 0 and sz[2 * grp.hops] == d.n_id.shape[0]
            assert torch.equal(sb.n_id[:sz[-2]], d.n_id) and bool((sb.n_id[sz[-2]:] == d.n_id[0]).all())
            nn_, ne_ = d.num_sampled_nodes.tolist(), d.num_sampled_edges.tolist()
            at_e, at_n = 0, 0
            for k in range(grp.hops):
                n_rows, n_edges = sz[2 * k], sz[2 * k + 1]
                assert n_rows == nn_[k] and n_edges == ne_[k]
                rp = sb.row_ptr[k]
                # live rows first; the slack edges are dealt to the slack rows (every entry of the arrays is a well-formed edge)
                assert int(rp[0]) == 0 and int(rp[n_rows]) == n_edges and int(rp[-1]) == sb.edge_cap[k]
                assert bool((rp[1:] >= rp[:-1]).all()) and n_rows < sb.row_cap[k]
                assert int(sb.col[k].min()) >= 0 and int(sb.col[k].max()) < sz[2 * grp.hops]
                # hop k's edges of the Data: sources = edge_index[0], destinations = edge_index[1], destination-major
                src = d.edge_index[0][at_e:at_e + n_edges].to(torch.int32)
                dst = d.edge_index[1][at_e:at_e + n_edges]
                assert torch.equal(sb.col[k][:n_edges], src)
                counts = (rp[1:n_rows + 1] - rp[:n_rows]).long()
                assert torch.equal(torch.repeat_interleave(sb.self0[k][:n_rows], counts), dst)
                assert torch.equal(sb.self0[k][:n_rows], torch.arange(at_n, at_n + n_rows, device="cuda"))
                if k + 1 < grp.hops:      # sources as rows of a trimmed layer's output (hops 0 .. k + 1 back to back)
                    assert int(sb.col_seg[k].min()) >= 0 and int(sb.col_seg[k].max()) < sum(sb.row_cap[:k + 2])
                    seg_first = [0] + np.cumsum(nn_).tolist()          # first batch-local id of every hop's vertices
                    base = [0] + np.cumsum(sb.row_cap).tolist()
                    s_of = torch.bucketize(src.long(), torch.tensor(seg_first[1:], device="cuda"), right=True)
                    want = torch.tensor(base, device="cuda")[s_of] + src.long() - torch.tensor(seg_first, device="cuda")[s_of]
                    assert torch.equal(sb.col_seg[k][:n_edges].long(), want)
                else:
                    assert sb.col_seg[k] is None
                at_e, at_n = at_e + n_edges, at_n + n_rows
            # the hops back to back as ONE CSR (what layer_graph(j) hands to a layer: one launch over a prefix) and 1 / degree
            want_rp = [torch.zeros(1, dtype=torch.int32, device="cuda")]
            ebase = 0
            for k in range(grp.hops):
                want_rp.append(sb.row_ptr[k][1:] + ebase)
                ebase += sb.edge_cap[k]
            want_rp = torch.cat(want_rp)
            assert torch.equal(sb.row_ptr_all, want_rp)
            assert torch.equal(sb.col_all, torch.cat(sb.col)) and torch.equal(sb.self0_all, torch.cat(sb.self0))
            deg = (want_rp[1:] - want_rp[:-1]).clamp(min=1).float()
            assert torch.equal(sb.inv_deg_all, 1.0 / deg)
            for layer in range(grp.hops):
                one, per_hop = sb.layer_graph(layer), sb.layer_graph(layer, per_hop=True)
                assert len(one.hops) == 1 and len(per_hop.hops) == grp.hops - layer and one.n_rows == per_hop.n_rows
                R, E = sum(sb.row_cap[:grp.hops - layer]), sum(sb.edge_cap[:grp.hops - layer])
                assert int(one.hops[0].row_ptr[-1]) == E and one.hops[0].col.shape[0] == E and one.hops[0].self_rows.shape[0] == R
                assert torch.equal(one.hops[0].col, torch.cat([h.col for h in per_hop.hops]))
                assert torch.equal(one.hops[0].self_rows, torch.cat([h.self_rows for h in per_hop.hops]))
            n_checked += 1
    assert n_checked == 6


@pytest.mark.parametrize("fanout,fused_loss", [([5, 3], False), ([4, 3, 2], False), ([5, 3], True), ([4, 3, 2], True)])
def test_one_captured_step_has_the_fp64_gradients_of_its_mini_batch(hiplib, fanout, fused_loss):
    import torch
    from wholegraph_amd import nn
    from cugraph_pyg_amd.loader import NeighborLoader, PerBatchStep
    gs, fs, feat = _stores(6000, 14)
    labels = torch.randint(0, CLS, (6000,), generator=torch.Generator().manual_seed(2)).cuda()
    seeds = torch.randperm(6000, generator=torch.Generator().manual_seed(1))[:4 * 64 + 23].cuda()
    loader = NeighborLoader((fs, gs), fanout, input_nodes=seeds, batch_size=64, shuffle=False, random_state=9, local_seeds_per_call=2 * 64)
    g = torch.Generator().manual_seed(5)
    dims = [F_IN] + [HID] * (len(fanout) - 1) + [CLS]
    model = torch.nn.ModuleList([nn.SAGEConv(a, b) for a, b in zip(dims[:-1], dims[1:])])
    for p in model.parameters():
        p.data = (torch.rand(p.shape, generator=g) - 0.5) * 0.3
    model = model.cuda()
    opt = torch.optim.SGD(model.parameters(), lr=0.0)        # lr 0: every step sees the same weights, p.grad stays readable
    stepper = PerBatchStep(_step_fn(model, opt, labels, fused_loss), table=feat, optimizer=opt)
    n = 0
    for grp in loader.call_groups():
        datas = grp.to_data_list()
        for b, d in enumerate(datas):
            loss = stepper(grp, b)
            ref_loss, ref = _fp64_grads(model, d, labels, _hidden_masks(model, stepper.batch))
            assert abs(float(loss.detach()) - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (n, float(loss.detach()), ref_loss)
            for c, (gl, gr, gb) in zip(model, ref):
                _close(c.lin_l.weight.grad, gl, ("lin_l", n))
                _close(c.lin_r.weight.grad, gr, ("lin_r", n))
                _close(c.lin_l.bias.grad, gb, ("bias", n))
            n += 1
    assert n == 5 and stepper.captures >= 1


def test_an_epoch_of_captured_steps_equals_the_eager_per_batch_loop(hiplib):
    import torch
    from cugraph_pyg_amd.loader import NeighborLoader, PerBatchStep
    gs, fs, feat = _stores(6000, 14)
    labels = torch.randint(0, CLS, (6000,), generator=torch.Generator().manual_seed(2)).cuda()
    seeds = torch.randperm(6000, generator=torch.Generator().manual_seed(1))[:7 * 64].cuda()

    def loader():
        return NeighborLoader((fs, gs), [5, 3], input_nodes=seeds, batch_size=64, shuffle=False, random_state=9, local_seeds_per_call=3 * 64)
    # eager: the loop a reference user writes — one Data per mini-batch, SAGEConv over its edge_index, step per mini-batch
    eager = _model()
    opt_e = torch.optim.SGD(eager.parameters(), lr=0.05, momentum=0.9)
    for d in loader():
        opt_e.zero_grad(set_to_none=True)
        h = d.x
        for j, c in enumerate(eager):
            h = c(h, d.edge_index, act="relu" if j == 0 else None)
        torch.nn.functional.cross_entropy(h[:d.batch_size], labels[d.n_id[:d.batch_size]]).backward()
        opt_e.step()
    # captured: the same steps, one graph replay per mini-batch
    model = _model()
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
    stepper = PerBatchStep(_step_fn(model, opt, labels, fused_loss=True), table=feat, optimizer=opt)
    steps = 0
    for grp in loader().call_groups():
        stepper.run_group(grp)
        steps += grp.n_batches
    assert steps == 7 and stepper.captures == 1
    for (name, p), q in zip(model.named_parameters(), eager.parameters()):
        scale = float(q.abs().max())
        assert float((p - q).abs().max()) <= 2e-5 * scale, (name, float((p - q).abs().max()), scale)
    # eager inference after the replays sees the stepped weights (derived-weight caches are invalidated by every replay)
    with torch.no_grad():
        d = next(iter(loader()))
        h, hq = d.x, d.x
        for j, (c, q) in enumerate(zip(model, eager)):
            h, hq = c(h, d.edge_index, act="relu" if j == 0 else None), q(hq, d.edge_index, act="relu" if j == 0 else None)
        assert float((h - hq).abs().max()) <= 1e-4 * float(hq.abs().max())


def test_layers_that_are_not_capture_safe_refuse_loudly(hiplib):
    """GATConv / HeteroConv keep derived forms of their parameters in Python-side caches: under capture they raise instead of
    replaying a graph that multiplies with the weights of the capture."""
    import torch
    from wholegraph_amd import nn
    from cugraph_pyg_amd.loader import NeighborLoader, PerBatchStep
    gs, fs, feat = _stores(3000, 10)
    seeds = torch.arange(128).cuda()
    loader = NeighborLoader((fs, gs), [4, 3], input_nodes=seeds, batch_size=64, shuffle=False, random_state=3, local_seeds_per_call=128)
    conv = nn.GATConv(F_IN, 16, heads=4).cuda()

    def step(batch):
        return conv(batch.x, batch.layer_graph(0), act="relu").sum()
    stepper = PerBatchStep(step, table=feat)
    grp = next(iter(loader.call_groups()))
    with pytest.raises(RuntimeError, match="not supported under HIP-graph capture"):
        with torch.no_grad():
            stepper(grp, 0)
    torch.cuda.synchronize()
    # the stream is usable afterwards
    assert float(torch.ones(4, device="cuda").sum()) == 4.0


@pytest.mark.parametrize("F_,N,n,relu", [(256, 47, 1248, False), (256, 256, 3000, True), (160, 64, 500, True), (208, 128, 77, False)])
def test_small_launch_tile_shape_matches_the_throughput_shape(hiplib, F_, N, n, relu):
    """A mini-batch's few thousand rows at a width whose throughput shape is 64-row half tiles (F > 148) run on whole 32-row
    tiles (`relu | WGAMD_SAGE_FULL_TILES`, planes made with full_tiles = 1): same layer — output and kept aggregate against the
    float64 formula at 1e-5 x sum|terms|, and against the half-tile launch of the same operands."""
    import torch
    from wholegraph_amd import nn
    g = torch.Generator().manual_seed(F_ + N + n)
    n_src = 4000
    deg = torch.randint(0, 13, (n,), generator=g)
    row_ptr = torch.zeros(n + 1, dtype=torch.int32)
    row_ptr[1:] = torch.cumsum(deg, 0)
    col = torch.randint(0, n_src, (int(row_ptr[-1]),), generator=g).int()
    self_rows = torch.randperm(n_src, generator=g)[:n]
    x = torch.rand((n_src, F_), generator=g) * 2 - 1
    w_l, w_r, bias = (torch.rand((N, F_), generator=g) - 0.5) * 0.2, (torch.rand((N, F_), generator=g) - 0.5) * 0.2, torch.rand(N, generator=g) - 0.5
    assert nn.sage_layer_small_launch(F_, n) and not nn.sage_layer_small_launch(100, n) and not nn.sage_layer_small_launch(F_, 10 ** 6)
    Np = nn._padded_width(N)
    outs = []
    for full in (False, True):
        prepared = nn.sage_layer_planes(w_l.cuda(), w_r.cuda(), bias.cuda(), Np, full_tiles=full)
        agg = torch.empty((n, F_), dtype=torch.float32, device="cuda")
        out = nn.sage_layer_fused_forward(row_ptr.cuda(), col.cuda(), x.cuda(), self_rows.cuda(), None, relu=relu, mean=True,
                                          agg_out=agg, prepared=prepared)
        outs.append((out.cpu().double(), agg.cpu().double()))
    xd = x.double()
    dst = torch.repeat_interleave(torch.arange(n), deg)
    mean = torch.zeros((n, F_), dtype=torch.float64).index_add_(0, dst, xd[col.long()]) / deg.clamp(min=1).double().unsqueeze(1)
    ref = mean @ w_l.double().t() + xd[self_rows] @ w_r.double().t() + bias.double()
    mag = mean.abs() @ w_l.double().abs().t() + xd[self_rows].abs() @ w_r.double().abs().t() + bias.double().abs()
    if relu:
        ref = ref.clamp(min=0)
    for out, agg in outs:
        assert float(((out - ref).abs() / mag).max()) <= 1e-5
        assert float((agg - mean).abs().max()) <= 1e-5
    assert float((outs[0][0] - outs[1][0]).abs().max()) <= 2e-6 * float(mag.max())
