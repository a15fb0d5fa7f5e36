"""GPU parity of the TRAINING path of the one-kernel SAGE layer: weight-gradient kernel (csrc/wg_sage_bwd.hip), input gradient
over the transposed hop, and the autograd Function of ``wholegraph_amd.nn.SAGEConv`` — every gradient against the fp64
restatement of the PyG formulas (north_star's 1e-5: |err| <= 1e-5 x the magnitude sum of the terms, and 1e-5 relative on the
elements that are not cancellations), the forward bit for bit the inference launch.  Semantics: torch_geometric.nn.SAGEConv
as the reference's models train it (python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:25-59,119-125)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hop(n_dst, n_src, max_deg, seed, hubs=True):
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    deg = torch.randint(0, max_deg + 1, (n_dst,), generator=g, device="cuda")
    deg[:3] = torch.tensor([0, 1, max_deg], device="cuda")
    rp = torch.zeros(n_dst + 1, dtype=torch.int32, device="cuda")
    rp[1:] = torch.cumsum(deg, 0)
    E = int(rp[-1])
    col = torch.randint(0, n_src, (E,), generator=g, device="cuda", dtype=torch.int32)
    if hubs:    # a power-law hop: a few input rows feed a large share of the destinations
        u = torch.rand(E, generator=g, device="cuda")
        col[u < 0.10] = 5
        col[(u >= 0.10) & (u < 0.13)] = n_src - 2
    self_rows = torch.randperm(n_src, generator=g, device="cuda")[:n_dst].contiguous()
    return rp, col, self_rows


def _close(got, ref, scale, what):
    import torch
    err = (got.double() - ref).abs()
    assert bool((err <= 1e-5 * scale + 1e-7).all()), (what, float((err - 1e-5 * scale).max()))
    big = (ref.abs() >= 0.1 * scale) & (scale > 0)      # (an input row no edge reads: gradient and scale exactly 0)
    if int(big.sum()) > 0:
        assert float((err[big] / ref.abs()[big]).max()) <= 1e-5, what
    assert bool(torch.isfinite(got).all()), what


@pytest.mark.parametrize("F,N", [(100, 256), (128, 256), (256, 256), (256, 47), (100, 47), (64, 16), (200, 172), (4, 1)])
@pytest.mark.parametrize("ids", [None, "int32", "int64"])
def test_wgrad_kernel_vs_fp64(hiplib, F, N, ids):
    """grad_w_l = dZ^T agg, grad_w_r = dZ^T X[self], grad_bias = colsum(dZ) with the ReLU mask folded in, for every tile plan
    of the kernel (feature tiles per wave 4 / 7 / 8, 1-8 column waves, 32- and 16-row tiles, two N-blocks), ragged row counts,
    with and without the id indirection; accumulate adds; two runs are bit-identical."""
    import torch
    from wholegraph_amd import nn
    n = 5003 if F < 256 else 3001
    g = torch.Generator(device="cuda").manual_seed(F * 1000 + N)
    n_src = 9000
    x = torch.randn((n_src, F), generator=g, device="cuda")
    agg = torch.randn((n, F), generator=g, device="cuda")
    self_rows = torch.randint(0, 4000, (n,), generator=g, device="cuda")
    src_ids = None if ids is None else torch.randperm(n_src, generator=g, device="cuda")[:4000].to(
        torch.int32 if ids == "int32" else torch.int64)
    gout = torch.randn((n, N), generator=g, device="cuda")
    act = torch.randn((n, N), generator=g, device="cuda")          # "ReLU output": its sign pattern is the mask
    xs = x[self_rows] if src_ids is None else x[src_ids.long()[self_rows]]
    for mask in (None, act):
        dz = gout.double() if mask is None else gout.double() * (mask > 0)
        a64 = torch.cat([agg, xs], 1).double()
        ref = dz.t() @ a64                                          # [N, 2F]
        scale = dz.abs().t() @ a64.abs()
        gwl, gwr, gb = torch.full((N, F), 7.0, device="cuda"), torch.full((N, F), 7.0, device="cuda"), torch.full((N,), 7.0, device="cuda")
        nn.sage_wgrad(agg, x, self_rows, gout, gwl, gwr, gb, act_out=mask, src_ids=src_ids)
        _close(gwl, ref[:, :F], scale[:, :F], "grad_w_l")
        _close(gwr, ref[:, F:], scale[:, F:], "grad_w_r")
        _close(gb, dz.sum(0), dz.abs().sum(0), "grad_bias")
        a, b, c = gwl.clone(), gwr.clone(), gb.clone()
        nn.sage_wgrad(agg, x, self_rows, gout, gwl, gwr, gb, act_out=mask, src_ids=src_ids)
        assert torch.equal(a, gwl) and torch.equal(b, gwr) and torch.equal(c, gb), "not deterministic"
        nn.sage_wgrad(agg, x, self_rows, gout, gwl, gwr, gb, act_out=mask, src_ids=src_ids, accumulate=True)
        assert torch.equal(gwl, a + a) and torch.equal(gwr, b + b) and torch.equal(gb, c + c)
        nn.sage_wgrad(agg, x, self_rows, gout, gwl, gwr, None, act_out=mask, src_ids=src_ids)     # no bias
        assert torch.equal(a, gwl) and torch.equal(b, gwr)


def test_wgrad_kernel_tiny_and_empty(hiplib):
    import torch
    from wholegraph_amd import nn
    for n in (0, 1, 31, 33):
        F, N = 100, 256
        x = torch.randn((50, F), device="cuda")
        agg, gout = torch.randn((n, F), device="cuda"), torch.randn((n, N), device="cuda")
        self_rows = torch.arange(n, device="cuda")
        gwl, gwr, gb = torch.full((N, F), 3.0, device="cuda"), torch.full((N, F), 3.0, device="cuda"), torch.full((N,), 3.0, device="cuda")
        nn.sage_wgrad(agg, x, self_rows, gout, gwl, gwr, gb)
        torch.testing.assert_close(gwl.double(), gout.double().t() @ agg.double(), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(gwr.double(), gout.double().t() @ x[:n].double(), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(gb.double(), gout.double().sum(0), rtol=1e-5, atol=1e-5)


def _ref_layer(x64, hops, wl, wr, b, relu, mean):
    """fp64 SAGEConv over the hops of a LayerGraph (PyG: lin_l(mean_j x_j) + lin_r(x_i)) — and the same with every term
    replaced by its magnitude (the tolerance scale)."""
    import torch
    outs = []
    for rp, col, self_rows in hops:
        n = rp.shape[0] - 1
        deg = (rp[1:] - rp[:-1]).long()
        dst = torch.repeat_interleave(torch.arange(n, device=x64.device), deg)
        agg = torch.zeros((n, x64.shape[1]), dtype=x64.dtype, device=x64.device).index_add_(0, dst, x64[col.long()])
        if mean:
            agg = agg / deg.clamp(min=1).unsqueeze(1)
        o = agg @ wl.t() + x64[self_rows] @ wr.t()
        outs.append(o + b if b is not None else o)
    out = torch.cat(outs)
    return torch.relu(out) if relu else out


@pytest.mark.parametrize("F,N,relu,mean,lazy", [(100, 256, True, True, "int64"), (100, 256, True, True, "int32"),
                                                 (256, 47, False, True, None), (128, 128, True, False, None),
                                                 (256, 256, True, True, None), (64, 64, True, True, None)])
def test_sage_layer_autograd_vs_fp64(hiplib, F, N, relu, mean, lazy):
    """``nn.SAGEConv`` over a two-hop LayerGraph under autograd: the output is bit for bit the no-grad (inference) launch, and
    d/d{lin_l.weight, lin_r.weight, bias, x} match the fp64 autograd of the dense formula — x as a resident tensor (its
    gradient runs the layer kernel over the transposed hops, hub rows included) or as LazyRows (table read through ids: no
    input gradient)."""
    import torch
    from wholegraph_amd import nn
    n_src = 6000
    hops = [_hop(1500, n_src, 10, F + N), _hop(700, n_src, 25, F + N + 1)]
    lg = nn.LayerGraph([nn.HopGraph(*h) for h in hops])
    g = torch.Generator(device="cuda").manual_seed(F + 3 * N)
    torch.manual_seed(F * 7 + N)      # (the layer's initial weights)
    conv = nn.SAGEConv(F, N, aggr="mean" if mean else "sum").cuda()
    if lazy:
        table = torch.randn((50000, F), generator=g, device="cuda")
        ids = torch.randperm(50000, generator=g, device="cuda")[:n_src].to(torch.int32 if lazy == "int32" else torch.int64)
        x_in, x64 = nn.LazyRows(table, ids), table[ids.long()].double()
    else:
        x_in = torch.randn((n_src, F), generator=g, device="cuda", requires_grad=True)
        x64 = x_in.detach().double().requires_grad_(True)
    with torch.no_grad():
        want = conv(x_in, lg, act="relu" if relu else None).clone()
    out = conv(x_in, lg, act="relu" if relu else None)
    assert out.requires_grad and torch.equal(out, want), "training forward differs from the inference launch"
    gout = torch.randn(out.shape, generator=g, device="cuda")
    out.backward(gout)

    wl, wr, b = (p.detach().double().requires_grad_(True) for p in (conv.lin_l.weight, conv.lin_r.weight, conv.lin_l.bias))
    pre = _ref_layer(x64, hops, wl, wr, b, False, mean)
    fscale = _ref_layer(x64.detach().abs(), hops, wl.detach().abs(), wr.detach().abs(), b.detach().abs(), False, mean)
    # The ReLU mask of the reference is the one the fp32 forward produced: where the pre-activation is within fp32 round-off
    # of zero (about one element in a million, so every other run of this test) fp32 and fp64 may disagree on its sign, and
    # the whole gradient column would differ by that one term.  The disagreement itself is held to the forward tolerance.
    mask = (out.detach() > 0) if relu else torch.ones_like(out, dtype=torch.bool)
    flip = mask != (pre.detach() > 0) if relu else torch.zeros_like(mask)
    assert bool((pre.detach().abs()[flip] <= 1e-5 * fscale[flip] + 1e-7).all()), "ReLU mask differs away from the kink"
    ref = pre * mask
    ref.backward(gout.double())
    _close(out.detach(), ref.detach(), fscale, "forward")
    # tolerance scales: the same gradient formulas over magnitudes
    dz = gout.double() * mask
    xa, wla, wra = (t.detach().abs().requires_grad_(True) for t in (x64, wl, wr))
    ba = b.detach().abs().requires_grad_(True)
    _ref_layer(xa, hops, wla, wra, ba, False, mean).backward(dz.abs())
    _close(conv.lin_l.weight.grad, wl.grad, wla.grad, "grad lin_l.weight")
    _close(conv.lin_r.weight.grad, wr.grad, wra.grad, "grad lin_r.weight")
    _close(conv.lin_l.bias.grad, b.grad, ba.grad, "grad bias")
    if not lazy:
        _close(x_in.grad, x64.grad, xa.grad, "grad x")
    # a second backward pass over the same graph: bit-identical (no atomics anywhere)
    first = [p.grad.clone() for p in conv.parameters()]
    for p in conv.parameters():
        p.grad = None
    conv(x_in, lg, act="relu" if relu else None).backward(gout)
    assert all(torch.equal(a, p.grad) for a, p in zip(first, conv.parameters()))


def test_sage_conv_edge_index_trains_on_the_one_kernel_layer(hiplib):
    """The reference's call shape — ``conv((x, x[:n]), [csr_row_ptr, csr_col_ind])`` and ``conv(x, edge_index)`` — under
    autograd: same gradients as the split formulation (aggregation kernel + torch Linear)."""
    import torch
    from wholegraph_amd import nn
    rp, col, _ = _hop(900, 900, 10, 11, hubs=False)
    n_src = 3000
    col = torch.randint(0, n_src, col.shape, device="cuda", dtype=torch.int32)
    conv = nn.SAGEConv(64, 128).cuda()
    x = torch.randn((n_src, 64), device="cuda", requires_grad=True)
    out = conv((x, x[:900]), [rp, col], act="relu")
    gout = torch.randn_like(out)
    out.backward(gout)
    got = [x.grad.clone()] + [p.grad.clone() for p in conv.parameters()]
    x.grad = None
    for p in conv.parameters():
        p.grad = None
    ref = torch.relu(conv.lin_l(nn.spmm_csr(x, rp, col, "mean")) + conv.lin_r(x[:900]))
    ref.backward(gout)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)
    for a, b in zip(got, [x.grad] + [p.grad for p in conv.parameters()]):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4)
    deg = (rp[1:] - rp[:-1]).long()
    ei = torch.stack([col.long(), torch.repeat_interleave(torch.arange(900, device="cuda"), deg)])
    out2 = conv(x, ei, act="relu")[:900]
    torch.testing.assert_close(out2, ref, rtol=1e-4, atol=1e-4)


def test_hop_transpose_is_computed_once(hiplib):
    import torch
    from wholegraph_amd import nn
    rp, col, self_rows = _hop(500, 2000, 8, 3)
    h = nn.HopGraph(rp, col, self_rows)
    a = h.transposed(2000)
    assert h.transposed(2000)[0] is a[0]
    row_ptr_t, col_t, self_t = a
    assert int(row_ptr_t[-1]) == col.shape[0] and torch.equal(torch.diff(row_ptr_t).long(), torch.bincount(col, minlength=2000))
    assert torch.equal(self_t[self_rows], torch.arange(500, 1000, device="cuda")) and int((self_t == 1000).sum()) == 1500


def test_two_layer_model_learns(hiplib):
    """A 2-layer SAGE model over a call-group-shaped LayerGraph pair with SGD (features lazy, ReLU between the layers, the
    47-class head padded to 64 columns inside the kernels): the loss goes down at every step."""
    import torch
    from wholegraph_amd import nn
    n0, n1, n2 = 8000, 2000, 256
    h_deep = nn.HopGraph(*_hop(n1, n0, 10, 21))
    h_top = nn.HopGraph(*_hop(n2, n1, 25, 22))
    table = torch.randn((30000, 100), device="cuda")
    ids = torch.randperm(30000, device="cuda")[:n0]
    convs = torch.nn.ModuleList([nn.SAGEConv(100, 256), nn.SAGEConv(256, 47)]).cuda()
    y = torch.randint(0, 47, (n2,), device="cuda")
    opt = torch.optim.SGD(convs.parameters(), lr=0.05)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        h = convs[0](nn.LazyRows(table, ids), nn.LayerGraph([h_deep]), act="relu")
        h = convs[1](h, nn.LayerGraph([h_top]))
        loss = torch.nn.functional.cross_entropy(h, y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(b < a for a, b in zip(losses, losses[1:])) and losses[-1] < losses[0] - 0.1, losses
