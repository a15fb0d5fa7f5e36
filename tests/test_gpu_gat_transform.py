"""wgamd_gat_transform_heads_bf16x3 (csrc/wg_gat_transform.hip): the per-head dense tail of aggregate-first GATConv with
HeteroConv's running sum, bias, ReLU and row placement folded in — against a float64 evaluation of the same expression
(north_star: 1e-5 relative for fp32 aggregation; the bound below is 1e-5 x the row's sum of |a||b|, plus element-wise rtol
1e-5 where the result is not a cancellation), and against the library formulation nn.gat_transform_heads it replaces."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(n, F, H, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    agg = (torch.rand((n, H * F), generator=g) - 0.5) * scale
    w = (torch.rand((F, H * 64), generator=g) - 0.5) * 0.2
    acc = torch.rand((n, H * 64), generator=g) - 0.5
    bias = torch.rand(H * 64, generator=g) - 0.5
    return agg, w, acc, bias


def _ref64(agg, w, H, acc=None, bias=None, relu=False):
    n, F = agg.shape[0], agg.shape[1] // H
    a = agg.double().view(n, H, F)
    b = w.double().view(F, H, 64)
    out = torch.einsum("nhf,fhc->nhc", a, b).reshape(n, H * 64)
    mag = torch.einsum("nhf,fhc->nhc", a.abs(), b.abs()).reshape(n, H * 64)
    if acc is not None:
        out = out + acc.double()
        mag = mag + acc.double().abs()
    if bias is not None:
        out = out + bias.double()
        mag = mag + bias.double().abs()
    if relu:
        out = out.clamp_min(0)
    return out, mag


@pytest.mark.parametrize("F,H", [(128, 4), (256, 4), (64, 4), (128, 1), (128, 2), (128, 3), (256, 8)])
@pytest.mark.parametrize("n", [1, 31, 32, 33, 1000, 70_001])
def test_transform_matches_float64(hiplib, F, H, n):
    from wholegraph_amd import nn
    assert nn.gat_transform_supported(F, H, 64)
    agg, w, acc, bias = _case(n, F, H, seed=F + H + n)
    cu = lambda t: t.cuda()
    for use_acc, use_bias, relu in [(False, False, False), (True, False, False), (True, True, True), (False, True, True)]:
        got = nn.gat_transform_heads_fused(cu(agg), cu(w), H, acc_in=cu(acc) if use_acc else None,
                                           bias=cu(bias) if use_bias else None, relu=relu)
        ref, mag = _ref64(agg, w, H, acc if use_acc else None, bias if use_bias else None, relu)
        err = (got.cpu().double() - ref).abs()
        assert float((err / mag.clamp_min(1e-30)).max()) < 1e-5 * 0.2, (use_acc, use_bias, relu)   # (measured: ~1e-7)
        np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5 * float(mag.max()) * 0.05)


def test_in_place_sum_and_row_placement(hiplib):
    """HeteroConv's chain: first relation writes, the next ones add in place, the last one adds bias, applies ReLU and places
    the rows — against the library formulation (baddbmm + bias_act_rows) it replaces."""
    from wholegraph_amd import nn
    n, F, H = 5000, 128, 4
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(5)
    aggs = [torch.rand((n, H * F), generator=g, device=dev) - 0.5 for _ in range(3)]
    ws = [(torch.rand((F, H * 64), generator=g, device=dev) - 0.5) * 0.2 for _ in range(3)]
    bias = torch.rand(H * 64, generator=g, device=dev) - 0.5
    rows = torch.randperm(2 * n, generator=g, device=dev)[:n]
    # library chain
    acc = torch.empty((n, H * 64), device=dev)
    for j in range(3):
        nn.gat_transform_heads(aggs[j], ws[j], H, out=acc, overwrite=j == 0, fused=False)
    want = torch.zeros((2 * n, H * 64), device=dev)
    nn.bias_act_rows(acc, bias, True, rows, want)
    # fused chain
    acc2 = torch.empty((n, H * 64), device=dev)
    got = torch.zeros((2 * n, H * 64), device=dev)
    nn.gat_transform_heads_fused(aggs[0], ws[0], H, out=acc2)
    nn.gat_transform_heads_fused(aggs[1], ws[1], H, acc_in=acc2, out=acc2)
    nn.gat_transform_heads_fused(aggs[2], ws[2], H, acc_in=acc2, bias=bias, relu=True, out_rows=rows, out=got)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=2e-5)
    untouched = torch.ones(2 * n, dtype=torch.bool, device=dev)
    untouched[rows] = False
    assert float(got[untouched].abs().max()) == 0.0
    # the drop-in form: nn.gat_transform_heads takes the fused kernel by default
    acc3 = torch.empty((n, H * 64), device=dev)
    for j in range(3):
        nn.gat_transform_heads(aggs[j], ws[j], H, out=acc3, overwrite=j == 0)
    torch.testing.assert_close(acc3, acc2 if False else acc, rtol=1e-5, atol=2e-5)


def test_adversarial_magnitudes(hiplib):
    """Rows whose terms span 2^-12 .. 2^12: the split product stays inside fp32 round-off of the row's sum of |a||b|."""
    from wholegraph_amd import nn
    n, F, H = 4096, 128, 4
    g = torch.Generator().manual_seed(9)
    agg = (torch.rand((n, H * F), generator=g) - 0.5) * torch.exp2(torch.randint(-12, 13, (n, H * F), generator=g).float())
    w = (torch.rand((F, H * 64), generator=g) - 0.5) * torch.exp2(torch.randint(-6, 7, (F, H * 64), generator=g).float())
    got = nn.gat_transform_heads_fused(agg.cuda(), w.cuda(), H)
    ref, mag = _ref64(agg, w, H)
    assert float(((got.cpu().double() - ref).abs() / mag).max()) < 2e-6


def _gat_ref64(rp, col, x, a_src, a_dst, w, H, slope, dst_rows=None):
    """float64: aggregate-first GATConv relation, row by row."""
    n, F = rp.shape[0] - 1, x.shape[1]
    out = np.zeros((n, H * 64))
    mag = np.zeros((n, H * 64))
    x64, w64 = x.double().numpy(), w.double().numpy().reshape(F, H, 64)
    a_s, a_d = a_src.double().numpy(), a_dst.double().numpy()
    for i in range(n):
        s, e = int(rp[i]), int(rp[i + 1])
        if e == s:
            continue
        nb = col[s:e].numpy()
        sc = a_s[nb] + a_d[int(dst_rows[i]) if dst_rows is not None else i][None, :]
        sc = np.where(sc > 0, sc, sc * slope)
        p = np.exp(sc - sc.max(0, keepdims=True))
        alpha = p / p.sum(0, keepdims=True)                      # [deg, H]
        agg = np.einsum("eh,ef->hf", alpha, x64[nb])             # [H, F]
        out[i] = np.einsum("hf,fhc->hc", agg, w64).reshape(-1)
        mag[i] = np.einsum("hf,fhc->hc", np.einsum("eh,ef->hf", alpha, np.abs(x64[nb])), np.abs(w64)).reshape(-1)
    return out, mag


@pytest.mark.parametrize("n,maxdeg", [(1, 10), (31, 10), (32, 10), (33, 10), (3000, 10), (9000, 10), (700, 25), (300, 70)])
def test_gat_layer_fused_matches_float64_and_two_kernels(hiplib, n, maxdeg):
    """wgamd_gat_layer_fused_bf16x3 against float64 and against the two kernels it fuses (empty rows, rows past the register
    window, dst_rows indirection, the HeteroConv tail)."""
    from wholegraph_amd import nn
    F, H, n_src = 128, 4, 5000
    g = torch.Generator().manual_seed(n + maxdeg)
    deg = torch.randint(0, maxdeg + 1, (n,), generator=g)
    if n > 2:
        deg[1] = 0
    rp = torch.zeros(n + 1, dtype=torch.int32)
    rp[1:] = torch.cumsum(deg, 0)
    E = int(rp[-1])
    if E == 0:
        deg[0] = 3
        rp[1:] = torch.cumsum(deg, 0)
        E = int(rp[-1])
    col = torch.randint(0, n_src, (E,), generator=g, dtype=torch.int32)
    x = torch.rand((n_src, F), generator=g) - 0.5
    a_src = (torch.rand((n_src, H), generator=g) - 0.5) * 4
    a_dst = (torch.rand((2 * n, H), generator=g) - 0.5) * 4
    dst_rows = torch.randperm(2 * n, generator=g)[:n]
    w = (torch.rand((F, H * 64), generator=g) - 0.5) * 0.2
    acc = torch.rand((n, H * 64), generator=g) - 0.5
    bias = torch.rand(H * 64, generator=g) - 0.5
    out_rows = torch.randperm(n + 7, generator=g)[:n]
    cu = lambda t: t.cuda()
    ref, mag = _gat_ref64(rp, col, x, a_src, a_dst, w, H, 0.2, dst_rows)
    got = nn.gat_layer_fused(cu(rp), cu(col), cu(x), cu(a_src), cu(a_dst), cu(w), H, dst_rows=cu(dst_rows))
    err = np.abs(got.cpu().double().numpy() - ref)
    assert float((err / np.maximum(mag, 1e-30)).max()) < 1e-5, "fused vs float64"
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * float(mag.max()) * 0.1)
    # the two kernels it replaces (their softmax is the online one: fp32 reassociation apart)
    agg = nn.gat_aggregate_heads(cu(rp), cu(col), cu(x), cu(a_src), cu(a_dst), H, dst_rows=cu(dst_rows))
    two = nn.gat_transform_heads_fused(agg, cu(w), H)
    torch.testing.assert_close(got, two, rtol=1e-5, atol=1e-5 * float(mag.max()))
    # the HeteroConv tail: running sum, bias, ReLU, row placement
    want = torch.zeros((n + 7, H * 64)).cuda()
    nn.gat_transform_heads_fused(agg, cu(w), H, acc_in=cu(acc), bias=cu(bias), relu=True, out_rows=cu(out_rows), out=want)
    full = torch.zeros((n + 7, H * 64)).cuda()
    nn.gat_layer_fused(cu(rp), cu(col), cu(x), cu(a_src), cu(a_dst), cu(w), H, dst_rows=cu(dst_rows), acc_in=cu(acc), bias=cu(bias),
                       relu=True, out_rows=cu(out_rows), out=full)
    torch.testing.assert_close(full, want, rtol=1e-5, atol=1e-5 * float(mag.max()))
