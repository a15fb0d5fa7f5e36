"""wholegraph_amd.nn.cross_entropy (wgamd_softmax_xent_{forward,backward}_f32): the loss of the reference's training loops
(/root/reference/python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:119-125) as one forward and one backward launch,
against torch.nn.functional.cross_entropy in float64 — value and gradient at 1e-6 relative — for class counts below, at and above
one wave's 64 lanes, ignored (negative) targets, per-row weights (the seed mask of a padded mini-batch), a strided logits view,
and an upstream gradient other than 1.  Run-to-run bit-identical (partial sums are added in workgroup order)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,C", [(1024, 47), (5, 3), (1, 1), (70000, 47), (333, 64), (257, 172), (100, 1000)])
@pytest.mark.parametrize("mode", ["plain", "ignore", "weights"])
def test_cross_entropy_matches_torch_float64(hiplib, n, C, mode):
    import torch
    from wholegraph_amd import nn
    g = torch.Generator(device="cuda").manual_seed(n * 131 + C)
    wide = torch.randn((n, C + 5), generator=g, device="cuda") * 3
    logits = wide[:, :C].detach().requires_grad_(True)              # (row stride C + 5: a view, as h[:batch_size] is)
    target = torch.randint(0, C, (n,), generator=g, device="cuda")
    weight = None
    if mode == "ignore" and n > 1:
        target[torch.rand(n, generator=g, device="cuda") < 0.3] = -100
        target[0] = 0
    if mode == "weights":
        weight = (torch.rand(n, generator=g, device="cuda") < 0.7).float()
        weight[0] = 1.0
    loss = nn.cross_entropy(logits, target, weight)
    (loss * 1.7).backward()
    ref_x = wide[:, :C].double().detach().requires_grad_(True)
    per_row = torch.nn.functional.cross_entropy(ref_x, target, reduction="none", ignore_index=-100)
    w = (target >= 0).double() if weight is None else weight.double() * (target >= 0)
    ref = (per_row * w).sum() / w.sum()
    (ref * 1.7).backward()
    assert abs(float(loss) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref)))
    assert float((logits.grad.double() - ref_x.grad).abs().max()) <= 1e-6 * max(float(ref_x.grad.abs().max()), 1e-12) + 1e-9
    again = nn.cross_entropy(logits.detach(), target, weight)
    assert torch.equal(again, loss.detach())


def test_cross_entropy_inside_a_captured_graph(hiplib):
    """The pair replays inside a HIP graph (the per-mini-batch step): new logits in the same buffer give the new loss and gradient."""
    import torch
    from wholegraph_amd import nn
    x = torch.randn((1024, 47), device="cuda", requires_grad=True)
    t = torch.randint(0, 47, (1024,), device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            x.grad = None
            nn.cross_entropy(x, t).backward()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    x.grad = None
    with torch.cuda.graph(graph):
        loss = nn.cross_entropy(x, t)
        loss.backward()
    for seed in (1, 2):
        with torch.no_grad():
            x.copy_(torch.randn((1024, 47), generator=torch.Generator(device="cuda").manual_seed(seed), device="cuda"))
        graph.replay()
        ref_x = x.detach().double().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(ref_x, t)
        ref.backward()
        assert abs(float(loss) - float(ref)) <= 1e-6 * abs(float(ref))
        assert float((x.grad.double() - ref_x.grad).abs().max()) <= 1e-6 * float(ref_x.grad.abs().max())


@pytest.mark.parametrize("n,H,F_,C", [(20000, 4, 128, 64), (3000, 4, 100, 64), (5000, 1, 256, 47), (70000, 4, 128, 64), (1, 2, 8, 4)])
def test_gat_dense_tail_weight_gradient_on_the_split_k_kernel(hiplib, n, H, F_, C, monkeypatch):
    """nn._heads_transform (the per-head weights of the aggregate-first GAT layer): its weight gradient agg_h^T dY_h — a product
    with hundreds of thousands of rows and a 128 x 64 result — runs on wgamd_sage_wgrad_bf16x3 (the two halves of agg_h's columns
    as its two operands; the whole row twice where the halves would not be 16-byte aligned).  Output and both gradients against
    the float64 einsum at 1e-5 x sum |terms|."""
    import torch
    from wholegraph_amd import nn
    monkeypatch.setattr(nn, "_HEADS_WGRAD_MIN_ROWS", 0)
    g = torch.Generator(device="cuda").manual_seed(n + F_)
    agg = (torch.rand((n, H, F_), generator=g, device="cuda") * 2 - 1).requires_grad_(True)
    w = ((torch.rand((F_, H, C), generator=g, device="cuda") - 0.5) * 0.3).requires_grad_(True)
    gy = torch.rand((n, H, C), generator=g, device="cuda") * 2 - 1
    y = nn._heads_transform(agg, w)
    assert y.grad_fn is not None and "HeadsTransform" in type(y.grad_fn).__name__
    y.backward(gy)
    a64, w64 = agg.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    y64 = torch.einsum("nhf,fhc->nhc", a64, w64)
    y64.backward(gy.double())
    mag_y = torch.einsum("nhf,fhc->nhc", a64.detach().abs(), w64.detach().abs())
    assert float(((y.double() - y64).abs() / (mag_y + 1e-30)).max()) <= 1e-5
    mag_w = torch.einsum("nhf,nhc->fhc", a64.detach().abs(), gy.double().abs())
    assert float(((w.grad.double() - w64.grad).abs() / (mag_w + 1e-30)).max()) <= 1e-5
    mag_a = torch.einsum("nhc,fhc->nhf", gy.double().abs(), w64.detach().abs())
    assert float(((agg.grad.double() - a64.grad).abs() / (mag_a + 1e-30)).max()) <= 1e-5
