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
