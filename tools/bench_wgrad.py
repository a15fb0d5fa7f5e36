"""Stand-alone timing (HIP events) of the training kernels at the products call-group shapes: weight gradient of layer 1
(1.55 M rows, F = 100 -> 256, ReLU mask, table read through ids) and layer 2 (196 k rows, 256 -> 47), the layer-1 forward with
and without the kept aggregate.  usage: python tools/bench_wgrad.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch  # noqa: E402

from wholegraph_amd import nn  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, n=8):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        torch.cuda._sleep(200_000)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


V = 2_449_029
table = torch.rand((V, 100), generator=g, device=dev) * 2 - 1
for (n, F, N, mask, lazy) in ((1_550_000, 100, 256, True, True), (195_584, 256, 47, False, False), (1_760_000, 256, 256, True, False)):
    n_src = 10_900_000 if lazy else 1_760_000
    x = table if lazy else torch.rand((n_src, F), generator=g, device=dev)
    ids = torch.randint(0, V, (n_src,), generator=g, device=dev) if lazy else None
    agg = torch.rand((n, F), generator=g, device=dev)
    self_rows = torch.randperm(n_src, generator=g, device=dev)[:n].contiguous()
    gout = torch.randn((n, N), generator=g, device=dev)
    act = torch.randn((n, N), generator=g, device=dev) if mask else None
    gwl, gwr, gb = torch.empty((N, F), device=dev), torch.empty((N, F), device=dev), torch.empty(N, device=dev)
    ms = timeit(lambda: nn.sage_wgrad(agg, x, self_rows, gout, gwl, gwr, gb, act_out=act, src_ids=ids))
    byts = n * (2 * F * 4 + 8 + N * 4 * (2 if mask else 1))
    flops = 6 * 2.0 * n * 2 * F * N
    print("wgrad n=%d F=%d N=%d mask=%s ids=%s: %.3f ms  %.2f TB/s  %.0f TF/s(bf16 x6)" % (n, F, N, mask, lazy, ms, byts / ms / 1e9, flops / ms / 1e9))
    del x, agg, gout, act
