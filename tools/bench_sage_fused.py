#!/usr/bin/env python
"""Micro-benchmark of the one-kernel SAGE layer (wgamd_sage_layer_fused_f32) against aggregate kernel + library GEMM at
the layer-1 shape of one products call group (610 k destination rows, ~8 neighbours, F = 100 -> 256)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch  # noqa: E402
from wholegraph_amd import nn  # noqa: E402


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    F, N = int(os.environ.get("F", 100)), int(os.environ.get("N", 256))
    n_dst, n_src, V = int(os.environ.get("ND", 610_000)), int(os.environ.get("NS", 3_650_000)), 2_449_029
    deg = torch.randint(5, 11, (n_dst,), generator=g, device=dev)   # hop-2 rows: fan-out 10, mean ~8
    rp = torch.zeros(n_dst + 1, dtype=torch.int32, device=dev)
    rp[1:] = torch.cumsum(deg, 0)
    E = int(rp[-1])
    col = torch.randint(0, n_src, (E,), generator=g, device=dev, dtype=torch.int32)
    x = torch.rand((n_src, F), generator=g, device=dev)
    table = torch.rand((V, F), generator=g, device=dev)
    n_id = torch.randint(0, V, (n_src,), generator=g, device=dev)
    rows = torch.randint(0, n_src, (n_dst,), generator=g, device=dev)
    w_t = torch.rand((2 * F, N), generator=g, device=dev) - 0.5
    bias = torch.rand(N, generator=g, device=dev)
    t_agg = timed(lambda: nn.sage_aggregate_forward(rp, col, x, rows, True))
    cat = nn.sage_aggregate_forward(rp, col, x, rows, True)
    t_gemm = timed(lambda: torch._addmm_activation(bias, cat, w_t, use_gelu=False))
    t_fused = timed(lambda: nn.sage_layer_fused_forward(rp, col, x, rows, w_t, bias, relu=True))
    t_aggf = timed(lambda: nn.sage_aggregate_fetch_forward(rp, col, table, n_id, rows, True))
    t_fusedf = timed(lambda: nn.sage_layer_fused_forward(rp, col, table, rows, w_t, bias, relu=True, src_ids=n_id))
    flops = 2.0 * n_dst * 2 * F * N
    print(f"E={E}  aggregate {t_agg:.3f} ms + gemm {t_gemm:.3f} ms = {t_agg + t_gemm:.3f} | fused {t_fused:.3f} ms "
          f"({flops / t_fused / 1e9:.1f} TF/s) || fetch-aggregate {t_aggf:.3f} + gemm = {t_aggf + t_gemm:.3f} | fused-fetch {t_fusedf:.3f} ms")


if __name__ == "__main__":
    main()
