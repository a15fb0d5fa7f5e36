"""A/B of the two gather+terms kernels (csrc/wg_gather_terms.hip) at the ogbn-mag call-group shapes:
   WGAMD_GATHER_TERMS_PIPELINED=0|1 python tools/bench_gather_terms.py
Prints ms per launch, the plain row gather of the same rows next to it, and a checksum of the outputs (the two kernels issue
the same MFMAs in the same order: the checksums must be equal)."""
import hashlib
import os
import sys

import torch

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cugraph-gnn_amd")]
from wholegraph_amd import nn  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)


def timed(fn, iters=20):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


print("pipelined =", os.environ.get("WGAMD_GATHER_TERMS_PIPELINED", "1 (default)"))
for name, rows, n, T in (("paper", 736_389, 10_040_933, 24), ("author", 1_134_649, 3_854_993, 12),
                         ("field_of_study", 59_965, 2_254_250, 8)):
    table = torch.randn((rows, 128), generator=g, device=dev)
    ids = torch.randint(0, rows, (n,), generator=g, device=dev)
    ids[::1001] = -1
    v = torch.randn((128, T), generator=g, device=dev) * 0.1
    out = torch.empty((n, 128), device=dev)
    x, terms = nn.gather_with_terms(table, ids, v, out=out, heads=4)
    torch.cuda.synchronize()
    keep = ids >= 0
    assert torch.equal(x[keep], table[ids[keep]])
    want = (table[ids[keep]][:100000].double() @ v.double()).float()
    got = terms.permute(1, 0, 2).reshape(n, T)[keep][:100000]
    err = float((got - want).abs().max())
    digest = hashlib.sha256(terms.cpu().numpy().tobytes()).hexdigest()[:16]
    t_both = timed(lambda: nn.gather_with_terms(table, ids, v, out=out, heads=4))
    t_plain = timed(lambda: torch.index_select(table, 0, ids.clamp(min=0), out=out))
    nbytes = n * 128 * 4 * 2 + n * T * 4
    print("%-15s n=%9d T=%2d  gather+terms %.3f ms (%.2f TB/s)  torch index_select %.3f ms  max err vs fp64 %.2e  terms sha %s"
          % (name, n, T, t_both, nbytes / t_both / 1e9, t_plain, err, digest))
