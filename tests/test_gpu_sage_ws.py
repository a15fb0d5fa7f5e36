"""The weight-stationary form of the one-kernel SAGE layer (csrc/wg_sage_ws.hip, opt-in WGAMD_SAGE_WS=1) returns the SAME BITS as
the producer / consumer kernel it is an alternative to (same products, same accumulation order) — plain x, x behind 32-bit
offsets, the fetch-folded variant with int64 / int32 ids, rows past the register window, empty rows, a ragged last tile.  The
switch is read once per process: each arm runs in its own interpreter."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ARM = r'''
import os, sys, torch
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "cugraph-gnn_amd")]
from wholegraph_amd import nn
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(11)
F, N, n_dst, n_src, V = 100, 256, 40_007, 90_000, 150_000
deg = torch.randint(0, 11, (n_dst,), generator=g, device=dev)
deg[::97] = 37
deg[3::101] = 0
rp = torch.zeros(n_dst + 1, dtype=torch.int32, device=dev)
rp[1:] = torch.cumsum(deg, 0)
col = torch.randint(0, n_src, (int(rp[-1]),), generator=g, device=dev, dtype=torch.int32)
x = torch.rand((n_src, F), generator=g, device=dev) - 0.5
table = torch.rand((V, F), generator=g, device=dev) - 0.5
n_id = torch.randint(0, V, (n_src,), generator=g, device=dev)
rows = torch.randint(0, n_src, (n_dst,), generator=g, device=dev)
w_t = torch.rand((2 * F, N), generator=g, device=dev) - 0.5
bias = torch.rand(N, generator=g, device=dev)
out = {
    "plain": nn.sage_layer_fused_forward(rp, col, x, rows, w_t, bias, relu=True),
    "sum": nn.sage_layer_fused_forward(rp, col, x, rows, w_t, None, relu=False, mean=False),
    "fetch64": nn.sage_layer_fused_forward(rp, col, table, rows, w_t, bias, relu=True, src_ids=n_id),
    "fetch32": nn.sage_layer_fused_forward(rp, col, table, rows, w_t, bias, relu=True, src_ids=n_id.to(torch.int32)),
}
torch.save({k: v.cpu() for k, v in out.items()}, sys.argv[1])
'''


def test_weight_stationary_kernel_is_bit_identical(tmp_path):
    import torch
    outs = []
    for ws in ("0", "1"):
        path = str(tmp_path / ("ws%s.pt" % ws))
        env = dict(os.environ, WGAMD_SAGE_WS=ws)
        subprocess.check_call([sys.executable, "-c", ARM % {"root": ROOT}, path], env=env)
        outs.append(torch.load(path))
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
        assert bool(torch.isfinite(outs[0][k]).all())
