#!/usr/bin/env python
"""A/B of the weight-stationary one-kernel SAGE layer (wg_sage_ws.hip, WGAMD_SAGE_WS=1) against the producer / consumer
kernel (WGAMD_SAGE_WS=0) at the layer-1 shape of a products call group: bit-equality of the outputs and HIP-event time.
The switch is read once per process, so each arm runs in its own subprocess."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def arm():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
    import torch
    from wholegraph_amd import nn
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    F, N = 100, 256
    n_dst, n_src, V = int(os.environ.get("ND", 1_640_000)), int(os.environ.get("NS", 10_900_000)), 2_449_029
    deg = torch.randint(5, 11, (n_dst,), generator=g, device=dev)
    if os.environ.get("AB_LONG"):   # (correctness arm: a few rows past both neighbour windows, a few empty ones)
        deg[::1000] = 37
        deg[5::1000] = 0
    rp = torch.zeros(n_dst + 1, dtype=torch.int32, device=dev)
    rp[1:] = torch.cumsum(deg, 0)
    E = int(rp[-1])
    col = torch.randint(0, n_src, (E,), generator=g, device=dev, dtype=torch.int32)
    x = torch.rand((n_src, F), generator=g, device=dev) - 0.5
    table = torch.rand((V, F), generator=g, device=dev) - 0.5
    n_id = torch.randint(0, V, (n_src,), generator=g, device=dev)
    rows = torch.randint(0, n_src, (n_dst,), generator=g, device=dev)
    w_t = torch.rand((2 * F, N), generator=g, device=dev) - 0.5
    bias = torch.rand(N, generator=g, device=dev)

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

    out = {}
    for name, fn in (("x>2GB", lambda: nn.sage_layer_fused_forward(rp, col, x, rows, w_t, bias, relu=True)),
                     ("x<2GB", lambda: nn.sage_layer_fused_forward(rp, col, x[:5_000_000], rows % 5_000_000, w_t, bias, relu=True)
                      if False else nn.sage_layer_fused_forward(rp, col % 5_000_000, x[:5_000_000], rows % 5_000_000, w_t, bias, relu=True)),
                     ("fetch", lambda: nn.sage_layer_fused_forward(rp, col, table, rows, w_t, bias, relu=True, src_ids=n_id))):
        y = fn()
        t = timed(fn)
        out[name] = (t, y)
        print("%s %-6s %.4f ms  checksum %.9e" % (os.environ.get("WGAMD_SAGE_WS", "default"), name, t, float(y.double().sum())),
              flush=True)
    torch.save({k: v[1][:200_000].cpu() for k, v in out.items()}, os.environ["AB_OUT"])


def main():
    outs = []
    for ws in ("0", "1"):
        path = "/tmp/ab_ws_%s.pt" % ws
        env = dict(os.environ, WGAMD_SAGE_WS=ws, AB_OUT=path, AB_ARM="1")
        subprocess.check_call([sys.executable, os.path.abspath(__file__)], env=env)
        outs.append(path)
    import torch
    a, b = torch.load(outs[0]), torch.load(outs[1])
    for k in a:
        print("bit-equal %-6s %s  (max |diff| %.3e)" % (k, bool(torch.equal(a[k], b[k])), float((a[k] - b[k]).abs().max())))


if __name__ == "__main__":
    arm() if os.environ.get("AB_ARM") else main()
