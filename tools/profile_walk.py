"""Walk-only pass of the products-like workload (the call-group walk of bench.py, nothing next to it): run it under
`rocprofv3 --kernel-trace --stats` to see what each of the walk's launches costs when it has the chip to itself.
    G=191 python tools/profile_walk.py          (prints ms per call group by HIP events)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from wholegraph_amd import fused  # noqa: E402  (bench.py put the package on sys.path)


def main():
    dev = torch.device("cuda:0")
    G = int(os.environ.get("G", 191))
    iters = int(os.environ.get("ITERS", 20))
    wv, we, _, _, fanout = bench.WORKLOADS[os.environ.get("WORKLOAD", "products")]
    bench.FANOUT = fanout
    row_ptr, col = bench.rmat_csr(wv, we, seed=0, device=dev)
    col = col.to(torch.int64)
    walk = fused.NoSyncWalk(row_ptr, col, bench.BATCH, bench.FANOUT, col.dtype, G, pad_unique=os.environ.get("PAD", "0") == "1")
    g = torch.Generator(device=dev).manual_seed(3)
    seeds = torch.randint(0, row_ptr.numel() - 1, (G * bench.BATCH,), generator=g, device=dev, dtype=col.dtype)
    hops = len(fanout)
    rs = (torch.arange(G, device=dev, dtype=torch.int64).view(1, -1) * hops + torch.arange(hops, device=dev, dtype=torch.int64).view(-1, 1) + 62)
    from wholegraph_amd.tensor import unique_bounded_nosync
    dedup = os.environ.get("DEDUP", "0") == "1"      # + the de-duplication of the group's node list (bench.py's headline fetch)

    def one(i):
        res = walk.run(seeds, rs + i * 7)
        if dedup:
            unique_bounded_nosync(res.unique[hops - 1], res.counts[hops - 1][1:2], row_ptr.numel() - 1)
    for i in range(5):
        one(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        one(i + 5)
    e.record()
    torch.cuda.synchronize()
    print("G=%d  walk %.3f ms per call group (%.3f per 64 mini-batches)" % (G, s.elapsed_time(e) / iters, s.elapsed_time(e) / iters * 64 / G))


if __name__ == "__main__":
    main()
