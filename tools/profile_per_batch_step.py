"""Per-mini-batch training step (cugraph_pyg_amd.loader.PerBatchStep) on the products workload, alone: run under
`rocprofv3 --kernel-trace --stats` to see what one replay is made of.   TRAIN=0: the forward alone."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    G = int(os.environ.get("G", 191))
    n_groups = int(os.environ.get("GROUPS", 2))
    which = ("train_step_per_batch",) if os.environ.get("TRAIN", "1") == "1" else ("forward_per_batch",)
    wv, we, F_, classes, fanout = bench.WORKLOADS["products"]
    bench.FANOUT, bench.FEAT_DIM, bench.CLASSES = fanout, F_, classes
    row_ptr, col = bench.rmat_csr(wv, we, seed=0, device=dev)
    table = torch.rand((wv, F_), generator=torch.Generator(device=dev).manual_seed(0), device=dev) * 2 - 1
    from wholegraph_amd import nn
    convs = torch.nn.ModuleList([nn.SAGEConv(F_, bench.HIDDEN), nn.SAGEConv(bench.HIDDEN, classes)]).to(dev)
    for c in convs:
        c.in_channels = (c.in_channels, c.in_channels) if isinstance(c.in_channels, int) else c.in_channels
    order = torch.randperm(wv, generator=torch.Generator(device=dev).manual_seed(1), device=dev)
    out = bench.loader_api_variants(row_ptr, col, table, convs, order, n_groups, G, which=which)
    print({k: {a: b for a, b in v.items() if a != "note"} for k, v in out.items()})


if __name__ == "__main__":
    main()
