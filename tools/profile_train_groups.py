"""The products workload's TRAINING loop alone (bench.py's `train_step` variant: NeighborLoader.call_groups() -> 2 x nn.SAGEConv
-> loss -> backward -> SGD), for rocprofv3 --kernel-trace --stats / --pmc passes of the backward kernels.
usage: python tools/profile_train_groups.py [n_groups]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from wholegraph_amd import nn  # noqa: E402
from cugraph_pyg_amd.sampler.sampler import default_local_seeds_per_call  # noqa: E402

n_groups = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0)
V, E_u, F, C, fan = bench.WORKLOADS["products"]
row_ptr, col = bench.rmat_csr(V, E_u, seed=0, device=dev)
table = torch.rand((V, F), generator=torch.Generator(device=dev).manual_seed(100), device=dev) * 2 - 1
G = max(1, default_local_seeds_per_call(fan, bench.BATCH, 8) // bench.BATCH)
g = torch.Generator(device=dev).manual_seed(1)
dims = [F, bench.HIDDEN, C]
convs = [nn.SAGEConv(dims[j], dims[j + 1]).to(dev) for j in range(2)]
for c in convs:
    for p in c.parameters():
        p.data = (torch.rand(p.shape, generator=g, device=dev) - 0.5) * 0.1
order = torch.cat([torch.randperm(V, generator=torch.Generator(device=dev).manual_seed(7), device=dev) for _ in range(3)])
out = bench.loader_api_variants(row_ptr, col, table, convs, order, n_groups, G, which=("loader_api", "train_step"))
print(json.dumps(out))
