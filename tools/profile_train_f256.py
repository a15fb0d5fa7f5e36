import json, os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch, bench
from wholegraph_amd import nn
dev = torch.device("cuda", 0)
# a smaller graph with the RMAT-26 model shape: F = 256, 3 layers 256-256-256-C, fan-out [15,10,5]
V, E_u = 4_000_000, 60_000_000
F, C = 256, 64
bench.FANOUT[:] = [15, 10, 5]
row_ptr, col = bench.rmat_csr(V, E_u, seed=0, device=dev)
table = torch.rand((V, F), generator=torch.Generator(device=dev).manual_seed(100), device=dev) * 2 - 1
g = torch.Generator(device=dev).manual_seed(1)
dims = [F, 256, 256, C]
convs = [nn.SAGEConv(dims[j], dims[j + 1]).to(dev) for j in range(3)]
order = torch.cat([torch.randperm(V, generator=torch.Generator(device=dev).manual_seed(7), device=dev) for _ in range(3)])
bench.CLASSES = C
out = bench.loader_api_variants(row_ptr, col, table, convs, order, 6, 66, which=("loader_api", "train_step"))
print(json.dumps({k: {a: b for a, b in v.items() if a not in ("note", "wgrad_roofline")} for k, v in out.items()}))
