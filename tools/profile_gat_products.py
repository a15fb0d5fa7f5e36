"""bench.py's GAT variants alone (2 x nn.GATConv over the products call groups, forward and training step) — for a kernel trace:
rocprofv3 --kernel-trace --stats -- python tools/profile_gat_products.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch
from bench import rmat_csr, V_PRODUCTS, E_UNDIRECTED, loader_api_variants
dev = torch.device("cuda", 0)
row_ptr, col = rmat_csr(V_PRODUCTS, E_UNDIRECTED, 0, dev)
g = torch.Generator(device=dev).manual_seed(1)
table = torch.randn((V_PRODUCTS, 100), generator=g, device=dev)
seeds = torch.cat([torch.randperm(V_PRODUCTS, generator=g, device=dev)] * 2)
from cugraph_pyg_amd.sampler.sampler import default_local_seeds_per_call
import bench
G = max(1, default_local_seeds_per_call(bench.FANOUT, bench.BATCH, 8) // bench.BATCH)      # the loader's own call-group size
out = loader_api_variants(row_ptr, col, table, [], seeds, 4, G, which=("gat",))
print({k: (round(v["value"] / 1e9, 3), round(v["ms_per_call_group"], 2)) for k, v in out.items()})
