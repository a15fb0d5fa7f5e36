import os, sys, time, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch
import bench
from wholegraph_amd import WholeMemoryTensor
dev = torch.device("cuda", 0)
G = int(sys.argv[1])
row_ptr, col = bench.rmat_csr(bench.V_PRODUCTS, bench.E_UNDIRECTED, 0, dev)
feat = WholeMemoryTensor(torch.rand((bench.V_PRODUCTS, 100), device=dev))
pipe = bench.SagePipeline(row_ptr, col, feat, dev, G)
order = torch.randperm(bench.V_PRODUCTS, device=dev)
n = G * 1024
def grp(i): return order[(i * n) % (bench.V_PRODUCTS - n):][:n].contiguous()
pend = pipe.sample(grp(0), 0)
for g in range(12):
    t0 = time.perf_counter()
    nxt = pipe.sample(grp(g + 1), g + 1)
    t1 = time.perf_counter()
    pipe.forward(*pend, mode="fused")
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    st = torch.cuda.memory_stats()
    print(f"g={g} sample_enq {1e3*(t1-t0):.2f} fwd_enq {1e3*(t2-t1):.2f} sync {1e3*(t3-t2):.2f} ms  reserved {st['reserved_bytes.all.current']/2**30:.1f} GiB "
          f"retries {st['num_alloc_retries']} segs {st['segment.all.allocated']}")
    pend = nxt
