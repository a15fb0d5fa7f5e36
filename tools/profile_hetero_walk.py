"""Where the time of the heterogeneous call-group sampler goes (ogbn-mag-like graph of bench_ops.py --hetero):
host enqueue of HeteroPygWalk.run, device time of the same, finalize_batches (per-batch tuples)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch
from cugraph_pyg_amd.data import GraphStore
from cugraph_pyg_amd.sampler.sampler import HeteroNeighborSampler
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(11)
n = {"paper": 736_389, "author": 1_134_649, "institution": 8_740, "field_of_study": 59_965}
import bench_mag
rel = bench_mag.MAG_RELS_R5 if os.environ.get("RELS", "all") == "r5" else bench_mag.MAG_RELS     # all 8 directed edge types by default
bench_mag.build_mag_like(dev, n, rel)
gs = bench_mag.build_mag_like.graph_store
B, G = 1024, int(os.environ.get("G", 32))
smp = HeteroNeighborSampler(gs._hetero_graphs, {et: [25, 10] for et in rel}, local_seeds_per_call=B * G, num_nodes=n)
walk = smp._call_group_walk(B, G)
seeds = torch.randperm(n["paper"], generator=g, device=dev)[:B * G].contiguous()
rs = torch.arange(2 * len(rel) * G, device=dev, dtype=torch.int64).view(2 * len(rel), G) + 7
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rec = walk.run("paper", seeds, rs)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    outs = list(walk.finalize_batches(rec))
    torch.cuda.synchronize(); t3 = time.perf_counter()
    e = sum(sum(sum(v) for v in o[5].values()) for o in outs)
    print("G=%d  run enqueue %.2f ms  + device tail %.2f ms  finalize %.2f ms  -> %.3f ms/batch, %.2f G edges/s" % (
        G, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3 / G, e / (t3 - t0) / 1e9), flush=True)
