"""Wall time of the partitioned feature fetch at world size 1 (self all-to-all through RCCL): torch.distributed pipeline
(wholegraph_amd/dist.py) vs the in-library pipeline behind wholememory_gather (csrc/wg_comm.hip) vs the plain local gather."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
import torch, torch.distributed as dist
import wholegraph_amd as wg
from wholegraph_amd.tensor import local_gather
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
V, F, n = 2_449_029, 100, 3_650_000
table = torch.rand((V, F), device=dev)
idx = torch.randint(0, V, (n,), device=dev)
def wall(fn, it=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3
out = torch.empty((n, F), device=dev)
print("local gather        %.2f ms" % wall(lambda: local_gather(table, idx, out)))
wm = wg.WholeMemoryTensor(table, global_rows=V, partition_offsets=[0, V])
print("torch.distributed   %.2f ms" % wall(lambda: wm.gather(idx)))
comm = wg.create_group_communicator()
t = wg.create_wholememory_tensor(comm, "distributed", "cuda", [V, F], torch.float32, [F, 1])
t.get_local_tensor()[0].copy_(table)
print("in-library (C ABI)  %.2f ms" % wall(lambda: t.gather(idx)))
assert torch.equal(t.gather(idx), table[idx])
dist.destroy_process_group()
