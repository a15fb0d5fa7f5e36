import sys, torch
sys.path[:0]=["/root/repo","/root/repo/cugraph-gnn_amd"]
import wholegraph_amd as wg
dev=torch.device("cuda:0"); g=torch.Generator(device=dev).manual_seed(1)
comm=wg.create_group_communicator()
n_host, dim, k = 4_000_000, 128, 500_000
hot=(torch.empty(k,device=dev).exponential_(1.0,generator=g)*(n_host*0.01)).long().clamp_(max=n_host-1)
pol=wg.create_wholememory_cache_policy(comm,memory_type="distributed",memory_location="cuda",access_type="readwrite",ratio=0.05)
emb=wg.create_embedding(comm,"distributed","cpu",torch.float32,[n_host,dim],cache_policy=pol)
for _ in range(30): emb.gather(hot)
torch.cuda.synchronize()
import time
t0=time.perf_counter()
for _ in range(20): emb.gather(hot)
torch.cuda.synchronize(); print("wall ms", (time.perf_counter()-t0)/20*1e3)
