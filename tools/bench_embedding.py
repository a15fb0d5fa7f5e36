"""Time the sparse optimizer step (wholememory_embedding_gather_gradient_apply) on one GPU; run under rocprofv3 for the
per-kernel split (route copies / radix sort / fused sum+update)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch
import wholegraph_amd as wg
dev = torch.device("cuda", 0)
comm = wg.create_group_communicator()
n_rows, k, dim = 1_000_000, 1_000_000, 128
for kind in ("sgd", "lazy_adam", "adagrad"):
    for dtype in (torch.float32, torch.bfloat16):
        emb = wg.create_embedding(comm, "distributed", "cuda", dtype, [n_rows, dim], random_init=True)
        opt = wg.create_wholememory_optimizer(emb, kind, {})
        idx = torch.randint(0, n_rows, (k,), device=dev)
        grads = torch.rand((k, dim), device=dev)
        def step():
            emb.add_gradients(idx, grads); emb.apply_gradients(0.01)
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print("%-10s %-8s %.3f ms/step (%d pairs, dim %d)" % (kind, str(dtype).split(".")[1], dt * 1e3, k, dim), flush=True)
        wg.destroy_embedding(emb); wg.destroy_wholememory_optimizer(opt)
