"""GAT backward (wgamd_gat_csr_bwd_f32) at the products hop-2 shape: a power-law hop seen from the sources."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch
from bench import rmat_csr, V_PRODUCTS, E_UNDIRECTED
from wholegraph_amd import wholegraph_ops, graph_ops, nn

dev = torch.device("cuda", 0)
row_ptr, col = rmat_csr(V_PRODUCTS, E_UNDIRECTED, 0, dev)
g = torch.Generator(device=dev).manual_seed(3)
seeds = torch.randperm(V_PRODUCTS, generator=g, device=dev)[:64 * 1024]
hop1 = wholegraph_ops.unweighted_sample_without_replacement(row_ptr, col, seeds, 25, random_seed=1)
frontier = torch.unique(hop1[1])
hop2 = wholegraph_ops.unweighted_sample_without_replacement(row_ptr, col, frontier, 10, random_seed=2)
u2, col2 = graph_ops.append_unique(frontier, hop2[1], need_neighbor_raw_to_unique=True)[:2]
rp = hop2[0]
T, E, n_src = frontier.numel(), col2.numel(), u2.numel()
print("T", T, "E", E, "n_src", n_src)


def timed(fn, iters=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for H, C in ((4, 64), (1, 256)):
    x = torch.rand((n_src, H * C), generator=g, device=dev)
    a_src = torch.rand((n_src, H), generator=g, device=dev)
    a_dst = torch.rand((T, H), generator=g, device=dev)
    out, alpha = nn.gat_forward(rp, col2, x, a_src, a_dst, H)
    gout = torch.rand_like(out)
    t_f = timed(lambda: nn.gat_forward(rp, col2, x, a_src, a_dst, H))
    t_b = timed(lambda: nn.gat_backward(rp, col2, x, a_src, a_dst, alpha, gout, H))
    print(f"H={H} C={C}: forward {t_f:.3f} ms, backward {t_b:.3f} ms")
