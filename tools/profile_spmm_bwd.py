"""Where the time of the SAGE mean SpMM backward goes (transpose, scaling, gather over the transposed hop)."""
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


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for F in (100, 256):
    gout = torch.rand((T, F), generator=g, device=dev)
    t_all = timed(lambda: nn.spmm_csr_backward(rp, col2, gout, n_src, True))
    t_tr = timed(lambda: nn.csr_transpose(rp, col2, n_src))
    deg = (rp[1:] - rp[:-1]).clamp_(min=1)
    t_div = timed(lambda: gout / deg.unsqueeze(1))
    rpt, colt = nn.csr_transpose(rp, col2, n_src)
    gs = gout / deg.unsqueeze(1)
    t_sp = timed(lambda: nn.spmm_csr_forward(rpt, colt, gs, mean=False))

    dt = (rpt[1:] - rpt[:-1])
    print(f"F={F}: backward {t_all:.3f} ms = transpose {t_tr:.3f} + scale {t_div:.3f} + gather over transposed {t_sp:.3f}"
          f"   (source rows: {n_src}, mean entries {E / n_src:.2f}, max {int(dt.max())})")
