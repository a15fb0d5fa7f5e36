"""Stand-alone timing of the aggregate-first GAT aggregation, forward and backward, at the deep-hop shape of the mag call group
(440 k destination rows, ~8 sampled neighbours, F = 128, H = 4, rows read through an id list from a 1.1 M-row table)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch
from wholegraph_amd import nn

g = torch.Generator(device="cuda").manual_seed(0)
n_table, n_src, n_rows, F, H = 1_134_649, 3_800_000, 440_000, 128, 4
table = torch.randn((n_table, F), generator=g, device="cuda")
ids = (torch.rand(n_src, generator=g, device="cuda") ** 2 * n_table).long().clamp_(max=n_table - 1)
deg = torch.randint(4, 11, (n_rows,), generator=g, device="cuda")
rp = torch.zeros(n_rows + 1, dtype=torch.int32, device="cuda"); rp[1:] = torch.cumsum(deg, 0)
E = int(rp[-1])
col = torch.randint(0, n_src, (E,), generator=g, device="cuda", dtype=torch.int32)
dst_rows = torch.randperm(n_src, generator=g, device="cuda")[:n_rows].contiguous()
gout = torch.randn((n_rows, H * F), generator=g, device="cuda")


def timed(fn, it=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it


for by_id in (True, False):
    a_src = (torch.randn((n_table if by_id else n_src, H), generator=g, device="cuda")).requires_grad_(True)
    a_dst = (torch.randn((n_table if by_id else n_src, H), generator=g, device="cuda")).requires_grad_(True)
    out = nn._GatAggregateHeads.apply(table, a_src, a_dst, rp, col, H, dst_rows, ids, ids if by_id else None, by_id, by_id, 0.2)
    t_f = timed(lambda: nn.gat_aggregate_heads(rp, col, table, a_src.detach(), a_dst.detach(), H, dst_rows=dst_rows, src_ids=ids,
                                               dst_ids=ids if by_id else None, src_terms_by_id=by_id, dst_terms_by_id=by_id))
    t_b = timed(lambda: torch.autograd.grad(out, (a_src, a_dst), gout, retain_graph=True))
    byt = E * (F * 4 + 8 + 4 + 16) + n_rows * (H * F * 4 + 16 + 8)
    print("terms by id %s: edges %d  forward %.3f ms (%.2f TB/s)  backward %.3f ms (%.2f TB/s)" % (
        by_id, E, t_f, byt / t_f / 1e9, t_b, byt / t_b / 1e9))
