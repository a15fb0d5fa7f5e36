"""Randomized parity sweep of nn._GatAggregateHeads (forward + every gradient) against float64 autograd: 80 draws over F, heads,
addressing mode (plain / id list / table-level terms on either end), degrees up to 200 (the chunked path), with and without
dst_rows.  Prints the draws whose relative error exceeds 5e-5; run on the GPU box: python tools/stress_gat_aggregate.py"""
import os, sys, random
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path[:0]=[R, R+"/cugraph-gnn_amd", R+"/tests"]
import torch
from wholegraph_amd import nn
sys.path.insert(0, R+"/tests")
from test_gpu_mag_pipeline import _agg_heads_reference
random.seed(7)
bad = 0
for trial in range(80):
    F = random.choice([4, 8, 20, 32, 36, 64, 100, 128, 132, 200, 256]); H = random.choice([1, 2, 4, 8])
    mode = random.choice(["plain", "ids", "by_id", "by_id_src", "by_id_dst"])
    maxdeg = random.choice([3, 10, 26, 70, 200])
    n_table, n_src, n_dst_list, n_rows = random.randint(50, 4000), random.randint(50, 6000), random.randint(50, 3000), random.randint(1, 1500)
    n_rows = min(n_rows, n_dst_list)
    g = torch.Generator(device="cuda").manual_seed(trial)
    deg = torch.randint(0, maxdeg + 1, (n_rows,), generator=g, device="cuda")
    if trial % 5 == 0: deg[::3] = 0
    rp = torch.zeros(n_rows + 1, dtype=torch.int32, device="cuda"); rp[1:] = torch.cumsum(deg, 0)
    E = int(rp[-1])
    if E == 0: continue
    col = torch.randint(0, n_src, (E,), generator=g, device="cuda", dtype=torch.int32)
    dst_rows = torch.randperm(n_dst_list, generator=g, device="cuda")[:n_rows].contiguous() if trial % 3 else None
    if dst_rows is None: n_dst_list = n_rows
    gout = torch.randn((n_rows, H * F), generator=g, device="cuda")
    lazy = mode != "plain"
    table = torch.randn((n_table if lazy else n_src, F), generator=g, device="cuda")
    ids = torch.randint(0, n_table, (n_src,), generator=g, device="cuda") if lazy else None
    dids = torch.randint(0, n_table, (n_dst_list,), generator=g, device="cuda") if lazy else None
    sbi, dbi = mode in ("by_id", "by_id_src"), mode in ("by_id", "by_id_dst")
    a_src = (torch.randn((n_table if sbi else n_src, H), generator=g, device="cuda") * 2).requires_grad_(True)
    a_dst = (torch.randn((n_table if dbi else n_dst_list, H), generator=g, device="cuda") * 2).requires_grad_(True)
    xg = (not lazy) or mode == "ids"
    x = table.clone().requires_grad_(xg)
    out = nn._GatAggregateHeads.apply(x, a_src, a_dst, rp, col, H, dst_rows, ids, dids if dbi else None, sbi, dbi, 0.2)
    out.backward(gout)
    x64 = table.double().requires_grad_(xg)
    s64, d64 = a_src.detach().double().requires_grad_(True), a_dst.detach().double().requires_grad_(True)
    xl = x64[ids] if lazy else x64
    sl = s64[ids] if sbi else s64
    dl = d64[dids] if dbi else d64
    dr = dst_rows if dst_rows is not None else torch.arange(n_rows, device="cuda")
    ref = _agg_heads_reference(xl, sl, dl, rp, col, H, dr)
    ref.backward(gout.double())
    errs = [float((out.double() - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)]
    for got, want in ((a_src.grad, s64.grad), (a_dst.grad, d64.grad)) + (((x.grad, x64.grad),) if xg else ()):
        errs.append(float((got.double() - want).abs().max()) / max(float(want.abs().max()), 1e-30))
    if max(errs) > 5e-5:
        bad += 1
        print("BAD", trial, F, H, mode, maxdeg, n_rows, [round(e, 7) for e in errs])
print("stress done, bad =", bad)
