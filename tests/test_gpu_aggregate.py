"""GPU parity: SAGE mean/sum SpMM and GAT edge-softmax aggregation vs the fp64 oracle (1e-5 rel,
the north-star tolerance), plus layer-level checks against plain torch fp32 formulas."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-5, 1e-6


def _csr(n_dst, n_src, max_deg, seed):
    rng = np.random.default_rng(seed)
    deg = rng.integers(0, max_deg + 1, n_dst)
    deg[:3] = [0, 1, max_deg]
    rp = np.zeros(n_dst + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    col = rng.integers(0, n_src, rp[-1]).astype(np.int32)
    return rp, col


@pytest.mark.parametrize("F", [1, 3, 16, 100, 128, 256, 300, 602])
@pytest.mark.parametrize("mean", [True, False])
def test_spmm_vs_oracle(oracle_mod, hiplib, F, mean):
    import torch
    from wholegraph_amd import nn
    rp, col = _csr(3001, 9000, 70, F)
    x = np.random.default_rng(F).standard_normal((9000, F)).astype(np.float32)
    out = nn.spmm_csr_forward(torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda(), torch.from_numpy(x).cuda(), mean)
    ref64 = oracle_mod.spmm_csr(rp, col, x, mean=mean, acc_double=True)
    np.testing.assert_allclose(out.cpu().numpy(), ref64, rtol=RTOL, atol=ATOL * 70)
    # sums run in CSR order -> bit-identical to the sequential fp32 loop
    ref32 = oracle_mod.spmm_csr(rp, col, x, mean=mean, acc_double=False)
    assert np.array_equal(out.cpu().numpy(), ref32)


def test_spmm_fused_feature_fetch(oracle_mod, hiplib):
    import torch
    from wholegraph_amd import nn
    rp, col = _csr(2000, 5000, 25, 1)
    table = np.random.default_rng(2).standard_normal((100000, 100)).astype(np.float32)
    gids = np.random.default_rng(3).permutation(100000)[:5000].astype(np.int64)
    out = nn.spmm_csr_forward(torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda(), torch.from_numpy(table).cuda(),
                              True, src_ids=torch.from_numpy(gids).cuda())
    ref = oracle_mod.spmm_csr(rp, col, table[gids], mean=True, acc_double=False)
    assert np.array_equal(out.cpu().numpy(), ref)


def test_spmm_backward_matches_torch(hiplib):
    import torch
    from wholegraph_amd import nn
    rp, col = _csr(500, 800, 12, 4)
    x = torch.randn(800, 64, device="cuda", requires_grad=True)
    rpt, ct = torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda()
    out = nn.spmm_csr(x, rpt, ct, "mean")
    gout = torch.randn_like(out)
    out.backward(gout)
    deg = torch.from_numpy(np.diff(rp)).cuda()
    dst = torch.repeat_interleave(torch.arange(500, device="cuda"), deg.long())
    x2 = x.detach().clone().requires_grad_(True)
    ref = torch.zeros(500, 64, device="cuda").index_add_(0, dst, x2[ct.long()]) / deg.clamp(min=1).view(-1, 1)
    ref.backward(gout)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(x.grad, x2.grad, rtol=1e-4, atol=1e-5)
    # both backward flavours (transposed gather = default, atomic scatter-add) and the sum reduction; the transposed
    # one is deterministic: two runs are bit-identical
    for mean in (True, False):
        a = nn.spmm_csr_backward(rpt, ct, gout, 800, mean)
        b = nn.spmm_csr_backward(rpt, ct, gout, 800, mean, atomic=True)
        assert torch.equal(a, nn.spmm_csr_backward(rpt, ct, gout, 800, mean))
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
        if mean:
            torch.testing.assert_close(a, x2.grad, rtol=1e-4, atol=1e-5)
    rpt_t, ct_t = nn.csr_transpose(rpt, ct, 800)
    assert rpt_t[-1].item() == ct.shape[0] and torch.equal(torch.diff(rpt_t).long(), torch.bincount(ct, minlength=800))


def test_spmm_backward_with_hub_sources(hiplib):
    """A power-law hop seen from the sources: a handful of them are neighbours of thousands of rows.  The backward gathers
    over the transposed hop in segments of at most 64 entries (then adds the segments of a row up in order): same values as
    the scatter-add and as torch's index_add_, bit-identical from run to run, empty sources included."""
    import torch
    from wholegraph_amd import nn
    g = torch.Generator(device="cuda").manual_seed(5)
    n_dst, n_src, F = 6000, 3000, 100
    deg = torch.randint(0, 11, (n_dst,), generator=g, device="cuda")
    rp = torch.zeros(n_dst + 1, dtype=torch.int32, device="cuda")
    rp[1:] = torch.cumsum(deg, 0)
    E = int(rp[-1])
    col = torch.randint(0, n_src - 500, (E,), generator=g, device="cuda", dtype=torch.int32)   # the last 500 sources: unused
    hub = torch.rand(E, generator=g, device="cuda")
    col[hub < 0.25] = 7            # ~ E / 4 entries on one source
    col[(hub >= 0.25) & (hub < 0.30)] = 1234
    gout = torch.randn((n_dst, F), generator=g, device="cuda")
    for mean in (True, False):
        a = nn.spmm_csr_backward(rp, col, gout, n_src, mean)
        assert torch.equal(a, nn.spmm_csr_backward(rp, col, gout, n_src, mean))
        b = nn.spmm_csr_backward(rp, col, gout, n_src, mean, atomic=True)
        scale = gout / deg.clamp(min=1).unsqueeze(1) if mean else gout
        dst = torch.repeat_interleave(torch.arange(n_dst, device="cuda"), deg.long())
        ref = torch.zeros((n_src, F), dtype=torch.float64, device="cuda").index_add_(0, col.long(), scale[dst].double())
        torch.testing.assert_close(a.double(), ref, rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-3)
        assert torch.count_nonzero(a[n_src - 500:]) == 0


@pytest.mark.parametrize("H,C", [(1, 8), (4, 32), (4, 16), (2, 5), (8, 64)])
def test_gat_vs_oracle(oracle_mod, hiplib, H, C):
    import torch
    from wholegraph_amd import nn
    rp, col = _csr(1500, 4000, 30, H * C)
    rng = np.random.default_rng(H)
    x = rng.standard_normal((4000, H * C)).astype(np.float32)
    a_src = rng.standard_normal((4000, H)).astype(np.float32)
    a_dst = rng.standard_normal((1500, H)).astype(np.float32)
    out, alpha = nn.gat_forward(torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda(), torch.from_numpy(x).cuda(),
                                torch.from_numpy(a_src).cuda(), torch.from_numpy(a_dst).cuda(), H, 0.2)
    oref, aref = oracle_mod.gat_csr(rp, col, x.reshape(-1, H, C), a_src, a_dst, 0.2)
    np.testing.assert_allclose(alpha.cpu().numpy(), aref, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(out.cpu().numpy().reshape(-1, H, C), oref, rtol=RTOL, atol=1e-5)
    # per-destination attention sums to 1 wherever there are edges
    sums = np.add.reduceat(alpha.cpu().numpy(), rp[:-1][np.diff(rp) > 0], axis=0)
    np.testing.assert_allclose(sums, 1.0, rtol=1e-5)


def test_sage_and_gat_layers_train_step(hiplib):
    import torch
    from wholegraph_amd import nn
    rp, col = _csr(300, 900, 10, 8)
    rpt, ct = torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda()
    x = torch.randn(900, 48, device="cuda")
    sage = nn.SAGEConv(48, 32).cuda()
    y = sage((x, x[:300]), [rpt, ct])
    deg = torch.from_numpy(np.diff(rp)).cuda()
    dst = torch.repeat_interleave(torch.arange(300, device="cuda"), deg.long())
    agg = torch.zeros(300, 48, device="cuda").index_add_(0, dst, x[ct.long()]) / deg.clamp(min=1).view(-1, 1)
    ref = sage.lin_l(agg) + sage.lin_r(x[:300])
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-4)
    gat = nn.GATConv(48, 16, heads=4).cuda()
    z = gat((x, x[:300]), [rpt, ct])
    assert z.shape == (300, 64)
    (y.sum() + z.sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in list(sage.parameters()) + list(gat.parameters()))


@pytest.mark.parametrize("F", [100, 256, 7])
def test_sage_aggregate_concat_self(oracle_mod, hiplib, F):
    import torch
    from wholegraph_amd import nn
    rp, col = _csr(2000, 6000, 25, F + 1)
    x = np.random.default_rng(F).standard_normal((6000, F)).astype(np.float32)
    self_rows = np.random.default_rng(1).integers(0, 6000, 2000).astype(np.int64)
    out = nn.sage_aggregate_forward(torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda(), torch.from_numpy(x).cuda(),
                                    torch.from_numpy(self_rows).cuda(), True).cpu().numpy()
    assert out.shape == (2000, 2 * F)
    assert np.array_equal(out[:, :F], oracle_mod.spmm_csr(rp, col, x, mean=True, acc_double=False))
    assert np.array_equal(out[:, F:], x[self_rows])


def test_sage_aggregate_fused_feature_fetch(oracle_mod, hiplib):
    import torch
    from wholegraph_amd import nn
    rp, col = _csr(1500, 4000, 25, 3)
    table = np.random.default_rng(0).standard_normal((50000, 100)).astype(np.float32)
    n_id = np.random.default_rng(1).permutation(50000)[:4000].astype(np.int64)
    self_rows = np.random.default_rng(2).integers(0, 4000, 1500).astype(np.int64)
    out = nn.sage_aggregate_fetch_forward(torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda(),
                                          torch.from_numpy(table).cuda(), torch.from_numpy(n_id).cuda(),
                                          torch.from_numpy(self_rows).cuda(), True).cpu().numpy()
    x = table[n_id]
    assert np.array_equal(out[:, :100], oracle_mod.spmm_csr(rp, col, x, mean=True, acc_double=False))
    assert np.array_equal(out[:, 100:], x[self_rows])
    # identical to the unfused pair (gather, then aggregate)
    ref = nn.sage_aggregate_forward(torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda(), torch.from_numpy(x).cuda(),
                                    torch.from_numpy(self_rows).cuda(), True).cpu().numpy()
    assert np.array_equal(out, ref)


# (F > 148 with F % 16 == 0: the bf16x3 kernel runs 64-row tiles in two halves; 152 and 204: its 32-row tiles)
@pytest.mark.parametrize("F,N", [(100, 256), (128, 256), (64, 64), (256, 128), (4, 128), (36, 256), (208, 128), (104, 64),
                                 (256, 256), (256, 64), (176, 256), (160, 64), (152, 128), (204, 256),
                                 (256, 47), (100, 172), (128, 1),    # widths padded to 64 / 256 / 64 on the way in
                                 (140, 128)])
@pytest.mark.parametrize("with_ids", [False, True])
@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_sage_layer_fused_matches_aggregate_plus_gemm(oracle_mod, hiplib, F, N, with_ids, precision):
    """The one-kernel SAGE layer (gather -> aggregate in LDS -> MFMA transform) against the oracle's sequential fp32 SpMM +
    an fp64 matmul, and against the two-kernel product path (aggregate kernel + library GEMM).  precision = "f32":
    wgamd_sage_layer_fused_f32 (exact fp32 MFMA); "bf16x3": wgamd_sage_layer_fused_bf16x3 (3-way bf16 split of both
    operands, six bf16 MFMA products per fp32 product, fp32 accumulation) — SAME tolerance, 1e-5 x scale (north_star)."""
    import torch
    from wholegraph_amd import nn
    n_dst, n_src, V = 1000 + F, 2500, 40000     # n_dst not a multiple of the 64-row tile
    rp, col = _csr(n_dst, n_src, 14, F)
    rng = np.random.default_rng(F + N)
    table = rng.standard_normal((V if with_ids else n_src, F)).astype(np.float32)
    ids = rng.permutation(V)[:n_src].astype(np.int64) if with_ids else None
    x_local = table[ids] if with_ids else table
    self_rows = rng.integers(0, n_src, n_dst).astype(np.int64)
    w_t = (rng.standard_normal((2 * F, N)) * 0.2).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    cu = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    for relu in (True, False):
        got = nn.sage_layer_fused_forward(cu(rp), cu(col), cu(table), cu(self_rows), cu(w_t), cu(bias), relu=relu, mean=True,
                                          src_ids=cu(ids) if with_ids else None, precision=precision).cpu().numpy()
        agg = oracle_mod.spmm_csr(rp, col, x_local, mean=True, acc_double=False)
        cat = np.concatenate([agg, x_local[self_rows]], axis=1)
        ref = cat.astype(np.float64) @ w_t.astype(np.float64) + bias
        if relu:
            ref = np.maximum(ref, 0)
        scale = np.abs(cat).astype(np.float64) @ np.abs(w_t).astype(np.float64) + np.abs(bias)
        # (a) the bound of a dot product: |err| <= 1e-5 * sum |a||b| — what fp32 accumulation itself can promise
        assert np.all(np.abs(got - ref) <= 1e-5 * scale + 1e-6), np.abs(got - ref).max()
        # (b) north_star's "1e-5 rel" taken literally, element by element, wherever the result is not a cancellation
        #     (|ref| >= 0.1 * scale: most entries of a bias-dominated / ReLU'd output)
        big = np.abs(ref) >= 0.1 * scale
        assert big.any() or N == 1      # (a single output column after ReLU may hold no such entry)
        if big.any():
            rel = np.abs(got - ref)[big] / np.abs(ref)[big]
            assert rel.max() <= 1e-5, rel.max()
    # the two-kernel path computes the same layer
    cat_g = nn.sage_aggregate_forward(cu(rp), cu(col), cu(x_local), cu(self_rows), True)
    two = torch.addmm(cu(bias), cat_g, cu(w_t)).cpu().numpy()
    np.testing.assert_allclose(got, two, rtol=2e-5, atol=2e-5)
    assert nn.sage_layer_fused_supported(F, N)


def test_sage_layer_fused_rejects_unsupported_shapes(hiplib):
    import torch
    import wholegraph_amd as wg
    from wholegraph_amd import nn
    rp = torch.tensor([0, 1], dtype=torch.int32, device="cuda")
    col = torch.zeros(1, dtype=torch.int32, device="cuda")
    x = torch.zeros((1, 100), device="cuda")
    with pytest.raises(wg.WholeMemoryError):       # more than 256 output columns
        nn.sage_layer_fused_forward(rp, col, x, torch.zeros(1, dtype=torch.int64, device="cuda"), torch.zeros((200, 320), device="cuda"))
    assert not nn.sage_layer_fused_supported(100, 320) and not nn.sage_layer_fused_supported(102, 256)
    assert nn.sage_layer_fused_supported(100, 47)  # a 47-class head runs as 64 zero-padded columns


def test_sage_layer_fused_64bit_offset_path_and_tiny_inputs(hiplib):
    """x_rows = 0 (unknown extent) selects the 64-bit row-offset code path; it must agree bit-for-bit with the 32-bit one.
    Also: fewer rows than one 64-row tile, a single row, int32 id indirection."""
    import torch
    from wholegraph_amd import _lib as L
    from wholegraph_amd import nn
    from wholegraph_amd.env import get_stream
    g = torch.Generator(device="cuda").manual_seed(0)
    for n_dst in (1, 37, 64, 65, 300):
        F, N, n_src = 100, 256, 500
        deg = torch.randint(0, 14, (n_dst,), generator=g, device="cuda")
        rp = torch.zeros(n_dst + 1, dtype=torch.int32, device="cuda")
        rp[1:] = torch.cumsum(deg, 0)
        col = torch.randint(0, n_src, (int(rp[-1]),), generator=g, device="cuda", dtype=torch.int32)
        table = torch.randn((3000, F), generator=g, device="cuda")
        ids = torch.randperm(3000, generator=g, device="cuda")[:n_src].int()
        rows = torch.randint(0, n_src, (n_dst,), generator=g, device="cuda")
        w_t = torch.randn((2 * F, N), generator=g, device="cuda") * 0.1
        bias = torch.randn(N, generator=g, device="cuda")
        a = nn.sage_layer_fused_forward(rp, col, table, rows, w_t, bias, relu=True, src_ids=ids, precision="f32")
        b = torch.empty_like(a)
        L.check(L.lib().wgamd_sage_layer_fused_f32(rp.data_ptr(), col.data_ptr() if col.numel() else rp.data_ptr(), n_dst,
                                                   table.data_ptr(), table.stride(0), 0, F, ids.data_ptr(), L.DT_INT,
                                                   rows.data_ptr(), 1, w_t.data_ptr(), w_t.stride(0), N, bias.data_ptr(), 1,
                                                   b.data_ptr(), b.stride(0), get_stream()), "fused")
        assert torch.equal(a, b)
        # the bf16x3 kernel: 32-bit vs 64-bit row offsets bit-for-bit, and within 1e-5-class distance of the fp32 kernel
        a3 = nn.sage_layer_fused_forward(rp, col, table, rows, w_t, bias, relu=True, src_ids=ids, precision="bf16x3")
        b3 = torch.empty_like(a3)
        L.check(L.lib().wgamd_sage_layer_fused_bf16x3(rp.data_ptr(), col.data_ptr() if col.numel() else rp.data_ptr(), n_dst,
                                                      table.data_ptr(), table.stride(0), 0, F, ids.data_ptr(), L.DT_INT,
                                                      rows.data_ptr(), 1, nn.sage_weight_planes(w_t).data_ptr(), N,
                                                      bias.data_ptr(), 1, b3.data_ptr(), b3.stride(0), get_stream()), "bf16x3")
        assert torch.equal(a3, b3)
        torch.testing.assert_close(a3, a, rtol=2e-5, atol=2e-5)
        x = table[ids.long()]
        cat = nn.sage_aggregate_forward(rp, col if col.numel() else torch.zeros(1, dtype=torch.int32, device="cuda")[:0].contiguous(),
                                        x, rows, True) if col.numel() else torch.cat([torch.zeros((n_dst, F), device="cuda"), x[rows]], 1)
        ref = torch.relu(torch.addmm(bias, cat, w_t))
        torch.testing.assert_close(a, ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("hubs", [False, True])
@pytest.mark.parametrize("H,C", [(4, 64), (1, 256), (4, 16), (8, 8), (2, 4), (1, 32)])
def test_gat_backward_kernels_match_autograd_of_dense_formula(hiplib, H, C, hubs):
    """wgamd_gat_csr_bwd_f32 against torch autograd on the plain edge-wise formulation of GAT attention; ``hubs``: a
    quarter of all edges leave ONE source (and 5 % another), so the source-major pass sums those rows in pieces."""
    import torch
    from wholegraph_amd import nn
    rp, col = _csr(700, 1500, 18, H * C)
    if hubs:
        r = np.random.default_rng(H + C).random(col.size)
        col = col.copy()
        col[r < 0.25] = 3
        col[(r >= 0.25) & (r < 0.30)] = 777
    rpt, ct = torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda()
    g = torch.Generator(device="cuda").manual_seed(H * 100 + C)
    x = torch.randn((1500, H * C), generator=g, device="cuda", requires_grad=True)
    a_s = torch.randn((1500, H), generator=g, device="cuda", requires_grad=True)
    a_d = torch.randn((700, H), generator=g, device="cuda", requires_grad=True)
    gout = torch.randn((700, H * C), generator=g, device="cuda")
    assert nn.gat_backward_supported(H, C)
    out = nn._GatCsr.apply(x, a_s, a_d, rpt, ct, H, 0.2)
    out.backward(gout)
    got = (x.grad.clone(), a_s.grad.clone(), a_d.grad.clone())
    # reference: edge-wise torch ops with autograd
    x2, s2, d2 = (t.detach().clone().requires_grad_(True) for t in (x, a_s, a_d))
    deg = torch.from_numpy(np.diff(rp)).cuda().long()
    dst = torch.repeat_interleave(torch.arange(700, device="cuda"), deg)
    src = ct.long()
    e = torch.nn.functional.leaky_relu(s2[src] + d2[dst], 0.2)                      # [E, H]
    m = torch.full((700, H), -1e30, device="cuda").scatter_reduce(0, dst.view(-1, 1).expand(-1, H), e, "amax")
    p = torch.exp(e - m[dst])
    den = torch.zeros((700, H), device="cuda").index_add_(0, dst, p)
    al = p / den[dst]
    ref = torch.zeros((700, H, C), device="cuda").index_add_(0, dst, al.unsqueeze(-1) * x2.view(-1, H, C)[src]).view(700, H * C)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-5)
    ref.backward(gout)
    for a, b in zip(got, (x2.grad, s2.grad, d2.grad)):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5)
    assert not nn.gat_backward_supported(2, 5)       # falls back to the torch-op backward (covered by the layer test)


@pytest.mark.parametrize("n_dst,n_src,max_deg", [(1000, 5000, 12), (1, 1, 1), (257, 3, 40), (5000, 100000, 3), (64, 10, 0),
                                                  (2000, 60, 10), (1200, 9800, 18), (3000, 700, 22), (20000, 9000, 12)])
def test_csr_transpose_matches_stable_sort(hiplib, n_dst, n_src, max_deg):
    """wgamd_csr_transpose_i32 (one radix sort over the bits a source row needs) against the torch formulation it replaces:
    stable sort of the sources, bincount + cumsum, repeat_interleave.  Sources without edges, rows without edges, no edges."""
    import torch
    from wholegraph_amd import nn
    g = torch.Generator().manual_seed(n_dst * 7 + n_src)
    deg = torch.randint(0, max_deg + 1, (n_dst,), generator=g)
    row_ptr = torch.zeros(n_dst + 1, dtype=torch.int32)
    row_ptr[1:] = torch.cumsum(deg, 0)
    E = int(row_ptr[-1])
    col = torch.randint(0, n_src, (E,), generator=g).int()
    rp, cc = row_ptr.cuda(), col.cuda()
    row_ptr_t, perm, dst, col_t = nn._csr_transpose(rp, cc, n_src, want_perm=True, want_dst=True, want_col_t=True)
    want_perm = torch.sort(col, stable=True).indices
    want_dst = torch.repeat_interleave(torch.arange(n_dst), deg)
    want_rpt = torch.zeros(n_src + 1, dtype=torch.int64)
    want_rpt[1:] = torch.cumsum(torch.bincount(col.long(), minlength=n_src), 0)
    assert torch.equal(row_ptr_t.cpu().long(), want_rpt)
    assert torch.equal(perm.cpu().long(), want_perm)
    assert torch.equal(dst.cpu().long(), want_dst)
    assert torch.equal(col_t.cpu().long(), want_dst[want_perm])
    rpt2, ct2 = nn.csr_transpose(rp, cc, n_src)
    assert torch.equal(rpt2, row_ptr_t) and torch.equal(ct2, col_t)


def test_csr_transpose_one_launch_path_segment_tiers(hiplib):
    """The one-launch transpose of a small hop (n_src + 1.5 E <= 32 k: one workgroup, counters and permutation in LDS) orders
    every source's segment by edge id with three mechanisms — <= 8 entries (a sorting network in registers), <= 512, longer —
    and keeps a list of at most 4096 longer-than-8 segments (it cannot overflow): 1800 sources of exactly 9 edges, 1400 of 9 +
    700 of 3, and a mix of degrees 1 / 40 / 600 / 3000 that takes every tier."""
    import torch
    from wholegraph_amd import nn
    g = torch.Generator().manual_seed(11)
    cases = [torch.arange(1800).repeat(9)[torch.randperm(1800 * 9, generator=g)],
             torch.cat([torch.arange(1400).repeat(9), torch.arange(1400, 2100).repeat(3)])[torch.randperm(1400 * 9 + 700 * 3, generator=g)],
             torch.cat([torch.arange(100, 2100), torch.full((40,), 3), torch.full((600,), 7), torch.full((3000,), 9),
                        torch.full((513,), 11), torch.full((33,), 13)])[torch.randperm(2000 + 40 + 600 + 3000 + 513 + 33, generator=g)]]
    for col in cases:
        E, n_src, n_dst = col.numel(), int(col.max()) + 5, 611
        cuts = torch.sort(torch.randint(0, E + 1, (n_dst - 1,), generator=g)).values
        row_ptr = torch.cat([torch.zeros(1, dtype=torch.int64), cuts, torch.tensor([E])]).int()
        assert n_src + 1 + E + (E + 1) // 2 <= 32 * 1024
        row_ptr_t, perm, dst, col_t = nn._csr_transpose(row_ptr.cuda(), col.int().cuda(), n_src, want_perm=True, want_dst=True, want_col_t=True)
        want_perm = torch.sort(col, stable=True).indices
        want_dst = torch.repeat_interleave(torch.arange(n_dst), (row_ptr[1:] - row_ptr[:-1]).long())
        want_rpt = torch.zeros(n_src + 1, dtype=torch.int64)
        want_rpt[1:] = torch.cumsum(torch.bincount(col, minlength=n_src), 0)
        assert torch.equal(row_ptr_t.cpu().long(), want_rpt)
        assert torch.equal(perm.cpu().long(), want_perm)
        assert torch.equal(dst.cpu().long(), want_dst)
        assert torch.equal(col_t.cpu().long(), want_dst[want_perm])


@pytest.mark.parametrize("n_src,used,E", [(5_000_000, 1000, 30000), (3_000_000, 3_000_000, 2000), (200_000, 50, 5), (40, 40, 3)])
def test_csr_transpose_with_long_stretches_of_unused_sources(hiplib, n_src, used, E):
    """The sources a hop never touches — the rest of a trimmed layer's input behind the last referenced row, wide holes between
    clusters — are filled by the whole grid, not by the one thread that owns the stretch (12 ms per transpose before): row
    offsets against torch's bincount + cumsum for millions of untouched sources, tiny edge lists (no room for the gap list)."""
    import torch
    from wholegraph_amd import nn
    g = torch.Generator().manual_seed(n_src + E)
    n_dst = 777
    cuts = torch.sort(torch.randint(0, E + 1, (n_dst - 1,), generator=g)).values
    row_ptr = torch.cat([torch.zeros(1, dtype=torch.int64), cuts, torch.tensor([E])]).int()
    clusters = torch.randint(0, max(n_src - used, 0) + 1, (3,), generator=g)               # three clusters of referenced sources
    col = (clusters[torch.randint(0, 3, (E,), generator=g)] + torch.randint(0, max(used // 3, 1), (E,), generator=g)).clamp_(max=n_src - 1).int()
    row_ptr_t, perm, dst, col_t = nn._csr_transpose(row_ptr.cuda(), col.cuda(), n_src, want_perm=True, want_dst=True, want_col_t=True)
    want_rpt = torch.zeros(n_src + 1, dtype=torch.int64)
    want_rpt[1:] = torch.cumsum(torch.bincount(col.long(), minlength=n_src), 0)
    assert torch.equal(row_ptr_t.cpu().long(), want_rpt)
    assert torch.equal(perm.cpu().long(), torch.sort(col, stable=True).indices)


@pytest.mark.parametrize("n_dst,n_src,E", [(1000, 4000, 20000), (1, 5, 9), (300, 7, 0), (70000, 70000, 400000)])
def test_coo_to_csr_matches_torch_formulation(hiplib, n_dst, n_src, E):
    import torch
    from wholegraph_amd import nn
    g = torch.Generator().manual_seed(E + n_dst)
    ei = torch.stack([torch.randint(0, n_src, (E,), generator=g), torch.randint(0, n_dst, (E,), generator=g)])
    rp, cc = nn._to_csr(ei.cuda(), n_dst)
    rp_ref, cc_ref = nn._to_csr(ei, n_dst)          # CPU tensors take the torch route
    assert rp.dtype == torch.int32 and cc.dtype == torch.int32
    assert torch.equal(rp.cpu(), rp_ref) and torch.equal(cc.cpu(), cc_ref)


@pytest.mark.parametrize("F,N", [(4, 128), (36, 256), (100, 256), (128, 64), (256, 47), (256, 256)])
@pytest.mark.parametrize("mean", [True, False])
def test_sage_layer_fused_long_rows(hiplib, F, N, mean):
    """Rows far past the 10-neighbour register window (a hop with fan-out 25 makes them the rule; hubs of 90 neighbours span
    several id chunks of the narrow lane groups): the fetching waves continue the window's partial sum in CSR order, several
    row loads in flight.  Against the aggregate kernel + fp64 product, 1e-5 x scale."""
    import torch
    from wholegraph_amd import nn
    g = torch.Generator(device="cuda").manual_seed(F * 3 + N)
    n_dst, n_src = 1500 + F, 4000
    deg = torch.randint(0, 26, (n_dst,), generator=g, device="cuda")
    deg[::5] = 25
    deg[3::97] = 90
    deg[1::11] = 0
    deg[-1] = 11
    rp = torch.zeros(n_dst + 1, dtype=torch.int32, device="cuda")
    rp[1:] = torch.cumsum(deg, 0)
    col = torch.randint(0, n_src, (int(rp[-1]),), generator=g, device="cuda", dtype=torch.int32)
    x = torch.randn((n_src, F), generator=g, device="cuda")
    rows = torch.randint(0, n_src, (n_dst,), generator=g, device="cuda")
    w_t = torch.randn((2 * F, N), generator=g, device="cuda") * 0.2
    bias = torch.randn(N, generator=g, device="cuda")
    cat = nn.sage_aggregate_forward(rp, col, x, rows, mean)
    ref = cat.double() @ w_t.double() + bias.double()
    scale = cat.double().abs() @ w_t.double().abs() + bias.double().abs()
    for precision in ("bf16x3", "f32"):
        got = nn.sage_layer_fused_forward(rp, col, x, rows, w_t, bias, relu=False, mean=mean, precision=precision)
        assert got.shape == (n_dst, N)
        assert torch.all((got.double() - ref).abs() <= 1e-5 * scale + 1e-6), precision
    # the feature fetch folded in takes the same path through an id indirection
    V = 9000
    table = torch.randn((V, F), generator=g, device="cuda")
    ids = torch.randperm(V, generator=g, device="cuda")[:n_src]
    cat = nn.sage_aggregate_forward(rp, col, table[ids].contiguous(), rows, mean)
    ref = cat.double() @ w_t.double() + bias.double()
    scale = cat.double().abs() @ w_t.double().abs() + bias.double().abs()
    got = nn.sage_layer_fused_forward(rp, col, table, rows, w_t, bias, relu=False, mean=mean, src_ids=ids)
    assert torch.all((got.double() - ref).abs() <= 1e-5 * scale + 1e-6)


def test_bf16x3_split_is_exact_and_product_is_fp32_class(hiplib):
    """The weight planes of wgamd_sage_split_weight_bf16x3 sum back to the fp32 weight EXACTLY (hi + mid + lo == w, each
    piece a bf16), zero rows pad K to the 16-wide k-step; and the layer on adversarial magnitudes (1e-30 .. 1e30 mixed
    signs, long rows past the 10-neighbour window, degree-0 rows) stays within 1e-5 x scale of the fp64 product."""
    import torch
    from wholegraph_amd import nn
    g = torch.Generator(device="cuda").manual_seed(5)
    K, N = 200, 256
    w_t = (torch.randn((K, N), generator=g, device="cuda") * torch.exp(torch.randn((K, N), generator=g, device="cuda") * 8))
    # the weight as the multiplying waves read it: fp32 tiles [k-step][column][16 consecutive k], zero rows past K — the
    # 3-way bf16 split happens in registers (exact: hi + mid + lo == w, checked through the kernel's results below)
    KS = (K + 15) // 16
    tiles = nn.sage_weight_planes(w_t).view(torch.float32)[:KS * N * 16].view(KS, N, 16)
    want = torch.zeros(((K + 15) // 16) * 16, N, dtype=torch.float32, device="cuda")
    want[:K] = w_t
    assert torch.equal(tiles.permute(0, 2, 1).reshape(-1, N), want)
    F, n_dst, n_src = 100, 777, 3000
    deg = torch.randint(0, 40, (n_dst,), generator=g, device="cuda")
    deg[::7] = 0
    rp = torch.zeros(n_dst + 1, dtype=torch.int32, device="cuda")
    rp[1:] = torch.cumsum(deg, 0)
    col = torch.randint(0, n_src, (int(rp[-1]),), generator=g, device="cuda", dtype=torch.int32)
    x = torch.randn((n_src, F), generator=g, device="cuda") * torch.exp(torch.randn((n_src, F), generator=g, device="cuda") * 4)
    rows = torch.randint(0, n_src, (n_dst,), generator=g, device="cuda")
    bias = torch.randn(N, generator=g, device="cuda")
    got = nn.sage_layer_fused_forward(rp, col, x, rows, w_t, bias, relu=False, precision="bf16x3")
    cat = nn.sage_aggregate_forward(rp, col, x, rows, True)
    ref = cat.double() @ w_t.double() + bias.double()
    scale = cat.double().abs() @ w_t.double().abs() + bias.double().abs()
    assert torch.all((got.double() - ref).abs() <= 1e-5 * scale + 1e-6)
    big = ref.abs() >= 0.1 * scale                                                    # element-wise rtol where no cancellation
    assert ((got.double() - ref).abs()[big] / ref.abs()[big]).max() <= 1e-5
    # and it is fp32-class, not merely inside the bound: the error is within a few fp32 ulps of the scale
    assert ((got.double() - ref).abs() / scale).max() < 2e-6


def test_gat_backward_refuses_a_workspace_smaller_than_the_hop_needs(hiplib):
    """The plan kernel hands out piece slots with atomics: the only bound is the workspace the entry point validated
    (a scratch sized for fewer entries than the transposed hop holds must be refused, not overrun)."""
    import torch
    from wholegraph_amd import _lib as L
    E, H, C = 5000, 4, 16
    need = hiplib.wgamd_gat_csr_bwd_workspace_bytes(E, H, C)
    assert need > hiplib.wgamd_gat_csr_bwd_workspace_bytes(E // 4, H, C)
    d = torch.zeros(1 << 16, dtype=torch.float32, device="cuda")
    i = torch.zeros(1 << 14, dtype=torch.int32, device="cuda")
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    p, q = d.data_ptr(), i.data_ptr()
    rc = hiplib.wgamd_gat_csr_bwd_f32_v2(q, q, 0, p, 64, p, p, H, C, 0.2, p, p, 64, q, q, q, 10, p, p, 64, p, p, E, ws.data_ptr(),
                                         hiplib.wgamd_gat_csr_bwd_workspace_bytes(E // 4, H, C), None)
    assert rc == L.WHOLEMEMORY_INVALID_INPUT
    rc = hiplib.wgamd_gat_csr_bwd_f32_v2(q, q, 0, p, 64, p, p, H, C, 0.2, p, p, 64, q, q, q, 10, p, p, 64, p, p, 0, ws.data_ptr(),
                                         need, None)      # empty hop, full-size scratch: fine
    assert rc == L.WHOLEMEMORY_SUCCESS
    torch.cuda.synchronize()


def test_gat_backward_old_entry_point_is_bounded_by_its_workspace(hiplib):
    """`wgamd_gat_csr_bwd_f32` keeps its round-1/2 signature (no n_entries; ADVICE r3: a caller built against the old header
    must not have its workspace pointer read as a count).  Its piece capacity is what the workspace holds, the plan kernel
    checks it ON THE DEVICE: with a full-size workspace the result equals the v2 entry point's bit for bit; with room for
    fewer pieces than the hub rows need nothing is written past the scratch and the overflow flag (int at byte 8) is set."""
    import numpy as np
    import torch
    from wholegraph_amd import nn
    H, C, n_dst, n_src = 2, 8, 40, 6
    g = torch.Generator().manual_seed(3)
    deg = torch.full((n_dst,), 30, dtype=torch.int64)
    row_ptr = torch.zeros(n_dst + 1, dtype=torch.int32)
    row_ptr[1:] = torch.cumsum(deg, 0)
    E = int(row_ptr[-1])
    col = torch.randint(0, n_src, (E,), generator=g).int()           # 6 sources x 200 entries each: every source row is long
    x = torch.randn(n_src, H * C, generator=g).cuda()
    a_s, a_d = torch.randn(n_src, H, generator=g).cuda(), torch.randn(n_dst, H, generator=g).cuda()
    go = torch.randn(n_dst, H * C, generator=g).cuda()
    rp, ci = row_ptr.cuda(), col.cuda()
    out, alpha = nn.gat_forward(rp, ci, x, a_s, a_d, H, 0.2, need_alpha=True)
    want = nn.gat_backward(rp, ci, x, a_s, a_d, alpha, go, H, 0.2)
    row_ptr_t, edge_perm, edge_dst, _ = nn._csr_transpose(rp, ci, n_src, want_perm=True, want_dst=True)

    def run(ws_bytes):
        de = torch.empty((E, H), dtype=torch.float32, device="cuda")
        gx, gs, gd = torch.empty_like(x), torch.empty_like(a_s), torch.empty_like(a_d)
        buf = torch.full((ws_bytes + 4096,), 0x5A, dtype=torch.uint8, device="cuda")
        off = (-buf.data_ptr()) % 256
        rc = hiplib.wgamd_gat_csr_bwd_f32(rp.data_ptr(), ci.data_ptr(), n_dst, x.data_ptr(), x.stride(0), a_s.data_ptr(),
                                          a_d.data_ptr(), H, C, 0.2, alpha.data_ptr(), go.data_ptr(), go.stride(0),
                                          row_ptr_t.data_ptr(), edge_perm.data_ptr(), edge_dst.data_ptr(), n_src, de.data_ptr(),
                                          gx.data_ptr(), gx.stride(0), gs.data_ptr(), gd.data_ptr(), buf.data_ptr() + off,
                                          ws_bytes, None)
        torch.cuda.synchronize()
        flag = int(buf[off + 8:off + 12].view(torch.int32).item())
        guard = buf[off + ws_bytes:].cpu()
        return rc, (gx, gs, gd), flag, guard

    full = hiplib.wgamd_gat_csr_bwd_workspace_bytes(E, H, C)
    rc, got, flag, guard = run(full)
    assert rc == 0 and flag == 0 and bool((guard == 0x5A).all())
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    small = hiplib.wgamd_gat_csr_bwd_workspace_bytes(64 * 3, H, C)      # 4 slots; the hop needs 6 rows x 3 further pieces
    rc, got, flag, guard = run(small)
    assert rc == 0 and flag == 1, (rc, flag)
    assert bool((guard == 0x5A).all()), "the plan kernel wrote past its workspace"
    assert torch.equal(got[2], want[2])                                   # the destination-major part does not use it


def test_sage_layer_fused_padded_head_respects_the_callers_out(hiplib):
    """A 47-column head runs as 64 zero-padded columns.  `out` is written in place only when it is the [:, :47] view of a
    [n, 64] scratch; a wider buffer keeps its own columns 47.. untouched and a compact [n, 47] buffer IS filled."""
    import torch
    from wholegraph_amd import nn
    rp, col = _csr(300, 900, 12, 100)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((900, 100), generator=g, device="cuda")
    w_t = torch.randn((200, 47), generator=g, device="cuda") * 0.1
    bias = torch.randn(47, generator=g, device="cuda")
    rows = torch.arange(300, dtype=torch.int64, device="cuda")
    rpt, ct = torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda()
    ref = nn.sage_layer_fused_forward(rpt, ct, x, rows, w_t, bias, relu=True)
    scratch = torch.full((300, 64), 7.0, device="cuda")
    a = nn.sage_layer_fused_forward(rpt, ct, x, rows, w_t, bias, relu=True, out=scratch[:, :47])
    assert a.data_ptr() == scratch.data_ptr() and torch.equal(a, ref)
    wide = torch.full((300, 128), 7.0, device="cuda")
    b = nn.sage_layer_fused_forward(rpt, ct, x, rows, w_t, bias, relu=True, out=wide[:, :47])
    assert b.data_ptr() == wide.data_ptr() and torch.equal(b, ref) and bool((wide[:, 47:] == 7.0).all())
    compact = torch.full((300, 47), 7.0, device="cuda")
    c = nn.sage_layer_fused_forward(rpt, ct, x, rows, w_t, bias, relu=True, out=compact)
    assert c.data_ptr() == compact.data_ptr() and torch.equal(compact, ref)


@pytest.mark.parametrize("precision", ["bf16x3", "f32", "two_kernel"])
def test_sage_layer_products_shape_matches_frozen_fp64_golden(hiplib, precision):
    """tests/golden/sage_layer_golden.npz: fp64 expectations (506 sampled rows) of SAGE layer 1 at the products shape
    (F = 100 -> 256, mean, ReLU, block-diagonal 8-batch hop with hub sources and empty rows), frozen by
    tests/golden/make_golden.py — the fp32 results of every layer kernel stay inside north_star's 1e-5 across rewrites."""
    import hashlib
    import os
    import torch
    from graphgen import sage_layer_case
    from wholegraph_amd import nn
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "sage_layer_golden.npz"))
    rp, col, self_rows, x, w_t, bias = sage_layer_case()
    h = hashlib.sha256()
    for a in (rp, col, self_rows, x, w_t, bias):
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == str(z["inputs_sha256"]), "the regenerated inputs are not the ones the golden was frozen for"
    cu = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    if precision == "two_kernel":
        cat = nn.sage_aggregate_forward(cu(rp), cu(col), cu(x), cu(self_rows), True)
        got = torch.addmm(cu(bias), cat, cu(w_t)).relu_()
    else:
        got = nn.sage_layer_fused_forward(cu(rp), cu(col), cu(x), cu(self_rows), cu(w_t), cu(bias), relu=True, mean=True,
                                          precision=precision)
    got = got.cpu().numpy()[z["rows"]].astype(np.float64)
    ref, scale = np.maximum(z["pre_activation"], 0.0), z["scale"]
    assert np.all(np.abs(got - ref) <= 1e-5 * scale + 1e-7), np.abs(got - ref).max()
    big = np.abs(ref) >= 0.1 * scale
    assert big.sum() > 1000 and (np.abs(got - ref)[big] / np.abs(ref)[big]).max() <= 1e-5
