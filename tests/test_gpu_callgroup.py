"""GPU parity of the no-sync walk: a CALL GROUP of G mini-batches processed by one launch sequence
must give every mini-batch exactly what the oracle's single-batch walk gives with that batch's
seeds (per-hop CSR, renumber map, COO) — plus the scalar-seed single-batch C entry point."""
import numpy as np
import pytest

from graphgen import powerlaw_csr

pytestmark = pytest.mark.gpu


def _check(oracle_mod, row_ptr, col, seeds_b, fanouts, rs_b, got):
    otg, oei, orp, oci = oracle_mod.multilayer_sample(row_ptr, col, seeds_b, fanouts, rs_b)
    tg, ei, rp, ci = got
    for name, a, b in (("target_gids", tg, otg), ("edge_indice", ei, oei), ("csr_row_ptr", rp, orp), ("csr_col_ind", ci, oci)):
        for lvl, (x, y) in enumerate(zip(a, b)):
            assert np.array_equal(x.cpu().numpy(), y), (name, lvl)


@pytest.mark.parametrize("G,B", [(1, 256), (5, 128), (16, 64), (3, 1)])
@pytest.mark.parametrize("fanouts", [[25, 10], [15, 10, 5], [5], [20, 7], [32, 3]])
@pytest.mark.parametrize("dtype,compact", [(np.int64, True), (np.int64, False), (np.int32, True)])
@pytest.mark.parametrize("grouped", [False, True])
def test_call_group_equals_per_batch_oracle(oracle_mod, hiplib, G, B, fanouts, dtype, compact, grouped):
    """``compact``: int64 ids over the 32-bit twin of the column array (WGAMD_HOP_COL_INT32, the default for graphs below
    2^31 vertices) — the same bits as the int64 columns and as the oracle.  ``grouped``: every hop walks its frontier grouped
    by vertex-id range (wgamd_set_sample_locality_min(1); off by default) — same bits.  Fan-outs 5 / 10 / 15 / 25 run lane groups
    exactly as wide as the fan-out, the others the next of 8 / 16 / 32 (sample_uniform_multi_kernel)."""
    import torch
    hiplib.wgamd_set_sample_locality_min(1 if grouped else 0)
    try:
        _call_group_equals_per_batch_oracle(oracle_mod, G, B, fanouts, dtype, compact)
    finally:
        hiplib.wgamd_set_sample_locality_min(0)


def _call_group_equals_per_batch_oracle(oracle_mod, G, B, fanouts, dtype, compact):
    import torch
    from wholegraph_amd.fused import NoSyncWalk, HOP_COL_INT32
    row_ptr, col = powerlaw_csr(20000, 18, seed=4, col_dtype=dtype, max_deg=3000)
    rng = np.random.default_rng(G * 100 + B)
    seeds = np.concatenate([rng.permutation(20000)[:B] for _ in range(G)]).astype(dtype)  # batches overlap on purpose
    rs = [[1000 * k + b + 62 for b in range(G)] for k in range(len(fanouts))]
    walk = NoSyncWalk(torch.from_numpy(row_ptr).cuda(), torch.from_numpy(col).cuda(), B, fanouts, torch.from_numpy(seeds).dtype, G,
                      compact_col=compact)
    assert bool(walk.flags & HOP_COL_INT32) == (compact and dtype == np.int64)
    assert walk.col.dtype == (torch.int32 if (compact or dtype == np.int32) else torch.int64)
    res = walk.run(torch.from_numpy(seeds).cuda(), rs)
    assert all(u.dtype == torch.from_numpy(seeds).dtype for u in res.unique)      # the API's id width, whatever the columns'
    per_batch = res.finalize_batches()
    assert len(per_batch) == G
    for b in range(G):
        _check(oracle_mod, row_ptr, col, seeds[b * B:(b + 1) * B], fanouts, [rs[k][b] for k in range(len(fanouts))], per_batch[b])
    # global (block-diagonal) view is consistent: unique_seg / counts / -1 padding
    for k in range(len(fanouts)):
        useg = res.unique_seg[k].cpu().numpy()
        n_e, n_u = res.counts[k].cpu().numpy()
        assert useg[0] == 0 and useg[-1] == n_u and np.all(np.diff(useg) > 0)
        assert np.all(res.unique[k][n_u:].cpu().numpy() == -1)
        nbr = res.neighbor_row[k][:n_e].cpu().numpy()
        assert nbr.min() >= 0 and nbr.max() < n_u
    # a second run with the same seeds is bit-identical (buffers reused, determinism)
    res2 = walk.run(torch.from_numpy(seeds).cuda(), torch.tensor(rs, dtype=torch.int64))
    for k in range(len(fanouts)):
        n_e, n_u = res.counts[k].cpu().numpy()
        assert torch.equal(res.unique[k][:n_u], res2.unique[k][:n_u])
        assert torch.equal(res.neighbor_row[k][:n_e], res2.neighbor_row[k][:n_e])


def test_unpadded_unique_lists_are_identical_below_the_live_end(oracle_mod, hiplib):
    """WGAMD_HOP_NO_UNIQUE_PAD (pad_unique=False): the capacity slack of `unique` is left alone, everything a consumer that
    reads the sizes can see is what the padded walk and the oracle give."""
    import torch
    from wholegraph_amd.fused import NoSyncWalk
    G, B, fanouts = 7, 96, [25, 10]
    row_ptr, col = powerlaw_csr(20000, 18, seed=4, max_deg=3000)
    rng = np.random.default_rng(5)
    seeds = np.concatenate([rng.permutation(20000)[:B] for _ in range(G)]).astype(np.int64)
    rs = [[31 * k + b + 7 for b in range(G)] for k in range(len(fanouts))]
    rp_d, col_d = torch.from_numpy(row_ptr).cuda(), torch.from_numpy(col).cuda()
    padded = NoSyncWalk(rp_d, col_d, B, fanouts, torch.int64, G).run(torch.from_numpy(seeds).cuda(), rs)
    walk = NoSyncWalk(rp_d, col_d, B, fanouts, torch.int64, G, pad_unique=False)
    res = walk.run(torch.from_numpy(seeds).cuda(), rs)
    for k in range(len(fanouts)):
        n_e, n_u = res.counts[k].cpu().numpy()
        assert torch.equal(res.counts[k], padded.counts[k]) and torch.equal(res.unique_seg[k], padded.unique_seg[k])
        assert torch.equal(res.unique[k][:n_u], padded.unique[k][:n_u])
        if k + 1 < len(fanouts):   # the batch of every unique entry = target_batch of the next hop
            assert torch.equal(res.target_batch[k + 1][:n_u], padded.target_batch[k + 1][:n_u])
        assert torch.equal(res.neighbor_row[k][:n_e], padded.neighbor_row[k][:n_e])
        assert np.all(padded.unique[k][n_u:].cpu().numpy() == -1)
    per_batch = res.finalize_batches()
    for b in range(G):
        _check(oracle_mod, row_ptr, col, seeds[b * B:(b + 1) * B], fanouts, [rs[k][b] for k in range(len(fanouts))], per_batch[b])


def test_unknown_hop_flag_bits_are_refused(hiplib):
    import torch
    import wholegraph_amd._lib as L
    from wholegraph_amd.fused import NoSyncWalk
    row_ptr, col = powerlaw_csr(2000, 8, seed=1, max_deg=300)
    walk = NoSyncWalk(torch.from_numpy(row_ptr).cuda(), torch.from_numpy(col).cuda(), 16, [5], torch.int64, 2)
    walk.flags = 6
    with pytest.raises(L.WholeMemoryError):
        walk.run(torch.arange(32, dtype=torch.int64).cuda(), [[1, 2]])


def test_single_batch_scalar_seed_entry(oracle_mod, hiplib):
    import torch
    from wholegraph_amd.fused import SingleBatchNoSyncWalk
    row_ptr, col = powerlaw_csr(20000, 18, seed=4, max_deg=3000)
    seeds = np.random.default_rng(0).permutation(20000)[:300].astype(np.int64)
    w = SingleBatchNoSyncWalk(torch.from_numpy(row_ptr).cuda(), torch.from_numpy(col).cuda(), 300, [25, 10])
    got = w.run(torch.from_numpy(seeds).cuda(), [2**64 - 5, 77]).finalize()
    _check(oracle_mod, row_ptr, col, seeds, [25, 10], [2**64 - 5, 77], got)


def test_batched_block_diagonal_aggregation_matches_per_batch(oracle_mod, hiplib):
    """The concatenated (block-diagonal) CSR feeds ONE gather + ONE SpMM for the whole call group;
    each batch's slice equals the per-batch oracle aggregation."""
    import torch
    from wholegraph_amd import nn
    from wholegraph_amd.fused import NoSyncWalk
    from wholegraph_amd.tensor import local_gather
    G, B, fan = 4, 96, [10, 5]
    row_ptr, col = powerlaw_csr(8000, 15, seed=9, max_deg=900)
    feat = np.random.default_rng(1).standard_normal((8000, 100)).astype(np.float32)
    seeds = np.concatenate([np.random.default_rng(b).permutation(8000)[:B] for b in range(G)]).astype(np.int64)
    rs = [[10 + b for b in range(G)], [20 + b for b in range(G)]]
    walk = NoSyncWalk(torch.from_numpy(row_ptr).cuda(), torch.from_numpy(col).cuda(), B, fan, torch.int64, G)
    res = walk.run(torch.from_numpy(seeds).cuda(), rs)
    cap = res.unique[1].shape[0]
    x = torch.zeros((cap, 100), device="cuda")
    local_gather(torch.from_numpy(feat).cuda(), res.unique[1], x)          # -1 padded slack is skipped
    tseg = res.target_seg[1].cpu().numpy()
    # rows = all LIVE hop-2 targets (offsets past the live count are capacity slack the walk never writes)
    agg = nn.spmm_csr_forward(res.offsets[1][:tseg[G] + 1], res.neighbor_row[1], x, True)
    for b in range(G):
        otg, oei, orp, oci = oracle_mod.multilayer_sample(row_ptr, col, seeds[b * B:(b + 1) * B], fan, [10 + b, 20 + b])
        ref = oracle_mod.spmm_csr(orp[0], oci[0], feat[otg[0]], mean=True, acc_double=False)
        assert np.array_equal(agg[tseg[b]:tseg[b + 1]].cpu().numpy(), ref)


@pytest.mark.parametrize("G,B", [(1, 200), (6, 100), (16, 33)])
@pytest.mark.parametrize("fanouts", [[25, 10], [10, 5, 3], [4]])
@pytest.mark.parametrize("dtype", [np.int64, np.int32])
def test_pyg_style_call_group_walk_vs_oracle(oracle_mod, hiplib, G, B, fanouts, dtype):
    """PyG hop semantics (expand only newly discovered vertices) for a call group: every mini-batch must equal
    the oracle composition sample(frontier) -> append_unique(all nodes so far, neighbours)."""
    import torch
    from wholegraph_amd.fused import PygNoSyncWalk
    from test_gpu_pyg_loader import oracle_neighbor_sample
    from cugraph_pyg_amd.sampler.sampler import hop_seed
    row_ptr, col = powerlaw_csr(15000, 14, seed=8, col_dtype=dtype, max_deg=2500)
    eid = np.random.default_rng(3).permutation(col.size).astype(np.int64)     # arbitrary slot -> edge-id map
    rng = np.random.default_rng(G + B)
    seeds = np.concatenate([rng.permutation(15000)[:B] for _ in range(G)]).astype(dtype)
    rstate = [500 + b for b in range(G)]
    rs = [[hop_seed(rstate[b], k) for b in range(G)] for k in range(len(fanouts))]
    walk = PygNoSyncWalk(torch.from_numpy(row_ptr).cuda(), torch.from_numpy(col).cuda(), B, fanouts, G)
    res = walk.run(torch.from_numpy(seeds).cuda(), rs)
    per_batch = res.finalize_batches(torch.from_numpy(eid).cuda())
    for b in range(G):
        node, row, colv, edge, nn, ne = oracle_neighbor_sample(oracle_mod, row_ptr, col, eid, seeds[b * B:(b + 1) * B],
                                                               fanouts, rstate[b])
        g_node, g_row, g_col, g_edge, g_nn, g_ne = per_batch[b]
        assert np.array_equal(g_node.cpu().numpy(), node)
        assert np.array_equal(g_row.cpu().numpy(), row) and np.array_equal(g_col.cpu().numpy(), colv)
        assert np.array_equal(g_edge.cpu().numpy(), edge)
        assert g_nn == nn and g_ne == ne


_RENUMBER_WORKER = r"""
import sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/cugraph-gnn_amd"); sys.path.insert(0, sys.argv[1] + "/tests")
import oracle
from graphgen import powerlaw_csr
from wholegraph_amd.fused import NoSyncWalk
oracle.build()
for dtype, G, B, fanouts, compact in ((np.int32, 6, 256, [25, 10], True), (np.int64, 3, 200, [15, 10, 5], True),
                                      (np.int64, 5, 300, [25, 10], False), (np.int32, 9, 7, [5, 5], True)):
    row_ptr, col = powerlaw_csr(30000, 18, seed=11, col_dtype=dtype, max_deg=3000)
    rng = np.random.default_rng(G + B)
    seeds = np.concatenate([rng.permutation(30000)[:B] for _ in range(G)]).astype(dtype)
    rs = [[500 * k + b + 3 for b in range(G)] for k in range(len(fanouts))]
    walk = NoSyncWalk(torch.from_numpy(row_ptr).cuda(), torch.from_numpy(col).cuda(), B, fanouts, torch.from_numpy(seeds).dtype, G,
                      compact_col=compact)
    per_batch = walk.run(torch.from_numpy(seeds).cuda(), rs).finalize_batches()
    for b in range(G):
        want = oracle.multilayer_sample(row_ptr, col, seeds[b * B:(b + 1) * B], fanouts, [rs[k][b] for k in range(len(fanouts))])
        for name, got_l, want_l in zip(("target_gids", "edge_indice", "csr_row_ptr", "csr_col_ind"), per_batch[b], want):
            for lvl, (x, y) in enumerate(zip(got_l, want_l)):
                assert np.array_equal(x.cpu().numpy(), y), (name, lvl, b, G, B)
print("RENUMBER_OK")
"""


@pytest.mark.parametrize("env", [
    {"WGAMD_RENUMBER_KEYS_TARGET": "100000000"},   # one hash range per batch: it overfills the LDS table and must split
    {"WGAMD_RENUMBER_KEYS_TARGET": "20000"},       # some ranges overfill, some do not
    {"WGAMD_RENUMBER_NO_LDS": "1"},                # the device-wide packed table (what huge call groups fall back to)
])
def test_renumber_paths_agree_with_the_oracle(hiplib, env):
    """The per-batch LDS renumbering splits a hash range whose keys do not fit its table; both that path and the
    device-wide table must give the oracle's first-appearance order (the library reads the knobs once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", _RENUMBER_WORKER, root], env=dict(os.environ, **env), capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0 and "RENUMBER_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
