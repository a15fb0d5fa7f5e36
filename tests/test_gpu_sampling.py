"""GPU parity: one-hop uniform sampling through the C ABI vs the oracle — bit-exact on all four
outputs, the bar of the reference's own gtest
(/root/reference/cpp/tests/wholegraph_ops/wholegraph_csr_unweighted_sample_without_replacement_tests.cu:330-353,
parameter sets :95-107,382-403) and pytest
(python/pylibwholegraph/pylibwholegraph/tests/wholegraph_torch/ops/test_wholegraph_unweighted_sample_without_replacement.py:309-361)."""
import numpy as np
import pytest

from graphgen import powerlaw_csr, random_csr

pytestmark = pytest.mark.gpu


def _run(oracle_mod, row_ptr, col, seeds, M, rs):
    import torch
    from wholegraph_amd import wholegraph_ops as ops
    out = ops.unweighted_sample_without_replacement(
        torch.from_numpy(row_ptr).cuda(), torch.from_numpy(col).cuda(), torch.from_numpy(seeds).cuda(), M,
        random_seed=rs, need_center_local_output=True, need_edge_output=True)
    ref = oracle_mod.unweighted_sample(row_ptr, col, seeds, M, rs)
    names = ["sample_offset", "dest", "center_localid", "edge_gid"]
    for name, a, b in zip(names, out, ref):
        a = a.cpu().numpy()
        assert a.dtype == b.dtype, (name, a.dtype, b.dtype)
        assert a.shape == b.shape, (name, a.shape, b.shape)
        assert np.array_equal(a, b), f"{name} differs at {np.nonzero(a != b)[0][:8]}"
    return out


# the reference gtest parameter sets + every kernel path of the launch table
@pytest.mark.parametrize("nodes,edges,n_seeds,M", [
    (9703, 104323, 512, 50), (23289, 689403, 35, 10), (103, 1043, 13, 11), (103, 1043, 13, -1),
    (5000, 400000, 300, 25), (5000, 400000, 300, 32), (5000, 400000, 300, 33), (5000, 400000, 300, 64),
    (5000, 400000, 300, 96), (5000, 400000, 200, 97), (3000, 900000, 100, 200), (3000, 900000, 60, 384),
    (3000, 900000, 60, 385), (2000, 3000000, 40, 1024), (1500, 3000000, 24, 1500),
])
@pytest.mark.parametrize("seed_dtype,col_dtype", [(np.int32, np.int32), (np.int64, np.int64),
                                                  (np.int64, np.int32), (np.int32, np.int64)])
def test_uniform_vs_oracle(oracle_mod, hiplib, nodes, edges, n_seeds, M, seed_dtype, col_dtype):
    row_ptr, col = random_csr(nodes, edges, seed=nodes + M, col_dtype=col_dtype)
    rng = np.random.default_rng(abs(M))
    seeds = rng.integers(0, nodes, n_seeds).astype(seed_dtype)
    _run(oracle_mod, row_ptr, col, seeds, M, 0x1234567 + M)


@pytest.mark.parametrize("M", [5, 10, 15, 25])
def test_powerlaw_baseline_fanouts(oracle_mod, hiplib, M):
    row_ptr, col = powerlaw_csr(20000, 30, seed=3, max_deg=6000)
    rng = np.random.default_rng(1)
    seeds = rng.permutation(20000)[:4096].astype(np.int64)
    out = _run(oracle_mod, row_ptr, col, seeds, M, 62)
    # size-independent properties: no repeated edge inside a seed, every edge is a real CSR edge
    off, dst, lid, gid = [t.cpu().numpy() for t in out]
    assert np.array_equal(col[gid], dst)
    assert np.all(np.diff(np.sort(gid + lid.astype(np.int64) * (1 << 40))) != 0)
    assert np.all(gid >= row_ptr[seeds[lid]]) and np.all(gid < row_ptr[seeds[lid] + 1])


def test_edge_cases(oracle_mod, hiplib):
    row_ptr, col = random_csr(500, 20000, seed=9, zero_deg_frac=0.3)
    deg = np.diff(row_ptr)
    M = 25
    picks = [np.nonzero(deg == 0)[0][:5], np.nonzero(deg == M)[0][:5], np.nonzero(deg == M + 1)[0][:5],
             np.nonzero(deg > M)[0][:7], np.array([0, 0, 499, 499])]
    seeds = np.concatenate(picks).astype(np.int64)  # repeated seeds, degree 0 / == M / M+1
    _run(oracle_mod, row_ptr, col, seeds, M, 1)
    _run(oracle_mod, row_ptr, col, seeds[:1], M, 2)          # single seed
    _run(oracle_mod, row_ptr, col, seeds[:0], M, 3)          # empty
    _run(oracle_mod, row_ptr, col, seeds, 1, 4)              # M = 1
    _run(oracle_mod, row_ptr, col, seeds, 0, 5)              # M = 0 -> sample all
    _run(oracle_mod, row_ptr, col, seeds, M, 2**64 - 1)      # max seed


def test_invalid_inputs_return_codes(hiplib):
    import torch
    from wholegraph_amd import WholeMemoryError, _lib, wholegraph_ops as ops
    rp = torch.zeros(11, dtype=torch.int32, device="cuda")   # row_ptr must be INT64
    col = torch.zeros(4, dtype=torch.int64, device="cuda")
    seeds = torch.zeros(3, dtype=torch.int64, device="cuda")
    with pytest.raises(WholeMemoryError) as e:
        ops.unweighted_sample_without_replacement(rp, col, seeds, 5)
    assert e.value.code == _lib.WHOLEMEMORY_LOGIC_ERROR
    with pytest.raises(WholeMemoryError) as e:
        ops.unweighted_sample_without_replacement(rp.long(), col.float(), seeds, 5)
    assert e.value.code == _lib.WHOLEMEMORY_INVALID_INPUT


@pytest.mark.parametrize("M", [1, 5, 10, 25, 64, 300])
@pytest.mark.parametrize("col_dtype,seed_dtype", [(np.int64, np.int64), (np.int32, np.int32), (np.int32, np.int64)])
def test_uniform_with_replacement_vs_oracle(oracle_mod, hiplib, M, col_dtype, seed_dtype):
    """wgamd_csr_uniform_sample_with_replacement (cugraph_pyg `replace=True`) against the oracle's statement of its draw
    layout, bit-exact on all four outputs; every seed with neighbours yields exactly M picks, repeats included."""
    import torch
    from wholegraph_amd import wholegraph_ops as ops
    row_ptr, col = random_csr(900, 40000, seed=M, col_dtype=col_dtype, zero_deg_frac=0.1)
    seeds = np.random.default_rng(M).integers(0, 900, 333).astype(seed_dtype)
    out = ops.unweighted_sample_with_replacement(torch.from_numpy(row_ptr).cuda(), torch.from_numpy(col).cuda(),
                                                 torch.from_numpy(seeds).cuda(), M, random_seed=777 + M,
                                                 need_center_local_output=True, need_edge_output=True)
    off, dst, lid, gid = (t.cpu().numpy() for t in out)
    ooff, odst, olid, ogid = oracle_mod.unweighted_sample_with_replacement(row_ptr, col, seeds, M, 777 + M)
    assert np.array_equal(off, ooff) and np.array_equal(dst, odst) and np.array_equal(lid, olid) and np.array_equal(gid, ogid)
    deg = np.diff(row_ptr)[seeds]
    assert np.array_equal(np.diff(off), np.where(deg > 0, M, 0))
    assert np.all(gid >= row_ptr[seeds[lid]]) and np.all(gid < row_ptr[seeds[lid] + 1]) and np.array_equal(col[gid], dst)
    short = np.nonzero((deg > 0) & (deg < M))[0]   # more picks than the row has neighbours: repeats must occur
    assert M < 64 or len(short) > 0
    for i in short[:5]:
        assert len(set(gid[off[i]:off[i + 1]].tolist())) < M
    # empty input
    e = ops.unweighted_sample_with_replacement(torch.from_numpy(row_ptr).cuda(), torch.from_numpy(col).cuda(),
                                               torch.from_numpy(seeds[:0]).cuda(), M, random_seed=1)
    assert e[0].tolist() == [0] and e[1].numel() == 0
