"""The oracle against its frozen golden vectors, its pure-Python twins and the small known-answer
cases the reference's own tests/docstrings hold."""
import os

import numpy as np
import pytest

from graphgen import powerlaw_csr, random_csr

GOLD = os.path.join(os.path.dirname(__file__), "golden", "hotpath_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_karate_fixture_is_the_reference_dataset(gold):
    # /root/reference/datasets/karate.csv: 34 nodes, 156 directed lines = 78 undirected edges, symmetric
    rp, col = gold["karate_row_ptr"], gold["karate_col"]
    assert rp.shape == (35,) and col.shape == (156,)
    pairs = {(int(s), int(d)) for s in range(34) for d in col[rp[s]:rp[s + 1]]}
    assert all((d, s) in pairs for s, d in pairs)


def test_karate_walk_matches_golden(oracle_mod, gold):
    rp, col = gold["karate_row_ptr"], gold["karate_col"]
    for b, seeds in enumerate(np.array_split(np.arange(34, dtype=np.int64), [16, 32])):
        tg, ei, orp, oci = oracle_mod.multilayer_sample(rp, col, seeds, [5, 5], [62 + 2 * b, 63 + 2 * b])
        for i, t in enumerate(tg):
            assert np.array_equal(t, gold[f"karate_b{b}_target_gids_{i}"])
        for i in range(2):
            assert np.array_equal(orp[i], gold[f"karate_b{b}_csr_row_ptr_{i}"])
            assert np.array_equal(oci[i], gold[f"karate_b{b}_csr_col_ind_{i}"])
            assert np.array_equal(ei[i], gold[f"karate_b{b}_edge_indice_{i}"])
        # walk invariants (graph_structure.py:186-195): seeds first; every sampled edge is a karate edge
        assert np.array_equal(tg[0][: len(seeds)], seeds)
        for i in range(2):
            dst_gid = tg[i + 1][ei[i][1]]
            src_gid = tg[i][ei[i][0]]
            for s, d in zip(dst_gid, src_gid):
                assert d in col[rp[s]:rp[s + 1]]


@pytest.mark.parametrize("M", [11, -1, 40, 70])
def test_g103_matches_golden_and_python_twin(oracle_mod, gold, M):
    rp, col, seeds = gold["g103_row_ptr"], gold["g103_col"], gold["g103_seeds"]
    got = oracle_mod.unweighted_sample(rp, col, seeds, M, 1234)
    for name, a in zip(("offset", "dst", "lid", "gid"), got):
        assert np.array_equal(a, gold[f"g103_M{M}_{name}"]), name
    twin = oracle_mod.py_unweighted_sample_small(rp, col, seeds, M, 1234)
    for a, b in zip(got, twin):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("M", [1, 5, 25, 32, 33, 64, 96, 97, 130, 300])
def test_c_oracle_vs_python_twin_random(oracle_mod, M):
    rp, col = random_csr(60, 9000, seed=M, col_dtype=np.int64, zero_deg_frac=0.1)
    seeds = np.random.default_rng(M).integers(0, 60, 9).astype(np.int64)
    a = oracle_mod.unweighted_sample(rp, col, seeds, M, 777 + M)
    b = oracle_mod.py_unweighted_sample_small(rp, col, seeds, M, 777 + M)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_uniform_sampling_properties(oracle_mod):
    rp, col = powerlaw_csr(5000, 30, seed=1, max_deg=2000)
    seeds = np.arange(5000, dtype=np.int64)
    deg = np.diff(rp)
    for M in (10, 25):
        off, dst, lid, gid = oracle_mod.unweighted_sample(rp, col, seeds, M, 5)
        assert np.array_equal(np.diff(off), np.minimum(deg, M))          # counts
        assert np.array_equal(col[gid], dst)                              # dst is the edge's column
        assert np.all((gid >= rp[lid]) & (gid < rp[lid + 1]))              # edge belongs to its seed's row
        assert np.unique(gid).size == gid.size                            # without replacement
        whole = deg <= M                                                  # small rows copied in CSR order
        for i in np.nonzero(whole)[0][:50]:
            assert np.array_equal(gid[off[i]:off[i + 1]], np.arange(rp[i], rp[i + 1]))
    # different seeds -> different picks; same seed -> identical (determinism)
    a = oracle_mod.unweighted_sample(rp, col, seeds, 10, 1)[3]
    assert np.array_equal(a, oracle_mod.unweighted_sample(rp, col, seeds, 10, 1)[3])
    assert not np.array_equal(a, oracle_mod.unweighted_sample(rp, col, seeds, 10, 2)[3])


def test_uniformity_chi_square(oracle_mod):
    # one row of 40 neighbours sampled M=10 under 4000 different seeds: every neighbour ~ 1000 hits
    rp = np.array([0, 40], np.int64)
    col = np.arange(40, dtype=np.int64)
    hits = np.zeros(40)
    for s in range(4000):
        hits[oracle_mod.unweighted_sample(rp, col, np.zeros(1, np.int64), 10, s)[1]] += 1
    chi2 = ((hits - 1000.0) ** 2 / 1000.0).sum()
    assert chi2 < 80  # 39 dof, p(chi2 > 80) ~ 1e-4


def test_append_unique_reference_docstring_example(oracle_mod):
    # python/pylibwholegraph/pylibwholegraph/torch/graph_ops.py:21-29
    u, m = oracle_mod.append_unique(np.array([3, 11, 2, 10]), np.array([4, 5, 2, 11, 6, 9, 10, 5]))
    assert list(u[:4]) == [3, 11, 2, 10] and sorted(u[4:]) == [4, 5, 6, 9]
    assert list(u[m]) == [4, 5, 2, 11, 6, 9, 10, 5]
    # first-appearance order == the reference host oracle's order (append_unique_test_utils.cu:52-84)
    assert list(u) == [3, 11, 2, 10, 4, 5, 6, 9] and list(m) == [4, 5, 2, 1, 6, 7, 3, 5]


def test_append_unique_vs_numpy(oracle_mod):
    rng = np.random.default_rng(0)
    for dtype in (np.int32, np.int64):
        t = rng.permutation(5000)[:300].astype(dtype)
        n = rng.integers(0, 5000, 4000).astype(dtype)
        u, m = oracle_mod.append_unique(t, n)
        assert np.array_equal(u[:300], t) and np.array_equal(u[m], n)
        new = n[~np.isin(n, t)]
        _, first = np.unique(new, return_index=True)
        assert np.array_equal(u[300:], new[np.sort(first)])
    u, m = oracle_mod.append_unique(np.array([], np.int64), np.array([], np.int64))
    assert u.size == 0 and m.size == 0


def test_weighted_sampling_properties(oracle_mod, gold):
    rp, col, seeds, w = gold["g103_row_ptr"], gold["g103_col"], gold["g103_seeds"], gold["g103_weight"]
    off, dst, lid, gid = oracle_mod.weighted_sample(rp, col, w, seeds, 5, 99)
    assert np.array_equal(off, gold["g103_w5_offset"]) and np.array_equal(gid, gold["g103_w5_gid"])
    # biased sampling never picks a weight-0 edge while positive-weight edges remain
    # (cugraph_pyg/tests/loader/test_neighbor_loader.py:99-133)
    w0 = w.copy()
    w0[::2] = 0.0
    off, dst, lid, gid = oracle_mod.weighted_sample(rp, col, w0, seeds, 3, 7)
    deg_pos = np.array([np.count_nonzero(w0[rp[s]:rp[s + 1]]) for s in seeds])
    for i in range(len(seeds)):
        seg = gid[off[i]:off[i + 1]]
        if rp[seeds[i] + 1] - rp[seeds[i]] > 3 and deg_pos[i] >= 3:
            assert np.all(w0[seg] > 0)
    # heavier edges are picked more often
    rp1, col1 = np.array([0, 20], np.int64), np.arange(20, dtype=np.int64)
    w1 = np.ones(20, np.float32)
    w1[:5] = 10.0
    heavy = sum(np.count_nonzero(oracle_mod.weighted_sample(rp1, col1, w1, np.zeros(1, np.int64), 5, s)[1] < 5) for s in range(300))
    assert heavy > 0.55 * 1500  # uniform sampling would give 0.25


def test_self_loop_and_aggregation_oracles(oracle_mod):
    rp = np.array([0, 2, 2, 5], np.int32)
    col = np.array([7, 8, 1, 2, 3], np.int32)
    orp, oc = oracle_mod.csr_add_self_loop(rp, col)
    assert list(orp) == [0, 3, 4, 8] and list(oc) == [0, 7, 8, 1, 2, 1, 2, 3]
    x = np.arange(40, dtype=np.float32).reshape(10, 4)
    out = oracle_mod.spmm_csr(rp, col, x, mean=True, acc_double=True)
    np.testing.assert_allclose(out[0], (x[7] + x[8]) / 2)
    np.testing.assert_allclose(out[1], 0)
    np.testing.assert_allclose(out[2], x[1:4].mean(0))
    # GAT oracle vs a direct numpy softmax
    rng = np.random.default_rng(0)
    xs = rng.standard_normal((10, 2, 3)).astype(np.float32)
    a_s, a_d = rng.standard_normal((10, 2)).astype(np.float32), rng.standard_normal((3, 2)).astype(np.float32)
    o, alpha = oracle_mod.gat_csr(rp, col, xs, a_s, a_d, 0.2)
    for i in (0, 2):
        nb = col[rp[i]:rp[i + 1]]
        s = a_s[nb] + a_d[i]
        s = np.where(s > 0, s, 0.2 * s)
        p = np.exp(s - s.max(0))
        p /= p.sum(0)
        np.testing.assert_allclose(alpha[rp[i]:rp[i + 1]], p, rtol=1e-6)
        np.testing.assert_allclose(o[i], (p[:, :, None] * xs[nb]).sum(0), rtol=1e-5, atol=1e-6)
    assert np.all(o[1] == 0)
