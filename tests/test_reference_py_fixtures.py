"""oracle/wg_oracle.c against fixtures produced by the REFERENCE's own Python host samplers.

tests/golden/reference_py_*.npz were written by tests/golden/make_reference_fixtures.py, which imports (build container
only) the reference's pure-Python restatements of its device ops —
tests/wholegraph_torch/ops/test_wholegraph_unweighted_sample_without_replacement.py:22-211,
test_wholegraph_weighted_sample_without_replacement.py:22-166, test_graph_append_unique.py:8-19,
test_graph_add_csr_self_loop.py:9-28, test_utils/test_comm.py:44-142 (all under
/root/reference/python/pylibwholegraph/pylibwholegraph/) — and runs them unmodified; only the two host RNG helpers they call
are this library's exported C symbols.  Everything between a raw PCG draw and a sampled edge is therefore reference code:
launch tables, RNG→index mapping, Fisher–Yates table, lane ownership of the weighted sampler, key composition, top-M.
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))


def uniform_cases():
    z = load("reference_py_unweighted.npz")
    return [str(c) for c in z["cases"]]


def weighted_cases():
    z = load("reference_py_weighted.npz")
    return [str(c) for c in z["cases"]]


def unpack_uniform(z, case):
    tag, g, M, seed, kd = case.split("|")
    col = z[f"{g}_col"].astype(np.int32 if kd == "int32" else np.int64)
    exp = tuple(z[f"{tag}_{k}"] for k in ("offset", "dst", "lid", "gid"))
    return z[f"{g}_row_ptr"], col, z[f"{tag}_centres"], int(M), int(seed), exp


def unpack_weighted(z, case):
    tag, g, M, seed, kd, wd = case.split("|")
    col = z[f"{g}_col"].astype(np.int32 if kd == "int32" else np.int64)
    w = z[f"{g}_weight_f32"].astype(np.float32 if wd == "f32" else np.float64)
    exp = tuple(z[f"{tag}_{k}"] for k in ("offset", "dst", "lid", "gid"))
    return z[f"{g}_row_ptr"], col, w, z[f"{tag}_centres"], int(M), int(seed), exp


def check_weighted_sets(row_ptr, col, centres, M, got, exp):
    """The reference's own bar (…weighted…py:301-346): offsets and centre ids exact, picks equal per seed as SORTED
    sets (its host function emits them in key order, the device in whatever order its top-k leaves them)."""
    off, dst, lid, gid = got
    eoff, edst, elid, egid = exp
    assert np.array_equal(off, eoff) and np.array_equal(lid, elid)
    assert np.array_equal(col[gid], dst)
    for i in range(len(centres)):
        a, b = np.sort(gid[off[i]:off[i + 1]]), np.sort(egid[off[i]:off[i + 1]])
        assert np.array_equal(a, b), f"seed {i}: picks differ {set(a.tolist()) ^ set(b.tolist())}"
        assert np.array_equal(np.sort(dst[off[i]:off[i + 1]]), np.sort(edst[off[i]:off[i + 1]]))


def test_fixture_coverage():
    """The fixtures hold the reference pytest's own parameter sets and every launch-table class."""
    u = [c.split("|") for c in uniform_cases()]
    assert {(g, int(M)) for _, g, M, _, _ in u} >= {("g103", 11), ("g103", -1), ("g103", 25), ("g103", 10), ("g103", 15),
                                                    ("g103", 5), ("g103", 40), ("g103", 70), ("gwide", 200),
                                                    ("ghub", 1000), ("ghub", 1024)}
    z = load("reference_py_unweighted.npz")
    assert z["g103_row_ptr"].shape == (104,) and z["g103_row_ptr"][-1] == 1043 and z["u0_centres"].shape == (13,)
    w = [c.split("|") for c in weighted_cases()]
    assert {(g, int(M), wd) for _, g, M, _, _, wd in w} >= {("g113", 11, "f32"), ("g113", 11, "f64"), ("gwide", 300, "f32")}
    zw = load("reference_py_weighted.npz")
    assert zw["g113_row_ptr"].shape == (114,) and zw["g113_row_ptr"][-1] == 1043
    # the sampled (not copied-whole) branch is what the fixtures are for: most cases must have rows longer than M
    sampled = 0
    for c in uniform_cases():
        rp, col, centres, M, seed, exp = unpack_uniform(z, c)
        sampled += bool(M > 0 and np.any(np.diff(rp)[centres] > M))
    assert sampled >= 24, sampled


@pytest.mark.parametrize("case", uniform_cases())
def test_oracle_uniform_equals_reference_python(oracle_mod, case):
    z = load("reference_py_unweighted.npz")
    rp, col, centres, M, seed, exp = unpack_uniform(z, case)
    got = oracle_mod.unweighted_sample(rp, col, centres, M, seed)
    for name, a, b in zip(("sample_offset", "dest", "center_localid", "edge_gid"), got, exp):
        assert a.dtype == b.dtype and np.array_equal(a, b), name   # bit-exact, …unweighted…py:309-345


@pytest.mark.parametrize("case", weighted_cases())
def test_oracle_weighted_equals_reference_python(oracle_mod, case):
    z = load("reference_py_weighted.npz")
    rp, col, w, centres, M, seed, exp = unpack_weighted(z, case)
    got = oracle_mod.weighted_sample(rp, col, w, centres, M, seed)
    check_weighted_sets(rp, col, centres, M, got, exp)


def test_oracle_append_unique_and_self_loop_equal_reference_python(oracle_mod):
    z = load("reference_py_graph_ops.npz")
    for k in range(int(z["n_append_unique"])):
        t, n = z[f"au{k}_targets"], z[f"au{k}_neighbors"]
        u, m = oracle_mod.append_unique(t, n)
        assert np.array_equal(u[: len(t)], t)                                   # targets verbatim
        assert np.array_equal(np.sort(u), z[f"au{k}_sorted_set"])                # test_graph_append_unique.py:55-59
        assert np.array_equal(u, z[f"au{k}_unique_first_appearance"])
        assert np.array_equal(m, z[f"au{k}_raw_to_unique"])                      # host_neighbor_raw_to_unique, :8-19
    for k in range(int(z["n_self_loop"])):
        orp, oc = oracle_mod.csr_add_self_loop(z[f"sl{k}_row_ptr"], z[f"sl{k}_col"])
        assert np.array_equal(orp, z[f"sl{k}_out_row_ptr"]) and np.array_equal(oc, z[f"sl{k}_out_col"])


# ----------------------------------------------------------------------------------------------- the HIP ops themselves
@pytest.mark.gpu
@pytest.mark.parametrize("case", uniform_cases())
def test_hip_uniform_equals_reference_python(hiplib, case):
    import torch
    from wholegraph_amd import wholegraph_ops as ops
    z = load("reference_py_unweighted.npz")
    rp, col, centres, M, seed, exp = unpack_uniform(z, case)
    out = ops.unweighted_sample_without_replacement(
        torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda(), torch.from_numpy(centres).cuda(), M,
        random_seed=seed, need_center_local_output=True, need_edge_output=True)
    for name, a, b in zip(("sample_offset", "dest", "center_localid", "edge_gid"), out, exp):
        a = a.cpu().numpy()
        assert a.dtype == b.dtype and np.array_equal(a, b), name


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [(1, 0), (0, 0), (1, 1)], ids=["pruned", "every_key", "redo"])
@pytest.mark.parametrize("case", weighted_cases())
def test_hip_weighted_equals_reference_python(hiplib, case, mode):
    import torch
    from wholegraph_amd import wholegraph_ops as ops
    z = load("reference_py_weighted.npz")
    rp, col, w, centres, M, seed, exp = unpack_weighted(z, case)
    hiplib.wgamd_set_weighted_sampling_mode(*mode)   # threshold-pruned (default) / every-key / exact-redo kernels
    try:
        out = ops.weighted_sample_without_replacement(
            torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda(), torch.from_numpy(w).cuda(),
            torch.from_numpy(centres).cuda(), M, random_seed=seed, need_center_local_output=True, need_edge_output=True)
    finally:
        hiplib.wgamd_set_weighted_sampling_mode(1, 0)
    check_weighted_sets(rp, col, centres, M, [t.cpu().numpy() for t in out], exp)


@pytest.mark.gpu
def test_hip_append_unique_and_self_loop_equal_reference_python(hiplib):
    import torch
    from wholegraph_amd import graph_ops
    z = load("reference_py_graph_ops.npz")
    for k in range(int(z["n_append_unique"])):
        t, n = z[f"au{k}_targets"], z[f"au{k}_neighbors"]
        u, m = graph_ops.append_unique(torch.from_numpy(t).cuda(), torch.from_numpy(n).cuda(), need_neighbor_raw_to_unique=True)
        u, m = u.cpu().numpy(), m.cpu().numpy()
        assert np.array_equal(np.sort(u), z[f"au{k}_sorted_set"]) and np.array_equal(u[: len(t)], t)
        assert np.array_equal(u, z[f"au{k}_unique_first_appearance"])
        assert np.array_equal(m, z[f"au{k}_raw_to_unique"])
    for k in range(int(z["n_self_loop"])):
        orp, oc = graph_ops.add_csr_self_loop(torch.from_numpy(z[f"sl{k}_row_ptr"]).cuda(), torch.from_numpy(z[f"sl{k}_col"]).cuda())
        assert np.array_equal(orp.cpu().numpy(), z[f"sl{k}_out_row_ptr"]) and np.array_equal(oc.cpu().numpy(), z[f"sl{k}_out_col"])
