"""GPU parity: append_unique, csr_add_self_loop, gather/scatter, multilayer walk (sync + no-sync)."""
import numpy as np
import pytest

from graphgen import powerlaw_csr, random_csr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,E,space", [(4, 8, 12), (10, 100, 64), (1000, 30000, 5000), (0, 100, 50),
                                      (100, 0, 50), (26624, 266240, 2449029), (1, 1, 1)])
@pytest.mark.parametrize("dtype", [np.int32, np.int64])
def test_append_unique(oracle_mod, hiplib, T, E, space, dtype):
    import torch
    from wholegraph_amd import graph_ops
    rng = np.random.default_rng(T + E)
    targets = rng.permutation(max(space, T))[:T].astype(dtype)
    nbrs = rng.integers(0, space, E).astype(dtype)
    uniq, mp = graph_ops.append_unique(torch.from_numpy(targets).cuda(), torch.from_numpy(nbrs).cuda(), True)
    ouniq, omp = oracle_mod.append_unique(targets, nbrs)
    uniq, mp = uniq.cpu().numpy(), mp.cpu().numpy()
    # the reference's own (weaker) checks: targets verbatim, tail equal as a set, mapping consistent
    # (cpp/tests/graph_ops/append_unique_tests.cu:160-199)
    assert np.array_equal(uniq[:T], targets)
    assert np.array_equal(np.sort(uniq[T:]), np.sort(ouniq[T:]))
    assert np.array_equal(uniq[mp], nbrs)
    # the build's stronger contract: first-appearance order == host oracle order, bit-exact
    assert np.array_equal(uniq, ouniq) and np.array_equal(mp, omp)
    only = graph_ops.append_unique(torch.from_numpy(targets).cuda(), torch.from_numpy(nbrs).cuda())
    assert np.array_equal(only.cpu().numpy(), ouniq)


def test_append_unique_docstring_example(hiplib):
    # python/pylibwholegraph/pylibwholegraph/torch/graph_ops.py:21-29
    import torch
    from wholegraph_amd import graph_ops
    t = torch.tensor([3, 11, 2, 10], device="cuda")
    n = torch.tensor([4, 5, 2, 11, 6, 9, 10, 5], device="cuda")
    u, m = graph_ops.append_unique(t, n, True)
    assert u[:4].tolist() == [3, 11, 2, 10] and sorted(u[4:].tolist()) == [4, 5, 6, 9]
    assert u[m.long()].tolist() == n.tolist()


def test_csr_add_self_loop(oracle_mod, hiplib):
    import torch
    from wholegraph_amd import graph_ops
    rng = np.random.default_rng(0)
    deg = rng.integers(0, 90, 777)
    rp = np.zeros(778, np.int32)
    rp[1:] = np.cumsum(deg)
    col = rng.integers(0, 5000, rp[-1]).astype(np.int32)
    a, b = graph_ops.add_csr_self_loop(torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda())
    oa, ob = oracle_mod.csr_add_self_loop(rp, col)
    assert np.array_equal(a.cpu().numpy(), oa) and np.array_equal(b.cpu().numpy(), ob)


def _torch_dt(name):
    import torch
    return getattr(torch, name)


# the reference gtest matrix: dims incl. odd / strided, dtype pairs, idx types, counts incl. 0
# (cpp/tests/wholememory_ops/wholememory_gather_tests.cu:106-116,282-412)
@pytest.mark.parametrize("dim,stride", [(1, 1), (11, 12), (32, 32), (100, 100), (127, 127), (128, 128),
                                        (129, 129), (256, 256), (513, 513), (1024, 1024)])
@pytest.mark.parametrize("tdt,odt", [("float32", "float32"), ("float16", "float16"), ("float32", "float16"),
                                     ("float16", "float32"), ("bfloat16", "float32"), ("int64", "int64"),
                                     ("int8", "int8"), ("float64", "float32"), ("int32", "int64")])
@pytest.mark.parametrize("idt,count", [("int32", 100005), ("int64", 1000), ("int64", 0)])
def test_gather_matrix(hiplib, dim, stride, tdt, odt, idt, count):
    import torch
    from wholegraph_amd import WholeMemoryTensor
    rows = 20011
    g = torch.Generator().manual_seed(dim * 7 + count)
    # KAT of the reference: table[i, j] = (i + j) masked to what the narrow type holds exactly
    base = (torch.arange(rows).view(-1, 1) + torch.arange(stride).view(1, -1)) % 120
    storage = base.to(_torch_dt(tdt)).cuda()
    table = storage[:, :dim]  # stride >= dim
    idx = torch.randint(0, rows, (count,), generator=g).to(_torch_dt(idt))
    if count > 10:
        idx[3] = -1  # skipped row keeps its old content
    out = WholeMemoryTensor(table).gather(idx.cuda(), force_dtype=_torch_dt(odt))
    ref = base[:, :dim][idx.clamp(min=0).long()].to(_torch_dt(odt))
    got = out.cpu()
    if count > 10:
        got[3] = ref[3]
    assert got.shape == (count, dim)
    assert torch.equal(got, ref)


def test_gather_1d_and_scatter_roundtrip(hiplib):
    import torch
    from wholegraph_amd import WholeMemoryTensor
    t1 = torch.arange(50000, dtype=torch.int64).cuda() * 3
    idx = torch.randperm(50000)[:9999].cuda()
    assert torch.equal(WholeMemoryTensor(t1).gather(idx), t1[idx])
    table = torch.zeros((4096, 100), dtype=torch.float32, device="cuda")
    wt = WholeMemoryTensor(table)
    rows = torch.randn(1000, 100, device="cuda")
    where = torch.randperm(4096)[:1000].cuda()
    wt.scatter(rows, where)
    assert torch.equal(table[where], rows)                 # scatter then gather == identity
    assert torch.equal(wt.gather(where), rows)
    untouched = torch.ones(4096, dtype=torch.bool, device="cuda")
    untouched[where] = False
    assert float(table[untouched].abs().max()) == 0.0       # nothing written outside the scattered rows


@pytest.mark.parametrize("col_dtype", [np.int32, np.int64])
@pytest.mark.parametrize("fanouts", [[5, 5], [25, 10], [15, 10, 5]])
def test_multilayer_walk_sync_and_nosync(oracle_mod, hiplib, col_dtype, fanouts):
    import torch
    from wholegraph_amd import GraphStructure
    row_ptr, col = powerlaw_csr(30000, 20, seed=11, col_dtype=col_dtype, max_deg=4000)
    seeds = np.random.default_rng(5).permutation(30000)[:256].astype(col_dtype)
    rs = [62 + k for k in range(len(fanouts))]
    g = GraphStructure()
    g.set_csr_graph(torch.from_numpy(row_ptr).cuda(), torch.from_numpy(col).cuda())
    otg, oei, orp, oci = oracle_mod.multilayer_sample(row_ptr, col, seeds, fanouts, rs)
    seeds_d = torch.from_numpy(seeds).cuda()
    captured = g.multilayer_sample_without_replacement(seeds_d, fanouts, random_seeds=rs)   # HIP-graph replay of the walk
    assert g._captured_ok and any(k[0] == "captured" for k in g._walk_cache if isinstance(k[0], str))
    again = g.multilayer_sample_without_replacement(seeds_d, fanouts, random_seeds=[r + 1000 for r in rs])   # replays reuse buffers:
    assert not torch.equal(again[0][0], captured[0][0])                               # ... the first result must be its own copy
    g._captured_ok = False                                                            # the op-by-op loop over the C-ABI ops
    op_by_op = g.multilayer_sample_without_replacement(seeds_d, fanouts, random_seeds=rs)
    g._captured_ok = True
    for result in (captured, op_by_op,
                   g.multilayer_sample_nosync(torch.from_numpy(seeds).cuda(), fanouts, random_seeds=rs).finalize()):
        tg, ei, rp, ci = result
        for name, got, want in (("target_gids", tg, otg), ("edge_indice", ei, oei), ("csr_row_ptr", rp, orp),
                                ("csr_col_ind", ci, oci)):
            for lvl, (a, b) in enumerate(zip(got, want)):
                assert np.array_equal(a.cpu().numpy(), b), (name, lvl)
    # structural invariants of the walk (graph_structure.py:186-195): seeds first, CSR rows match
    assert np.array_equal(tg[0][: len(seeds)].cpu().numpy(), seeds)
    for i in range(len(fanouts)):
        assert rp[i].shape[0] == tg[i + 1].shape[0] + 1 and int(ci[i].max()) < tg[i].shape[0]


def test_walk_is_identical_with_the_single_launch_scan(hiplib):
    """WGAMD_SCAN_CHAINED=1 swaps every multi-tile exclusive scan for the single-pass (decoupled look-back) kernel of
    wg_scan_chain.hpp — slower on gfx950 (wg_scan.hip header) and therefore off by default, but it must stay correct: the
    walk / call-group / sampling parity tests pass under it unchanged."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_callgroup.py",
                        "tests/test_gpu_renumber_gather.py", "-k", "not single_launch_scan"],
                       cwd=root, env=dict(os.environ, WGAMD_SCAN_CHAINED="1"), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]


@pytest.mark.parametrize("n,bound,dtype", [(0, 10, np.int64), (1, 1, np.int64), (5000, 300, np.int32), (300, 5000, np.int64),
                                           (200000, 70001, np.int64), (4097, 4096, np.int32)])
def test_unique_bounded_matches_numpy(oracle_mod, hiplib, n, bound, dtype):
    """wgamd_unique_bounded: distinct ids ascending + the position of every id (mark / scan / compact / look up)."""
    import torch
    from wholegraph_amd.tensor import unique_bounded
    rng = np.random.default_rng(n + bound)
    ids = (rng.zipf(1.3, n) % bound).astype(dtype) if n else np.zeros(0, dtype)
    if n > 10:
        ids[::7] = -1          # rows to skip
        ids[1] = bound - 1     # the largest legal id
        ids[2] = 0
    want_d, want_i = oracle_mod.unique_bounded(ids, bound)
    d, inv = unique_bounded(torch.from_numpy(ids).cuda(), bound)
    assert d.dtype == torch.int64 and inv.dtype == torch.int32
    assert np.array_equal(d.cpu().numpy(), want_d) and np.array_equal(inv.cpu().numpy(), want_i)
    if n > 10:
        bad = ids.copy()
        bad[3] = bound
        with pytest.raises(IndexError):
            unique_bounded(torch.from_numpy(bad).cuda(), bound)


def test_unique_bounded_refuses_an_unsupported_bound(hiplib):
    import torch
    from wholegraph_amd.tensor import unique_bounded
    with pytest.raises(ValueError):
        unique_bounded(torch.zeros(4, dtype=torch.int64, device="cuda"), 1 << 31)
