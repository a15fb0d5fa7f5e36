"""READONLY embedding cache (include/wgamd_embedding.h, csrc/wg_embedding.hip: lookup / hit copy / miss fetch / insert).

The property the reference's own test checks (/root/reference/cpp/tests/wholememory_ops/wholememory_embedding_tests.cu:
gather through every cache configuration == host gather of the table) is bit-exact here: a cache line is a byte copy of a
table row.  On top of that the tests prove that hits really are served from the cache lines (a table row rewritten behind
the cache's back keeps its old value until `drop_all_cache`), that `adjust_cache=False` inserts nothing, and the error
behaviour of embedding.cpp:55-60 / :917-920 / :986-1003.  World sizes > 1: ranks are threads over the in-process RCCL
stand-in (tests/test_gpu_comm_multirank.py), table DISTRIBUTED and CHUNKED.
"""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_DIR = os.path.join(ROOT, "tests", "shim")
SHIM = os.path.join(SHIM_DIR, "build", "libfake_rccl.so")


@pytest.fixture(scope="module")
def comm():
    import wholegraph_amd as wg
    c = wg.create_group_communicator()
    yield c
    c.destroy()


def _skewed(rng, n, k):
    """Power-law row popularity: a few rows are asked for again and again (what a cache is for)."""
    hot = (rng.pareto(1.2, k) * 3).astype(np.int64) % n
    cold = rng.integers(0, n, k)
    return np.where(rng.random(k) < 0.7, hot, cold)


def _make(comm, n, dim, dtype, ratio, mtype="distributed", policy_mtype="chunked"):
    import wholegraph_amd as wg
    pol = wg.create_wholememory_cache_policy(comm, memory_type=policy_mtype, memory_location="cuda",
                                             access_type="readonly", ratio=ratio)
    emb = wg.create_embedding(comm, mtype, "cuda", dtype, [n, dim], cache_policy=pol)
    return emb, pol


def _fill(emb, table):
    local, first = emb.get_embedding_tensor().get_local_tensor()
    local.copy_(table[first:first + local.shape[0]].to(local.device))
    torch.cuda.synchronize()
    return local


@pytest.mark.parametrize("n,dim,dtype,idt,ratio", [
    (100003, 128, torch.float32, torch.int64, 0.05),
    (5000, 127, torch.float16, torch.int32, 0.5),      # odd half rows: 2-byte copies
    (20011, 1, torch.float32, torch.int64, 1.0),       # one element per row
    (3000, 100, torch.bfloat16, torch.int64, 1.0 / 512),  # 5 lines asked for -> one set
    (70000, 256, torch.int64, torch.int32, 0.1),       # non-floating rows are cached the same way
    (40, 33, torch.int8, torch.int64, 1.0),            # 33-byte rows: byte copies
])
def test_cached_gather_is_bit_exact(comm, n, dim, dtype, idt, ratio):
    import wholegraph_amd as wg
    rng = np.random.default_rng(n + dim)
    emb, pol = _make(comm, n, dim, dtype, ratio)
    if dtype.is_floating_point:
        table = torch.from_numpy(rng.uniform(-10, 10, (n, dim)).astype(np.float32)).to(dtype)
    else:
        table = torch.from_numpy(rng.integers(-100, 100, (n, dim))).to(dtype)
    _fill(emb, table)
    dev_table = table.cuda()
    assert emb.adjust_cache is True                      # embedding.py:290
    h0, l0, lines = emb.cache_stats()
    assert (h0, l0) == (0, 0) and lines % 32 == 0 and lines >= max(32, int(ratio * n))
    k = 20000
    for rnd in range(6):
        idx = torch.from_numpy(_skewed(rng, n, k)).to(idt).cuda()
        out = emb.gather(idx)
        assert out.dtype == dtype and torch.equal(out, dev_table[idx.long()]), "round %d" % rnd
    hits, looked, _ = emb.cache_stats()
    assert looked == 6 * k and 0 < hits < looked
    # the same ids again: every row that found a line is a hit now
    before = hits
    out = emb.gather(idx)
    assert torch.equal(out, dev_table[idx.long()])
    hits, looked, _ = emb.cache_stats()
    assert hits - before > 0
    if ratio == 1.0 and n > 1000:
        assert hits - before > 0.9 * k, "a cache as big as the table holds (almost) every row asked twice"
    wg.destroy_embedding(emb)
    wg.destroy_wholememory_cache_policy(pol)


def test_hits_are_served_from_the_cache_lines_until_dropped(comm):
    import wholegraph_amd as wg
    n, dim = 4096, 64
    emb, pol = _make(comm, n, dim, torch.float32, 1.0)
    table = torch.arange(n * dim, dtype=torch.float32).reshape(n, dim)
    local = _fill(emb, table)
    idx = torch.arange(0, n, 7, device="cuda")
    first = emb.gather(idx)                      # cold: everything misses, rows are inserted
    assert torch.equal(first, table.cuda()[idx])
    h, looked, _ = emb.cache_stats()
    assert h == 0 and looked == idx.numel()
    local.mul_(-1.0)                             # the table changes behind the READONLY cache's back
    torch.cuda.synchronize()
    second = emb.gather(idx)
    h, _, _ = emb.cache_stats()
    cached = (second == first).all(dim=1)
    fresh = (second == -first).all(dim=1)
    assert bool((cached | fresh).all()) and int(cached.sum()) == h > 0.9 * idx.numel()
    emb.writeback_all_cache()                    # nothing is ever dirty: a no-op
    emb.drop_all_cache()
    assert emb.cache_stats()[:2] == (0, 0)
    third = emb.gather(idx)
    assert torch.equal(third, -first) and emb.cache_stats()[0] == 0
    wg.destroy_embedding(emb)
    wg.destroy_wholememory_cache_policy(pol)


def test_adjust_cache_false_reads_through_without_inserting(comm):
    import wholegraph_amd as wg
    n, dim = 10000, 32
    emb, pol = _make(comm, n, dim, torch.float32, 0.5)
    table = torch.randn(n, dim)
    _fill(emb, table)
    idx = torch.randint(0, n, (5000,), device="cuda")
    emb.set_adjust_cache(False)
    for _ in range(3):
        assert torch.equal(emb.gather(idx), table.cuda()[idx])
    assert emb.cache_stats()[0] == 0
    emb.set_adjust_cache(True)
    assert torch.equal(emb.gather(idx), table.cuda()[idx])       # inserts
    emb.set_adjust_cache(False)
    assert torch.equal(emb.gather(idx), table.cuda()[idx])       # hits, counts untouched
    assert emb.cache_stats()[0] > 0.5 * idx.numel()          # 157 sets of 32 lines for ~3900 distinct rows: some sets overflow
    # a converting gather bypasses the cache (a line is a byte copy of a table row) and still answers
    h = emb.cache_stats()[0]
    out = emb.gather(idx, force_dtype=torch.float16)
    assert torch.equal(out, table.cuda()[idx].half()) and emb.cache_stats()[0] == h
    # empty request
    assert emb.gather(idx[:0]).shape == (0, dim)
    wg.destroy_embedding(emb)
    wg.destroy_wholememory_cache_policy(pol)


def test_skipped_and_strided_rows(comm):
    """Negative ids leave their output rows untouched, through the cache as without it; output rows may be strided."""
    import ctypes
    import wholegraph_amd as wg
    from wholegraph_amd import _lib as L
    from wholegraph_amd.env import get_wholegraph_env_fns, wrap_torch_tensor
    n, dim = 3000, 24
    emb, pol = _make(comm, n, dim, torch.float32, 1.0)
    table = torch.randn(n, dim)
    _fill(emb, table)
    idx = torch.randint(0, n, (4000,), device="cuda")
    idx[::5] = -1
    for rnd in range(3):
        wide = torch.full((idx.numel(), dim + 8), 7.0, device="cuda")
        out = wide[:, 4:4 + dim]                                  # row stride dim + 8, 16-byte misaligned start
        w_i, w_o = wrap_torch_tensor(idx), wrap_torch_tensor(out)
        L.check(L.lib().wholememory_embedding_gather(emb.c_embedding, w_i.c, w_o.c, True, get_wholegraph_env_fns(), 0),
                "gather")
        torch.cuda.synchronize()
        want = table.cuda()[idx.clamp(min=0)]
        want[idx < 0] = 7.0
        assert torch.equal(out, want), "round %d" % rnd
        assert bool((wide[:, :4] == 7.0).all()) and bool((wide[:, 4 + dim:] == 7.0).all())
    assert emb.cache_stats()[0] > 0
    wg.destroy_embedding(emb)
    wg.destroy_wholememory_cache_policy(pol)


def test_policy_rules(comm):
    import wholegraph_amd as wg
    from wholegraph_amd import _lib as L
    with pytest.raises(L.WholeMemoryError):      # embedding.cpp:917-920
        wg.create_wholememory_cache_policy(comm, ratio=2.0)
    with pytest.raises(L.WholeMemoryError):
        wg.create_wholememory_cache_policy(comm, ratio=1.0 / 1024)
    # the built-in flavours (embedding.py:124-216)
    for kind in ("local_device", "local_node", "all_devices"):
        pol = wg.create_builtin_cache_policy(kind, "distributed", "cuda", "readonly", 0.25)
        # "all_devices" puts a DISTRIBUTED cache on the global communicator: the table must live on that one too
        table_comm = wg.get_global_communicator() if kind == "all_devices" else comm
        emb = wg.create_embedding(table_comm, "distributed", "cuda", torch.float32, [1000, 16], cache_policy=pol)
        t = torch.randn(1000, 16)
        _fill(emb, t)
        idx = torch.randint(0, 1000, (512,), device="cuda")
        assert torch.equal(emb.gather(idx), t.cuda()[idx]) and torch.equal(emb.gather(idx), t.cuda()[idx])
        assert emb.cache_stats()[0] > 0
        # embedding.cpp:55-60: no optimizer on a local cached global readonly embedding
        with pytest.raises(L.WholeMemoryError):
            wg.create_wholememory_optimizer(emb, "sgd", {})
        wg.destroy_embedding(emb)
        wg.destroy_wholememory_cache_policy(pol)
    assert wg.create_builtin_cache_policy("none", "distributed", "cuda", "readonly", 0.5) is None
    with pytest.raises(ValueError):
        wg.create_builtin_cache_policy("local_device", "bogus", "cuda", "readonly", 0.5)
    # embedding.cpp:986-992: a cache on ANOTHER communicator cannot be DISTRIBUTED
    other = wg.get_local_device_communicator()
    pol = wg.create_wholememory_cache_policy(other, memory_type="distributed", access_type="readonly", ratio=0.5)
    with pytest.raises(L.WholeMemoryError):
        wg.create_embedding(comm, "distributed", "cuda", torch.float32, [100, 8], cache_policy=pol)
    wg.destroy_wholememory_cache_policy(pol)
    wg.destroy_wholememory_cache_policy(None)


WORKER = textwrap.dedent(r"""
    import ctypes, sys, threading
    import numpy as np, torch
    sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/cugraph-gnn_amd")
    import wholegraph_amd as wg
    from wholegraph_amd import _lib as L
    from wholegraph_amd.comm import WholeMemoryCommunicator
    W, n, dim = (int(v) for v in sys.argv[2:5])
    mtype, ratio = sys.argv[5], float(sys.argv[6])
    lib = L.lib()
    uid = L.UniqueId()
    L.check(lib.wholememory_create_unique_id(ctypes.byref(uid)), "uid")
    table = torch.from_numpy(np.random.default_rng(W + n).uniform(-10, 10, (n, dim)).astype(np.float32))
    results = [None] * W

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            c = ctypes.c_void_p()
            L.check(lib.wholememory_create_communicator(ctypes.byref(c), uid, r, W), "create_communicator")
            comm = WholeMemoryCommunicator(c.value)
            pol = wg.create_wholememory_cache_policy(comm, memory_type="chunked", access_type="readonly", ratio=ratio)
            emb = wg.create_embedding(comm, mtype, "cuda", torch.float32, [n, dim], cache_policy=pol)
            local, first = emb.get_embedding_tensor().get_local_tensor()
            local.copy_(table[first:first + local.shape[0]].cuda())
            torch.cuda.synchronize()
            comm.barrier()
            dev = table.cuda()
            rng = np.random.default_rng(1000 + r)
            for rnd in range(5):
                k = 3000 + 500 * r if (rnd + r) % 3 else 0          # ranks bring different amounts, sometimes nothing
                hot = (rng.pareto(1.2, k) * 3).astype(np.int64) % n
                idx = torch.from_numpy(np.where(rng.random(k) < 0.7, hot, rng.integers(0, n, k))).cuda()
                out = emb.gather(idx)
                assert torch.equal(out, dev[idx]), f"rank {r} round {rnd}"
            hits, looked, lines = emb.cache_stats()
            assert 0 < hits < looked, (hits, looked)
            emb.drop_all_cache()
            assert emb.cache_stats()[:2] == (0, 0)
            idx = torch.from_numpy(rng.integers(0, n, 1000)).cuda()
            assert torch.equal(emb.gather(idx), dev[idx])
            comm.barrier()
            wg.destroy_embedding(emb)
            wg.destroy_wholememory_cache_policy(pol)
            comm.destroy()
            results[r] = "ok"
        except BaseException as e:  # noqa
            import traceback; traceback.print_exc()
            print("FAILED rank", r, repr(e)); sys.stdout.flush(); sys.stderr.flush()
            import os; os._exit(1)

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(W)]
    for th in threads: th.start()
    for th in threads: th.join(240)
    alive = [i for i, th in enumerate(threads) if th.is_alive()]
    if alive or any(v != "ok" for v in results):
        print("FAILED", alive, results); sys.stdout.flush()
        import os; os._exit(1)
    print("ALL_RANKS_OK")
""")


@pytest.fixture(scope="module")
def shim():
    if not os.path.exists(SHIM):
        subprocess.run(["make", "-C", SHIM_DIR, "-s"], check=True)
    return SHIM


@pytest.mark.parametrize("W,n,dim,mtype,ratio", [
    (2, 20000, 64, "distributed", 0.1),
    (3, 9001, 100, "distributed", 0.5),
    (4, 50000, 32, "chunked", 0.05),
    (2, 4000, 128, "continuous", 1.0),
])
def test_cached_gather_world_gt1(shim, W, n, dim, mtype, ratio):
    env = dict(os.environ, WGAMD_RCCL_LIBRARY=shim)
    p = subprocess.run([sys.executable, "-c", WORKER, ROOT, str(W), str(n), str(dim), mtype, str(ratio)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ALL_RANKS_OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
