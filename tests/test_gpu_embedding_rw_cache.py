"""Host-resident tables (WHOLEMEMORY_ML_HOST) and the READWRITE device cache in front of them
(csrc/wg_comm.hip wholememory_malloc, csrc/wg_embedding.hip "READWRITE device cache").

What the reference tests for this configuration
(/root/reference/cpp/tests/wholememory_ops/wholememory_embedding_tests.cu: gather through a device cache over a host table
== host gather; wholememory_embedding_gradient_apply_tests.cu: training through the cache, then write-back, == the CPU
optimizer on the table and on every state, every optimizer, cache ratios down to a few sets) is checked here against the
same oracle as the uncached step (oracle/embedding_optimizer.py), with the same tolerances.  On top of that: the cache is
really write-back (the table is STALE until writeback_all_cache, exact after it), displaced modified lines reach the
table (cache much smaller than the working set), `adjust_cache=False` never inserts, and the policy rules of
embedding.cpp:957-1004.  World sizes > 1: ranks are threads over the in-process RCCL stand-in.
"""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from oracle import embedding_optimizer as eo

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


def _close(got, want, tol):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    err = np.abs(got - want)
    ok = (err <= tol) | (err <= tol * np.maximum(np.abs(got), np.abs(want)))
    assert ok.all(), "max abs err %g at %s" % (err.max(), np.unravel_index(err.argmax(), err.shape))


# ---- host location on its own --------------------------------------------------------------------------------------
@pytest.mark.parametrize("mtype", ["distributed", "chunked", "continuous"])
@pytest.mark.parametrize("n,dim,dtype", [(100003, 128, torch.float32), (5000, 33, torch.float16), (777, 1, torch.int64)])
def test_host_table_gather_scatter(comm, mtype, n, dim, dtype):
    """A table in pinned host memory: its local view is a CPU tensor over the bytes the GPU reads and writes in place;
    gather / scatter equal torch indexing, bit for bit (the reference's gather / scatter tests with location HOST)."""
    import wholegraph_amd as wg
    assert comm.support_type_location(mtype, "cpu")
    rng = np.random.default_rng(n)
    t = wg.create_wholememory_tensor(comm, mtype, "cpu", [n, dim], dtype, None, None)
    local, first = t.get_local_tensor()
    assert local.device.type == "cpu" and first == 0 and tuple(local.shape) == (n, dim)
    if dtype.is_floating_point:
        table = torch.from_numpy(rng.uniform(-10, 10, (n, dim)).astype(np.float32)).to(dtype)
    else:
        table = torch.from_numpy(rng.integers(-100, 100, (n, dim))).to(dtype)
    local.copy_(table)                                   # a host write; the GPU sees it (pinned, coherent)
    idx = torch.from_numpy(rng.integers(0, n, 20000)).cuda()
    assert torch.equal(t.gather(idx).cpu(), table[idx.cpu()])
    # scatter distinct rows from the GPU, read them back on the host
    rows = torch.from_numpy(rng.permutation(n)[:min(n, 3000)]).cuda()
    new = (torch.arange(rows.numel() * dim, device="cuda").reshape(-1, dim) % 97).to(dtype)
    t.scatter(new, rows)
    torch.cuda.synchronize()
    table[rows.cpu()] = new.cpu()
    assert torch.equal(local, table)
    assert torch.equal(t.gather(idx).cpu(), table[idx.cpu()])
    t.destroy()


def test_host_table_file_roundtrip_and_embedding_save_load(comm, tmp_path):
    """Binary file I/O (wholememory_store_to_file / load_from_file) on a pinned-host partition, and save / load of a
    host-resident embedding with its optimizer states after a write-back (reference embedding.py:378-407)."""
    import wholegraph_amd as wg
    n, dim = 3001, 33
    t = wg.create_wholememory_tensor(comm, "distributed", "cpu", [n, dim], torch.float32, None, None)
    local, _ = t.get_local_tensor()
    table = torch.randn(n, dim)
    local.copy_(table)
    t.to_file_prefix(str(tmp_path / "host_tensor"))
    local.zero_()
    t.from_file_prefix(str(tmp_path / "host_tensor"))
    torch.cuda.synchronize()
    assert torch.equal(local, table)
    t.destroy()
    emb, pol = _make(comm, n, dim, torch.float32, 0.25)
    opt = wg.create_wholememory_optimizer(emb, "adagrad", {})
    e_local = emb.get_embedding_tensor().get_local_tensor()[0]
    e_local.copy_(table)
    idx = torch.randint(0, n, (2000,), device="cuda")
    emb.add_gradients(idx, torch.ones(2000, dim, device="cuda"))
    emb.need_apply = True
    opt.step(0.1)
    emb.writeback_all_cache()
    trained = e_local.clone()
    state = emb.get_optimizer_state("state_sum").get_local_tensor()[0].clone()
    assert not torch.equal(trained, table) and float(state.abs().sum()) > 0
    emb.save(str(tmp_path / "host_emb"))
    emb.drop_all_cache()
    e_local.zero_()
    emb.get_optimizer_state("state_sum").get_local_tensor()[0].zero_()
    emb.load(str(tmp_path / "host_emb"))
    torch.cuda.synchronize()
    assert torch.equal(e_local, trained)
    assert torch.equal(emb.get_optimizer_state("state_sum").get_local_tensor()[0], state)
    assert torch.equal(emb.gather(idx).cpu(), trained[idx.cpu()])
    wg.destroy_embedding(emb)
    wg.destroy_wholememory_optimizer(opt)
    wg.destroy_wholememory_cache_policy(pol)


# ---- READWRITE cache: gather ---------------------------------------------------------------------------------------
def _make(comm, n, dim, dtype, ratio, location="cpu", mtype="distributed", cache_mtype="distributed"):
    import wholegraph_amd as wg
    pol = wg.create_wholememory_cache_policy(comm, memory_type=cache_mtype, memory_location="cuda",
                                             access_type="readwrite", ratio=ratio)
    emb = wg.create_embedding(comm, mtype, location, dtype, [n, dim], cache_policy=pol)
    return emb, pol


@pytest.mark.parametrize("n,dim,dtype,idt,ratio,location", [
    (100003, 128, torch.float32, torch.int64, 0.05, "cpu"),
    (5000, 127, torch.float16, torch.int32, 0.5, "cpu"),
    (20011, 1, torch.float32, torch.int64, 1.0, "cpu"),
    (3000, 100, torch.bfloat16, torch.int64, 1.0 / 512, "cpu"),     # one set
    (70000, 256, torch.float32, torch.int32, 0.1, "cuda"),          # the same cache over an HBM table
])
def test_rw_cached_gather_is_bit_exact(comm, n, dim, dtype, idt, ratio, location):
    import wholegraph_amd as wg
    rng = np.random.default_rng(n + dim)
    emb, pol = _make(comm, n, dim, dtype, ratio, location)
    table = torch.from_numpy(rng.uniform(-10, 10, (n, dim)).astype(np.float32)).to(dtype)
    local, _ = emb.get_embedding_tensor().get_local_tensor()
    assert local.device.type == location
    local.copy_(table)
    torch.cuda.synchronize()
    dev = table.cuda()
    _, _, lines = emb.cache_stats()
    assert lines % 32 == 0 and lines >= max(32, int(ratio * n))
    k = 20000
    for rnd in range(5):
        hot = (rng.pareto(1.2, k) * 3).astype(np.int64) % n
        idx = torch.from_numpy(np.where(rng.random(k) < 0.7, hot, rng.integers(0, n, k))).to(idt).cuda()
        if rnd == 3:
            idx[::9] = -1                                   # skipped rows keep what the output held
        out = torch.full((k, dim), 3.0, dtype=dtype, device="cuda")
        from wholegraph_amd import _lib as L
        from wholegraph_amd.env import get_wholegraph_env_fns, wrap_torch_tensor
        w_i, w_o = wrap_torch_tensor(idx), wrap_torch_tensor(out)
        L.check(L.lib().wholememory_embedding_gather(emb.c_embedding, w_i.c, w_o.c, True, get_wholegraph_env_fns(), 0), "gather")
        torch.cuda.synchronize()
        want = dev[idx.long().clamp(min=0)]
        want[idx < 0] = 3.0
        assert torch.equal(out, want), "round %d" % rnd
    hits, looked, _ = emb.cache_stats()
    assert 0 < hits < looked
    # a converting gather goes through the cache as well (rows travel in the table's dtype, the last copy converts)
    idx = torch.from_numpy(rng.integers(0, n, 1000)).cuda()
    if dtype == torch.float32:
        assert torch.equal(emb.gather(idx, force_dtype=torch.float16), dev[idx].half())
    assert emb.gather(idx[:0]).shape == (0, dim)
    wg.destroy_embedding(emb)
    wg.destroy_wholememory_cache_policy(pol)


def test_rw_cache_is_write_back(comm):
    """Hits are served from the lines (a host row rewritten behind the cache keeps its cached value), an SGD step on a
    resident row changes the LINE and leaves the host row stale, writeback_all_cache makes the host table exact and keeps
    the lines, drop_all_cache flushes and empties."""
    import wholegraph_amd as wg
    n, dim = 4096, 64
    emb, pol = _make(comm, n, dim, torch.float32, 1.0)
    opt = wg.create_wholememory_optimizer(emb, "sgd", {})
    table = torch.arange(n * dim, dtype=torch.float32).reshape(n, dim) / 1024
    local, _ = emb.get_embedding_tensor().get_local_tensor()
    local.copy_(table)
    idx = torch.arange(0, n, 7, device="cuda")
    assert torch.equal(emb.gather(idx).cpu(), table[idx.cpu()])        # cold: rows enter the cache
    resident = emb.cache_stats()
    assert resident[0] == 0
    local[idx.cpu()] = -1.0                                             # the host table changes behind the cache's back
    again = emb.gather(idx).cpu()
    hit = (again == table[idx.cpu()]).all(dim=1)
    miss = (again == -1.0).all(dim=1)
    assert bool((hit | miss).all()) and int(hit.sum()) == emb.cache_stats()[0] > 0.9 * idx.numel()
    local.copy_(table)
    # (a row that missed just now was READ as -1 and may have been given a line — whether it was depends on the replacement
    #  counters, i.e. on timing: about one run in fifty kept such a line and failed the checks below.  Start the training part
    #  from a cache that holds what the table holds.)
    emb.drop_all_cache()
    assert torch.equal(emb.gather(idx).cpu(), table[idx.cpu()])
    # a training step: resident rows are updated in their lines only
    grads = torch.ones(idx.numel(), dim, device="cuda")
    emb.add_gradients(idx, grads)
    emb.need_apply = True
    opt.step(0.5)
    torch.cuda.synchronize()
    want = table.clone()
    want[idx.cpu()] -= 0.5
    stale = (local[idx.cpu()] == table[idx.cpu()]).all(dim=1)
    assert int(stale.sum()) > 0.9 * idx.numel(), "resident rows must not be written through"
    assert torch.equal(emb.gather(idx).cpu(), want[idx.cpu()])          # reads see the new values whatever holds them
    emb.writeback_all_cache()
    assert torch.equal(local, want)
    h0 = emb.cache_stats()[0]
    assert torch.equal(emb.gather(idx).cpu(), want[idx.cpu()]) and emb.cache_stats()[0] - h0 > 0.9 * idx.numel()
    emb.drop_all_cache()
    assert emb.cache_stats()[:2] == (0, 0)
    emb.set_adjust_cache(False)
    assert torch.equal(emb.gather(idx).cpu(), want[idx.cpu()]) and emb.cache_stats()[0] == 0   # nothing was inserted
    wg.destroy_embedding(emb)
    wg.destroy_wholememory_optimizer(opt)
    wg.destroy_wholememory_cache_policy(pol)


# ---- READWRITE cache: training --------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,params", [("sgd", {}), ("rmsprop", {}), ("adagrad", {}), ("lazy_adam", {}),
                                         ("lazy_adam", {"adam_w": 1.0, "weight_decay": 0.01})])
@pytest.mark.parametrize("n,dim,k,idt,ratio,location", [
    (50021, 128, 30011, torch.int32, 0.02, "cpu"),      # cache of 1 k lines, 22 k distinct rows per step: constant eviction
    (1000, 129, 5000, torch.int64, 1.0, "cpu"),         # everything resident after the first step
    (20000, 127, 8000, torch.int64, 0.1, "cuda"),
])
def test_training_through_the_rw_cache_matches_oracle(comm, kind, params, n, dim, k, idt, ratio, location):
    import wholegraph_amd as wg
    rng = np.random.default_rng(n + dim)
    emb, pol = _make(comm, n, dim, torch.float32, ratio, location)
    opt = wg.create_wholememory_optimizer(emb, kind, params)
    assert emb.get_optimizer_state_names() == eo.STATE_NAMES[kind]
    ref = rng.uniform(-10, 10, (n, dim)).astype(np.float32)
    local, _ = emb.get_embedding_tensor().get_local_tensor()
    local.copy_(torch.from_numpy(ref))
    torch.cuda.synchronize()
    cpu = eo.SparseOptimizer(kind, n, dim, **params)
    for step in range(4):
        idx = rng.integers(0, n, k)
        grads = rng.uniform(-5, 5, (k, dim)).astype(np.float32)
        d_idx = torch.from_numpy(idx).to(idt).cuda()
        emb.set_adjust_cache(step != 2)                  # one step without cache adjustment: resident rows still go to lines
        fwd = emb.gather(d_idx)                          # the forward pass of the step reads through the cache
        _close(fwd.cpu().numpy(), ref[idx], 1e-5)
        emb.add_gradients(d_idx, torch.from_numpy(grads).cuda())
        emb.need_apply = True
        opt.step(0.1)
        cpu.step(ref, idx, grads, 0.1)
    q = torch.from_numpy(rng.integers(0, n, 5000)).cuda()
    _close(emb.gather(q).cpu().numpy(), ref[q.cpu().numpy()], 1e-5)
    hits, looked, _ = emb.cache_stats()
    assert 0 < hits <= looked
    emb.writeback_all_cache()
    _close(local.cpu().numpy(), ref, 1e-5)
    for name, want in cpu.states.items():
        got = emb.get_optimizer_state(name).get_local_tensor()[0].cpu().numpy()
        assert got.shape == want.shape
        _close(got, want, 1e-5)
    wg.destroy_embedding(emb)
    wg.destroy_wholememory_optimizer(opt)
    wg.destroy_wholememory_cache_policy(pol)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 5e-3), (torch.bfloat16, 2e-2)])
def test_training_low_precision_host_table(comm, dtype, tol):
    import wholegraph_amd as wg
    rng = np.random.default_rng(5)
    n, dim, k = 500, 127, 400
    emb, pol = _make(comm, n, dim, dtype, 0.25)
    opt = wg.create_wholememory_optimizer(emb, "lazy_adam", {})
    t = torch.from_numpy(rng.uniform(-10, 10, (n, dim)).astype(np.float32)).to(dtype)
    ref = t.float().numpy().copy()
    local = emb.get_embedding_tensor().get_local_tensor()[0]
    local.copy_(t)
    cpu = eo.SparseOptimizer("lazy_adam", n, dim, {torch.float16: "half", torch.bfloat16: "bf16"}[dtype])
    for step in range(3):
        idx = rng.integers(0, n, k)
        grads = rng.uniform(-5, 5, (k, dim)).astype(np.float32)
        emb.add_gradients(torch.from_numpy(idx).cuda(), torch.from_numpy(grads).cuda())
        emb.need_apply = True
        opt.step(0.1)
        cpu.step(ref, idx, grads, 0.1)
    emb.drop_all_cache()
    _close(local.float().numpy(), ref, tol)
    wg.destroy_embedding(emb)
    wg.destroy_wholememory_optimizer(opt)
    wg.destroy_wholememory_cache_policy(pol)


def test_rw_policy_rules(comm):
    import wholegraph_amd as wg
    from wholegraph_amd import _lib as L
    # embedding.cpp:1000-1004: READWRITE only on the table's own communicator
    other = wg.get_local_device_communicator()
    assert other.c_comm.value != comm.c_comm.value
    pol = wg.create_wholememory_cache_policy(other, memory_type="chunked", access_type="readwrite", ratio=0.5)
    with pytest.raises(L.WholeMemoryError):
        wg.create_embedding(comm, "distributed", "cpu", torch.float32, [100, 8], cache_policy=pol)
    wg.destroy_wholememory_cache_policy(pol)
    # embedding.cpp:962-967: the cache itself lives on the device
    pol = wg.create_wholememory_cache_policy(comm, memory_type="distributed", memory_location="cpu", access_type="readwrite", ratio=0.5)
    with pytest.raises(L.WholeMemoryError):
        wg.create_embedding(comm, "distributed", "cpu", torch.float32, [100, 8], cache_policy=pol)
    wg.destroy_wholememory_cache_policy(pol)
    # embedding.cpp:968-972: the table's addressing must cover the cache's (continuous < chunked < distributed)
    pol = wg.create_wholememory_cache_policy(comm, memory_type="continuous", access_type="readwrite", ratio=0.5)
    with pytest.raises(L.WholeMemoryError):
        wg.create_embedding(comm, "distributed", "cpu", torch.float32, [100, 8], cache_policy=pol)
    wg.destroy_wholememory_cache_policy(pol)
    # the builtin flavour of the reference's examples: all_devices + cpu table + readwrite
    pol = wg.create_builtin_cache_policy("all_devices", "distributed", "cpu", "readwrite", 0.25)
    emb = wg.create_embedding(wg.get_global_communicator(), "distributed", "cpu", torch.float32, [1000, 16], cache_policy=pol)
    opt = wg.create_wholememory_optimizer(emb, "adagrad", {})           # trainable, unlike a READONLY cache
    assert emb.get_optimizer_state("state_sum").get_local_tensor()[0].device.type == "cpu"
    wg.destroy_embedding(emb)
    wg.destroy_wholememory_optimizer(opt)
    wg.destroy_wholememory_cache_policy(pol)


# ---- world sizes > 1 ------------------------------------------------------------------------------------------------
WORKER = textwrap.dedent(r"""
    import ctypes, sys, threading
    import numpy as np, torch
    sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/cugraph-gnn_amd")
    import wholegraph_amd as wg
    from wholegraph_amd import _lib as L
    from wholegraph_amd.comm import WholeMemoryCommunicator
    from oracle import embedding_optimizer as eo
    W, n, dim = (int(v) for v in sys.argv[2:5])
    mtype, ratio, kind = sys.argv[5], float(sys.argv[6]), sys.argv[7]
    lib = L.lib()
    uid = L.UniqueId()
    L.check(lib.wholememory_create_unique_id(ctypes.byref(uid)), "uid")
    rng0 = np.random.default_rng(W + n)
    ref = rng0.uniform(-10, 10, (n, dim)).astype(np.float32)
    start = ref.copy()
    STEPS = 3
    # every rank's (ids, grads) of every step, known to all: the oracle applies the union in rank order
    plan = [[(rng0.integers(0, n, 2000 + 300 * r if (s + r) % 3 else 0), None) for r in range(W)] for s in range(STEPS)]
    plan = [[(ids, rng0.uniform(-5, 5, (len(ids), dim)).astype(np.float32)) for ids, _ in row] for row in plan]
    cpu = eo.SparseOptimizer(kind, n, dim)
    snapshots = []
    for s in range(STEPS):
        snapshots.append(ref.copy())
        cpu.step(ref, np.concatenate([p[0] for p in plan[s]]), np.concatenate([p[1] for p in plan[s]]), 0.1)
    results = [None] * W
    gate = threading.Barrier(W)

    def close(got, want, tol=1e-5):
        err = np.abs(got - want)
        assert ((err <= tol) | (err <= tol * np.maximum(np.abs(got), np.abs(want)))).all(), err.max()

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            c = ctypes.c_void_p()
            L.check(lib.wholememory_create_communicator(ctypes.byref(c), uid, r, W), "create_communicator")
            comm = WholeMemoryCommunicator(c.value)
            assert comm.support_type_location(mtype, "cpu")
            pol = wg.create_wholememory_cache_policy(comm, memory_type=mtype, access_type="readwrite", ratio=ratio)
            emb = wg.create_embedding(comm, mtype, "cpu", torch.float32, [n, dim], cache_policy=pol)
            opt = wg.create_wholememory_optimizer(emb, kind, {})
            local, first = emb.get_embedding_tensor().get_local_tensor()
            assert local.device.type == "cpu"
            local.copy_(torch.from_numpy(start[first:first + local.shape[0]]))
            comm.barrier()
            for s in range(STEPS):
                ids, grads = plan[s][r]
                d_ids = torch.from_numpy(ids).cuda()
                fwd = emb.gather(d_ids)
                close(fwd.cpu().numpy(), snapshots[s][ids])
                d_grads = torch.from_numpy(grads).cuda() if len(ids) else torch.empty((0, dim), device="cuda")
                emb.add_gradients(d_ids, d_grads)
                emb.need_apply = True
                opt.step(0.1)
            q = np.random.default_rng(r).integers(0, n, 3000)
            close(emb.gather(torch.from_numpy(q).cuda()).cpu().numpy(), ref[q])
            hits, looked, lines = emb.cache_stats()
            assert looked > 0 and hits > 0, (hits, looked)
            emb.writeback_all_cache()
            close(local.numpy(), ref[first:first + local.shape[0]])
            for name, want in cpu.states.items():
                st, f0 = emb.get_optimizer_state(name).get_local_tensor()
                close(st.numpy(), want[f0:f0 + st.shape[0]])
            comm.barrier()
            wg.destroy_embedding(emb)
            wg.destroy_wholememory_optimizer(opt)
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


@pytest.mark.parametrize("W,n,dim,mtype,ratio,kind", [
    (2, 20000, 64, "distributed", 0.1, "lazy_adam"),
    (3, 9001, 100, "distributed", 0.5, "adagrad"),
    (4, 30000, 32, "chunked", 0.05, "sgd"),
    (2, 4000, 128, "continuous", 1.0, "rmsprop"),
    (3, 12000, 64, "chunked+shm", 0.2, "lazy_adam"),    # thread ranks map each other's segments the way processes do
])
def test_rw_cache_world_gt1(shim, W, n, dim, mtype, ratio, kind):
    env = dict(os.environ, WGAMD_RCCL_LIBRARY=shim)
    if mtype.endswith("+shm"):
        mtype, env["WGAMD_HOST_SHM_MAP_ALWAYS"] = mtype[:-4], "1"
    p = subprocess.run([sys.executable, "-c", WORKER, ROOT, str(W), str(n), str(dim), mtype, str(ratio), kind],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ALL_RANKS_OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
