"""Trainable embeddings + sparse optimizers (include/wgamd_embedding.h, csrc/wg_embedding.hip) against the oracle.

Shapes and tolerances follow the reference's own test
(/root/reference/cpp/tests/wholememory_ops/wholememory_embedding_gradient_apply_tests.cu): tables of dim 127 / 129 / 392 /
32 / 64, duplicate-heavy index sets, 3 (or 10) steps, every optimizer, int32 and int64 indices, fp32 / half / bf16 tables
with atol = rtol = 1e-5 / 5e-3 / 2e-2 (:759-767).  One GPU, a world_size-1 RCCL communicator; world sizes > 1 run in
tests/test_gpu_embedding_multirank.py.
"""
import numpy as np
import pytest
import torch

from oracle import embedding_optimizer as eo

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-5, torch.float16: 5e-3, torch.bfloat16: 2e-2}
ORACLE_DT = {torch.float32: "float", torch.float16: "half", torch.bfloat16: "bf16"}


@pytest.fixture(scope="module")
def comm():
    import wholegraph_amd as wg
    c = wg.create_group_communicator()
    yield c
    c.destroy()


def _table(rng, n, dim, dtype):
    t = torch.from_numpy(rng.uniform(-10, 10, (n, dim)).astype(np.float32)).to(dtype)
    return t, t.float().numpy().copy()


def _close(got, want, start, tol):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    err = np.abs(got - want)
    ok = (err <= tol) | (err <= tol * np.maximum(np.abs(got), np.abs(want)))
    assert ok.all(), "max abs err %g at %s (start value %g)" % (err.max(), np.unravel_index(err.argmax(), err.shape),
                                                                 start[np.unravel_index(err.argmax(), err.shape)])


@pytest.mark.parametrize("kind,params", [("sgd", {}), ("rmsprop", {}), ("adagrad", {}), ("lazy_adam", {}),
                                         ("lazy_adam", {"adam_w": 1.0, "weight_decay": 0.01}),
                                         ("sgd", {"weight_decay": 0.05}), ("rmsprop", {"alpha": 0.9, "epsilon": 1e-6})])
@pytest.mark.parametrize("n,dim,k,idt", [(400001, 127, 100005, torch.int64), (50021, 128, 30011, torch.int32),
                                         (1000, 129, 5000, torch.int64)])
def test_gradient_apply_matches_oracle_fp32(comm, kind, params, n, dim, k, idt):
    import wholegraph_amd as wg
    rng = np.random.default_rng(n + dim)
    emb = wg.create_embedding(comm, "distributed", "cuda", torch.float32, [n, dim])
    opt = wg.create_wholememory_optimizer(emb, kind, params)
    assert emb.get_optimizer_state_names() == eo.STATE_NAMES[kind]
    t, ref = _table(rng, n, dim, torch.float32)
    start = ref.copy()
    local, first = emb.get_embedding_tensor().get_local_tensor()
    assert first == 0 and tuple(local.shape) == (n, dim)
    local.copy_(t.cuda())
    cpu = eo.SparseOptimizer(kind, n, dim, **params)
    for step in range(3):
        idx = rng.integers(0, n, k)
        grads = rng.uniform(-5, 5, (k, dim)).astype(np.float32)
        emb.add_gradients(torch.from_numpy(idx).to(idt).cuda(), torch.from_numpy(grads).cuda())
        emb.need_apply = True
        opt.step(0.1)
        cpu.step(ref, idx, grads, 0.1)
    _close(local.cpu().numpy(), ref, start, 1e-5)
    for name, want in cpu.states.items():
        got = emb.get_optimizer_state(name).get_local_tensor()[0].cpu().numpy()
        assert got.shape == want.shape
        _close(got, want, want, 1e-5)
    # gather reads the trained rows back (forward of the next iteration)
    q = torch.from_numpy(rng.integers(0, n, 777)).cuda()
    assert torch.equal(emb.gather(q).cpu(), local.cpu()[q.cpu()])
    wg.destroy_embedding(emb)
    wg.destroy_wholememory_optimizer(opt)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kind", ["sgd", "rmsprop", "adagrad", "lazy_adam"])
@pytest.mark.parametrize("dim", [32, 64, 127])
def test_gradient_apply_low_precision_tables(comm, dtype, kind, dim):
    """Mixed precision: half / bf16 storage, fp32 states and arithmetic (the reference's FP16/BF16 cases, 500 x dim,
    400 indices)."""
    import wholegraph_amd as wg
    rng = np.random.default_rng(dim)
    n, k = 500, 400
    emb = wg.create_embedding(comm, "distributed", "cuda", dtype, [n, dim])
    opt = wg.create_wholememory_optimizer(emb, kind, {})
    t, ref = _table(rng, n, dim, dtype)
    start = ref.copy()
    local = emb.get_embedding_tensor().get_local_tensor()[0]
    assert local.dtype == dtype and local.stride(0) % (16 // t.element_size()) == 0  # rows padded to 16 bytes
    local.copy_(t.cuda())
    cpu = eo.SparseOptimizer(kind, n, dim, ORACLE_DT[dtype], )
    for step in range(3):
        idx = rng.integers(0, n, k)
        grads = rng.uniform(-5, 5, (k, dim)).astype(np.float32)
        emb.add_gradients(torch.from_numpy(idx).cuda(), torch.from_numpy(grads).cuda())
        emb.need_apply = True
        opt.step(0.1)
        cpu.step(ref, idx, grads, 0.1)
    _close(local.float().cpu().numpy(), ref, start, TOL[dtype])
    wg.destroy_embedding(emb)
    wg.destroy_wholememory_optimizer(opt)


def test_ten_steps_non_default_betas_strided_grads_and_negative_ids(comm):
    """run_count 10 with beta1 0.8 / beta2 0.9 (the reference's long cases), gradient rows with a stride of 131 behind a
    view, ids < 0 skipped, rows never touched stay bit-identical."""
    import ctypes
    import wholegraph_amd as wg
    from wholegraph_amd import _lib as L
    rng = np.random.default_rng(3)
    n, dim, k = 3000, 128, 4096
    params = {"beta1": 0.8, "beta2": 0.9}
    emb = wg.create_embedding(comm, "distributed", "cuda", torch.float32, [n, dim])
    opt = wg.create_wholememory_optimizer(emb, "lazy_adam", params)
    t, ref = _table(rng, n, dim, torch.float32)
    start = ref.copy()
    local = emb.get_embedding_tensor().get_local_tensor()[0]
    local.copy_(t.cuda())
    cpu = eo.SparseOptimizer("lazy_adam", n, dim, **params)
    touched = np.zeros(n, bool)
    for step in range(10):
        idx = rng.integers(0, n // 2, k)          # the upper half of the table is never touched
        idx[::7] = -1
        wide = torch.from_numpy(rng.uniform(-5, 5, (k, 131)).astype(np.float32)).cuda()
        grads = wide[:, :dim]                      # row stride 131
        w_i, w_g = wg.env.wrap_torch_tensor(torch.from_numpy(idx).cuda()), wg.env.wrap_torch_tensor(grads)
        L.check(L.lib().wholememory_embedding_gather_gradient_apply(emb.c_embedding, w_i.c, w_g.c, False, ctypes.c_float(0.05),
                                                                    wg.env.get_wholegraph_env_fns(), 0), "apply")
        cpu.step(ref, idx, grads.cpu().numpy(), 0.05)
        touched[idx[idx >= 0]] = True
    got = local.cpu().numpy()
    _close(got, ref, start, 1e-5)
    assert np.array_equal(got[~touched], start[~touched])
    wg.destroy_embedding(emb)
    wg.destroy_wholememory_optimizer(opt)


def test_module_autograd_path_trains_the_table(comm):
    """WholeMemoryEmbeddingModule.forward -> loss.backward() -> optimizer.step(lr) (embedding.py:220-247,578-600)."""
    import wholegraph_amd as wg
    torch.manual_seed(0)
    n, dim = 2000, 64
    emb = wg.create_embedding(comm, "distributed", "cuda", torch.float32, [n, dim], random_init=True)
    opt = wg.create_wholememory_optimizer(emb, "adagrad", {"epsilon": 1e-6})
    mod = wg.WholeMemoryEmbeddingModule(emb).train()
    local = emb.get_embedding_tensor().get_local_tensor()[0]
    before = local.clone()
    dense = torch.nn.Parameter(before.clone())
    topt = torch.optim.Adagrad([dense], lr=0.5, eps=1e-6)
    target = torch.randn(dim, device="cuda")
    for step in range(4):
        idx = torch.randint(0, n, (512,), device="cuda")
        loss = ((mod(idx) - target) ** 2).sum()
        loss.backward()
        opt.step(0.5)
        topt.zero_grad()
        ((dense[idx] - target) ** 2).sum().backward()
        topt.step()
    assert not torch.equal(local, before)
    assert torch.allclose(local, dense.detach(), rtol=1e-4, atol=1e-5)
    mod.eval()
    with torch.no_grad():
        assert torch.equal(mod(idx), local[idx])
    wg.destroy_embedding(emb)
    wg.destroy_wholememory_optimizer(opt)


def test_save_load_roundtrip_with_states(comm, tmp_path):
    import wholegraph_amd as wg
    rng = np.random.default_rng(5)
    n, dim = 1234, 33   # padded to 36 floats in memory, files hold 33 per row
    emb = wg.create_embedding(comm, "distributed", "cuda", torch.float32, [n, dim], random_init=True)
    opt = wg.create_wholememory_optimizer(emb, "lazy_adam", {})
    emb.add_gradients(torch.from_numpy(rng.integers(0, n, 900)).cuda(),
                      torch.from_numpy(rng.uniform(-1, 1, (900, dim)).astype(np.float32)).cuda())
    emb.need_apply = True
    opt.step(0.01)
    prefix = str(tmp_path / "emb")
    emb.save(prefix)
    import os
    assert os.path.getsize(prefix + "_embedding_tensor_part_0_of_1") == n * dim * 4
    assert os.path.getsize(prefix + "_beta12t_part_0_of_1") == n * 2 * 4
    emb2 = wg.create_embedding(comm, "distributed", "cuda", torch.float32, [n, dim])
    opt2 = wg.create_wholememory_optimizer(emb2, "lazy_adam", {})
    emb2.load(prefix)
    assert torch.equal(emb2.get_embedding_tensor().get_local_tensor()[0], emb.get_embedding_tensor().get_local_tensor()[0])
    for name in ("m", "v", "beta12t"):
        assert torch.equal(emb2.get_optimizer_state(name).get_local_tensor()[0],
                           emb.get_optimizer_state(name).get_local_tensor()[0]), name
    for e, o in ((emb, opt), (emb2, opt2)):
        wg.destroy_embedding(e)
        wg.destroy_wholememory_optimizer(o)


def test_error_behaviour(comm):
    import ctypes
    import wholegraph_amd as wg
    from wholegraph_amd import _lib as L
    lib = L.lib()
    c = ctypes.c_void_p()
    assert lib.wholememory_create_embedding_optimizer(ctypes.byref(c), 0) == L.WHOLEMEMORY_NOT_IMPLEMENTED
    assert lib.wholememory_create_embedding_optimizer(ctypes.byref(c), 1) == L.WHOLEMEMORY_SUCCESS   # SGD
    v = ctypes.c_float(0.5)
    assert lib.wholememory_optimizer_set_parameter(c, b"weight_decay", ctypes.byref(v)) == L.WHOLEMEMORY_SUCCESS
    assert lib.wholememory_optimizer_set_parameter(c, b"beta1", ctypes.byref(v)) == L.WHOLEMEMORY_INVALID_INPUT
    assert lib.wholememory_optimizer_set_parameter(c, b"nonsense", ctypes.byref(v)) == L.WHOLEMEMORY_INVALID_INPUT
    lib.wholememory_destroy_embedding_optimizer(c)
    pol = ctypes.c_void_p()
    # embedding.cpp:917-920: the ratio range is the only thing a policy is judged on at creation
    assert lib.wholememory_create_embedding_cache_policy(ctypes.byref(pol), comm.c_comm, 2, 2, 1,
                                                         ctypes.c_float(1.5)) == L.WHOLEMEMORY_INVALID_VALUE
    assert lib.wholememory_create_embedding_cache_policy(ctypes.byref(pol), comm.c_comm, 2, 2, 1,
                                                         ctypes.c_float(0.001)) == L.WHOLEMEMORY_INVALID_VALUE
    # a READWRITE cache whose addressing the table does not cover (cache CHUNKED over a DISTRIBUTED table) is refused
    # (embedding.cpp:968-972); the accepted combinations are tests/test_gpu_embedding_rw_cache.py
    rw = wg.create_wholememory_cache_policy(comm, memory_type="chunked", access_type="readwrite", ratio=0.5)
    with pytest.raises(L.WholeMemoryError):
        wg.create_embedding(comm, "distributed", "cuda", torch.float32, [100, 8], cache_policy=rw)
    desc = L.TensorDescription()
    lib.wholememory_initialize_tensor_desc(ctypes.byref(desc))
    desc.dim, desc.dtype = 2, L.DT_FLOAT
    desc.sizes[0], desc.sizes[1], desc.strides[0], desc.strides[1] = 100, 8, 8, 1
    e = ctypes.c_void_p()
    assert lib.wholememory_create_embedding(ctypes.byref(e), ctypes.byref(desc), comm.c_comm, L.MT_DISTRIBUTED, L.ML_DEVICE, rw.c_policy, None, -1,
                                            0) == L.WHOLEMEMORY_INVALID_INPUT
    wg.destroy_wholememory_cache_policy(rw)
    assert wg.create_builtin_cache_policy("none", "distributed", "cuda", "readonly", 0.5) is None
    with pytest.raises(ValueError):
        wg.create_builtin_cache_policy("bogus", "distributed", "cuda", "readonly", 0.5)
    emb = wg.create_embedding(comm, "distributed", "cuda", torch.float32, [100, 8])
    assert emb.get_optimizer_state_names() == []
    # no optimizer: applying gradients is a logic error, nothing is written
    emb.add_gradients(torch.zeros(4, dtype=torch.int64, device="cuda"), torch.ones((4, 8), device="cuda"))
    with pytest.raises(L.WholeMemoryError):
        emb.apply_gradients(0.1)
    opt = wg.create_wholememory_optimizer(emb, "sgd", {})
    with pytest.raises(ValueError):
        opt.add_embedding(emb)
    emb.discard_gradients()
    # wrong gradient width
    emb.add_gradients(torch.zeros(4, dtype=torch.int64, device="cuda"), torch.ones((4, 9), device="cuda"))
    with pytest.raises(L.WholeMemoryError):
        emb.apply_gradients(0.1)
    emb.writeback_all_cache()
    emb.drop_all_cache()
    with pytest.raises(L.WholeMemoryError):   # round-robin sharding is refused, not ignored
        wg.create_embedding(comm, "distributed", "cuda", torch.float32, [100, 8], round_robin_size=16)
    with pytest.raises(L.WholeMemoryError):   # integer tables cannot be trained
        e2 = wg.create_embedding(comm, "distributed", "cuda", torch.int32, [10, 4])
        wg.create_wholememory_optimizer(e2, "sgd", {})
    wg.destroy_embedding(emb)
    wg.destroy_wholememory_optimizer(opt)
