"""cugraph_pyg_amd.tensor on the GPU (one rank: the HIP row kernels behind ``__getitem__`` / ``__setitem__``), mirrored
from the reference's tests/tensor/test_dist_tensor_mg.py and test_dist_matrix_mg.py; worlds of 2 run over gloo in
tests/test_dist_gloo.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("device", ["cpu", "cuda"])
@pytest.mark.parametrize("clx_name", ["DistTensor", "DistEmbedding"])
def test_dist_tensor_creation_and_files(tmp_path, clx_name, device, dtype):
    import cugraph_pyg_amd.tensor as T
    clx = getattr(T, clx_name)
    features = torch.randn(100 * 10, dtype=torch.float32, device="cuda").to(dtype).reshape((-1, 10)).to(device)
    t = clx.from_tensor(tensor=features, device=device)
    assert t.shape == features.shape and t.dtype == features.dtype and t.device == device and t.dim == 2
    ix = torch.randint(0, features.shape[0], (10,))
    out = t[ix]
    assert out.is_cuda and torch.equal(features[ix].cuda(), out)
    full = torch.arange(0, 1000).reshape((10, 100)).to(dtype)
    pt, npy = str(tmp_path / "f.pt"), str(tmp_path / "f.npy")
    torch.save(full, pt)
    t = clx.from_file(pt, device=device)
    assert t.shape == full.shape and t.dtype == full.dtype and torch.equal(t[torch.arange(10)].cpu(), full)
    if dtype != torch.bfloat16:      # numpy has no bfloat16
        np.save(npy, full.numpy())
        t = clx.from_file(npy, device=device)
        assert torch.equal(t[torch.tensor([9, 0, 9])].cpu(), full[[9, 0, 9]])
    parts = [str(tmp_path / ("p_part_%d_of_2" % i)) for i in range(2)]
    src = full.view(torch.int16) if dtype == torch.bfloat16 else full
    src[:4].numpy().tofile(parts[0])
    src[4:].numpy().tofile(parts[1])
    t = clx(src=parts, shape=[10, 100], dtype=dtype)
    assert torch.equal(t.get_local_tensor().cpu(), full)
    t[torch.tensor([1, 3])] = torch.ones((2, 100))       # value converted to the table dtype
    assert bool((t[torch.tensor([3])] == 1).all()) and torch.equal(t[torch.tensor([2])].cpu(), full[2:3])


def test_host_pinned_rows_are_read_and_written_in_place():
    """device="cpu" / FeatureStore(location="cpu") — the reference's default placement (feature_store.py:42-58,
    dist_tensor.py:60-75): the rows live in pinned host memory, indices and results on the GPU, and the HIP row kernels read
    and write the host rows in place."""
    import cugraph_pyg_amd.tensor as T
    from cugraph_pyg_amd.data import FeatureStore
    feats = torch.randn(5000, 64)
    t = T.DistEmbedding.from_tensor(feats, device="cpu")
    local = t.get_local_tensor()
    assert local.device.type == "cpu" and local.is_pinned() and t.device == "cpu"
    ix = torch.randint(0, 5000, (2000,), device="cuda")
    out = t[ix]
    assert out.is_cuda and torch.equal(out.cpu(), feats[ix.cpu()])
    local[7] = 5.0                                            # a host write: the next gather sees it
    assert bool((t[torch.tensor([7])] == 5.0).all())
    t[torch.tensor([9, 11])] = torch.full((2, 64), 3.0)       # a GPU scatter into the host rows
    torch.cuda.synchronize()
    assert bool((local[[9, 11]] == 3.0).all())
    ids = T.DistTensor(shape=[1000], dtype=torch.int64, device="cpu")
    ids[torch.arange(1000)] = torch.arange(1000) * 3
    assert ids.get_local_tensor().is_pinned() and torch.equal(ids[torch.tensor([0, 10, 999])].cpu(), torch.tensor([0, 30, 2997]))
    fs = FeatureStore(location="cpu")
    fs["n", "x", None] = feats
    held = fs["n", "x", None]
    assert held.get_local_tensor().is_pinned() and torch.equal(held[ix].cpu(), feats[ix.cpu()])
    assert torch.equal(fs["n", "x", ix].cpu(), feats[ix.cpu()])
    dev_fs = FeatureStore()                                   # the default here: HBM
    dev_fs["n", "x", None] = feats
    assert dev_fs["n", "x", None].get_local_tensor().is_cuda
    with pytest.raises(ValueError):
        FeatureStore(location="disk")


def test_dist_tensor_invalid_cases():
    from cugraph_pyg_amd.tensor import DistEmbedding, DistTensor
    for kwargs in (dict(shape=[1, 2, 3], dtype=torch.float32), dict(), dict(src="invalid.txt"), dict(shape=[4])):
        with pytest.raises(ValueError):
            DistTensor(**kwargs)
    with pytest.raises(Exception):      # a policy that is not one of wholegraph_amd's is refused by create_embedding
        DistEmbedding(shape=[4, 4], dtype=torch.float32, cache_policy=object())
    t = DistTensor(shape=[5], dtype=torch.int64)
    assert "DistTensor(" in repr(t) and t.dim == 1
    t[torch.arange(5)] = torch.arange(5) * 2
    assert t[torch.tensor([4, 0])].tolist() == [8, 0]


@pytest.mark.parametrize("device", ["cuda", "cpu"])
@pytest.mark.parametrize("access", ["readonly", "readwrite"])
def test_dist_embedding_forwards_its_cache_policy(device, access):
    """``DistEmbedding(cache_policy=...)`` (reference dist_tensor.py:385-399 hands the policy to pylibwholegraph's
    create_embedding): the table is a handle of the HIP library behind the policy's device cache; lookups are bit-exact, are
    served from cache lines on the second pass, and a write through ``__setitem__`` is what the next lookup returns."""
    import wholegraph_amd as wg
    from cugraph_pyg_amd.tensor import DistEmbedding
    comm = wg.get_global_communicator()
    pol = wg.create_wholememory_cache_policy(comm, memory_type="chunked" if access == "readonly" else "distributed",
                                             memory_location="cuda", access_type=access, ratio=0.25)
    table = torch.randn((20011, 64))
    emb = DistEmbedding.from_tensor(table, device=device, name="emb", cache_policy=pol)
    assert "DistEmbedding(name=emb" in repr(emb) and tuple(emb.shape) == (20011, 64) and emb.dtype == torch.float32
    g = torch.Generator().manual_seed(3)
    hot = torch.randint(0, 500, (30000,), generator=g)
    idx = torch.where(torch.rand(30000, generator=g) < 0.7, hot, torch.randint(0, 20011, (30000,), generator=g))
    for _ in range(2):
        got = emb[idx]
        assert got.is_cuda and torch.equal(got.cpu(), table[idx])
    hits, misses, _ = emb._embedding.cache_stats()
    assert hits > 0
    rows = torch.tensor([3, 17, 20010])
    emb[rows] = torch.full((3, 64), 9.0)
    table[rows] = 9.0
    assert torch.equal(emb[idx].cpu(), table[idx]) and torch.equal(emb[rows].cpu(), table[rows])
    out = torch.empty((100, 64), device="cuda")
    assert torch.equal(emb.gather_into(idx[:100], out).cpu(), table[idx[:100]])
    wg.destroy_embedding(emb._embedding)
    wg.destroy_wholememory_cache_policy(pol)


def test_dist_embedding_setitem_with_dirty_rw_cache_lines():
    """with a READWRITE policy, rows left DIRTY in the cache by an optimizer step used to be written
    back over the rows a later ``emb[rows] = v`` had just scattered (drop_all_cache flushes before it empties).  The write
    must win for the written rows, and the optimizer's update must survive for every other row."""
    import wholegraph_amd as wg
    from cugraph_pyg_amd.tensor import DistEmbedding
    comm = wg.get_global_communicator()
    pol = wg.create_wholememory_cache_policy(comm, memory_type="distributed", memory_location="cuda", access_type="readwrite", ratio=0.5)
    n, dim = 4001, 32
    table = torch.randn((n, dim))
    emb = DistEmbedding.from_tensor(table, device="cuda", name="emb_rw", cache_policy=pol)
    opt = wg.create_wholememory_optimizer(emb._embedding, "sgd", {})
    touched = torch.arange(0, 600, dtype=torch.int64).cuda()
    grads = torch.ones((600, dim), device="cuda")
    emb._embedding.gather(touched)                        # lines become resident
    emb._embedding.add_gradients(touched, grads)
    emb._embedding.need_apply = True
    opt.step(0.5)                                         # rows 0..599 are now modified IN THE CACHE (dirty lines)
    want = table.clone()
    want[:600] -= 0.5
    rows = torch.tensor([5, 17, 599, 3000])
    emb[rows] = torch.full((4, dim), 9.0)
    want[rows] = 9.0
    torch.testing.assert_close(emb[torch.arange(n)].cpu(), want, rtol=1e-6, atol=1e-6)
    emb._embedding.writeback_all_cache()
    torch.testing.assert_close(emb.get_local_tensor().cpu(), want, rtol=1e-6, atol=1e-6)
    wg.destroy_wholememory_optimizer(opt)
    wg.destroy_embedding(emb._embedding)
    wg.destroy_wholememory_cache_policy(pol)


def test_pinned_host_rows_are_fenced_before_the_host_sees_them():
    """A scatter into pinned-host rows runs on the current stream; ``get_local_tensor`` / ``load_from_*`` wait for it
    (a host read straight after ``__setitem__`` used to be able to see the rows from before it)."""
    from cugraph_pyg_amd.tensor import DistTensor
    t = DistTensor(shape=[200000, 64], dtype=torch.float32, device="cpu")
    idx = torch.randperm(200000, device="cuda")
    for k in range(5):
        torch.cuda._sleep(20_000_000)                         # the scatter waits behind this on the stream
        t[idx] = torch.full((200000, 64), float(k + 1), device="cuda")
        local = t.get_local_tensor()
        assert not local.is_cuda and float(local.min()) == k + 1 and float(local.max()) == k + 1
    torch.cuda._sleep(20_000_000)
    t[idx[:1000]] = torch.zeros((1000, 64), device="cuda")
    t.load_from_local_tensor(torch.full((200000, 64), 7.0))   # must land AFTER the queued scatter
    torch.cuda.synchronize()
    assert float(t.get_local_tensor().min()) == 7.0


def test_dist_matrix():
    from cugraph_pyg_amd.tensor import DistMatrix
    col = torch.randint(0, 100, (1000,), dtype=torch.long, device="cuda")
    row = torch.randint(0, 100, (1000,), dtype=torch.long, device="cuda")
    m = DistMatrix(src=(col, row), device="cuda", format="coo")
    assert m.shape == (1000, 1000) and m.dtype == torch.long and m._format == "coo"
    idx = torch.randint(0, 1000, (10,))
    res = m[idx]
    assert res.shape == (2, 10) and torch.equal(res[0], col[idx.cuda()]) and torch.equal(res[1], row[idx.cuda()])
    assert torch.equal(m.local_coo, torch.stack([col, row]))
    e = DistMatrix(shape=(50, 50), dtype=torch.int64, format="coo")
    ix = torch.arange(0, 50, 2)
    e[ix] = torch.stack([ix * 2, ix * 3])
    assert torch.equal(e[ix].cpu(), torch.stack([ix * 2, ix * 3]))
    e[:] = (torch.arange(50), torch.arange(50) + 1)
    assert torch.equal(e[torch.tensor([49])].cpu(), torch.tensor([[49], [50]]))
    with pytest.raises(ValueError):
        DistMatrix(shape=(5, 5), dtype=torch.int64, format="csc")
    with pytest.raises(ValueError):
        e[ix] = torch.zeros(3, ix.numel(), dtype=torch.int64)
    with pytest.raises(ValueError):
        m[idx.view(2, 5)]
