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
    with pytest.raises(NotImplementedError):
        DistEmbedding(shape=[4, 4], dtype=torch.float32, cache_policy=object())
    t = DistTensor(shape=[5], dtype=torch.int64)
    assert "DistTensor(" in repr(t) and t.dim == 1
    t[torch.arange(5)] = torch.arange(5) * 2
    assert t[torch.tensor([4, 0])].tolist() == [8, 0]


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
