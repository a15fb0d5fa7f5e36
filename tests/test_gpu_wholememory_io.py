"""Binary file load / store of DISTRIBUTED matrices, mirrored from the reference's
python/pylibwholegraph/pylibwholegraph/tests/pylibwholegraph/test_wholememory_io.py:166-401 for one GPU (its `gpu_count`
ranks collapse to one here; worlds > 1 of the same entry points — part files that do not line up with the row partition,
round-robin shards — run in tests/test_gpu_comm_multirank.py): int32 rows split over 3 / 5 part files at random
boundaries, loaded into a column sub-view (`storage_offset`) of a matrix whose rows are wider than the payload
(`embedding_stride`), optionally dealt round-robin; stored back from such a view."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm():
    import wholegraph_amd as wg
    c = wg.create_group_communicator()
    yield c
    c.destroy()


def _valid(dim, stride, offset, rr=0):
    if stride < offset + dim:
        pytest.skip("embedding_stride, embedding_dim and storage_offset configuration not valid")
    if rr != 0 and offset != 0:
        pytest.skip("round_robin_size != 0 with a storage offset is not valid")


@pytest.mark.parametrize("file_part_count", [3, 5])
@pytest.mark.parametrize("embedding_entry_count", [100003])
@pytest.mark.parametrize("embedding_dim", [16, 31, 33])
@pytest.mark.parametrize("embedding_stride", [16, 32, 64])
@pytest.mark.parametrize("storage_offset", [0, 3])
@pytest.mark.parametrize("round_robin_size", [256, 1024, 0])
def test_wholememory_load(comm, tmp_path, file_part_count, embedding_entry_count, embedding_dim, embedding_stride,
                          storage_offset, round_robin_size):
    import wholegraph_amd as wg
    _valid(embedding_dim, embedding_stride, storage_offset, round_robin_size)
    random.seed(file_part_count * 1000 + embedding_dim)
    n = embedding_entry_count
    base = torch.randint(-1000000000, 1000000000, (n, embedding_dim), dtype=torch.int)
    cuts = sorted(random.sample(range(1, n), file_part_count - 1)) + [n]
    counts = [cuts[0]] + [cuts[i] - cuts[i - 1] for i in range(1, file_part_count)]
    prefix = str(tmp_path / "pytest_load_temp_file")
    files = ["%s_part_%d_of_%d" % (prefix, i, file_part_count) for i in range(file_part_count)]
    for f, part in zip(files, torch.split(base, counts, dim=0)):
        part.numpy().tofile(f)
    extra = n   # one rank: the round-robin padding of the reference (test_wholememory_io.py:57-70) is empty
    root = wg.create_wholememory_tensor(comm, "distributed", "cuda", [extra, embedding_dim + storage_offset], torch.int32,
                                        [embedding_stride, 1])
    root.get_local_tensor()[0].fill_(-7)
    view = root.get_sub_tensor([-1, storage_offset], [-1, -1])
    assert view.shape == (extra, embedding_dim)
    view.from_filelist(files, round_robin_size)
    local, first = view.get_local_tensor()
    assert first == 0 and local.dim() == 2 and tuple(local.shape) == (n, embedding_dim)
    assert torch.equal(local.cpu(), base)
    if storage_offset:   # the columns in front of the view were not written
        assert bool((root.get_local_tensor()[0][:, :storage_offset] == -7).all())
    # the loaded rows are what a gather through the view returns
    q = torch.randint(0, n, (1000,), device="cuda")
    assert torch.equal(view.gather(q).cpu(), base[q.cpu()])
    view.destroy()
    wg.destroy_wholememory_tensor(root)


@pytest.mark.parametrize("embedding_entry_count", [100003])
@pytest.mark.parametrize("embedding_dim", [16, 31, 33])
@pytest.mark.parametrize("embedding_stride", [16, 32, 64])
@pytest.mark.parametrize("storage_offset", [0, 3])
def test_wholememory_store(comm, tmp_path, embedding_entry_count, embedding_dim, embedding_stride, storage_offset):
    import wholegraph_amd as wg
    _valid(embedding_dim, embedding_stride, storage_offset)
    n = embedding_entry_count
    base = torch.randint(-1000000000, 1000000000, (n, embedding_dim), dtype=torch.int)
    root = wg.create_wholememory_tensor(comm, "distributed", "cuda", [n, embedding_dim + storage_offset], torch.int32,
                                        [embedding_stride, 1])
    view = root.get_sub_tensor([-1, storage_offset], [-1, -1])
    view.get_local_tensor()[0].copy_(base.cuda())
    name = str(tmp_path / "pytest_store_temp_file")
    view.local_to_file(name)
    assert os.path.getsize(name) == n * embedding_dim * 4
    assert np.array_equal(np.fromfile(name, dtype=np.int32).reshape(n, embedding_dim), base.numpy())
    view.destroy()
    wg.destroy_wholememory_tensor(root)
