"""C-level communicator + DISTRIBUTED handle (include/wgamd_comm.h, csrc/wg_comm.hip) on one GPU.

A world_size-1 RCCL communicator exercises the full pipeline (owner histogram, bucket, the grouped
send/recv self-exchange, local gather, un-permute); results are compared with the oracle's row gather /
scatter (wholememory_ops/functions/gather_scatter_func.cuh semantics: a negative index leaves its row
untouched).  World sizes > 1 of the same algorithm are covered on CPU by tests/test_dist_gloo.py and run by
the driver's multi-GPU bench.
"""
import ctypes

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


def test_communicator_queries(comm):
    assert comm.get_rank() == 0 and comm.get_size() == 1
    assert comm.support_type_location("distributed", "cuda")
    assert comm.support_type_location("distributed", "cpu")      # pinned host partitions (tests/test_gpu_embedding_rw_cache.py)
    assert not comm.support_type_location("hierarchy", "cuda")
    comm.barrier()


@pytest.mark.parametrize("dtype,out_dtype", [(torch.float32, None), (torch.float16, torch.float32),
                                             (torch.bfloat16, None), (torch.int64, None), (torch.int8, torch.int32)])
@pytest.mark.parametrize("idx_dtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("rows,dim,n", [(1000, 100, 4096), (37, 1, 100), (5000, 33, 0), (5000, 128, 1)])
def test_handle_scatter_then_gather_matches_oracle(comm, dtype, out_dtype, idx_dtype, rows, dim, n):
    import wholegraph_amd as wg
    from oracle import oracle
    rng = np.random.default_rng(rows * 7 + dim + n)
    t = wg.create_wholememory_tensor(comm, "distributed", "cuda", [rows, dim], dtype, [dim, 1])
    try:
        local, start = t.get_local_tensor()
        assert start == 0 and tuple(local.shape) == (rows, dim) and local.dtype == dtype
        if dtype.is_floating_point:
            host = torch.from_numpy(rng.standard_normal((rows, dim)).astype(np.float32)).to(dtype)
        else:
            host = torch.from_numpy(rng.integers(-100, 100, (rows, dim))).to(dtype)
        # fill through the collective scatter (a permutation of all rows), read back through the local view
        perm = torch.from_numpy(rng.permutation(rows)).to(idx_dtype).cuda()
        t.scatter(host.cuda()[perm.long()], perm)
        torch.cuda.synchronize()
        assert torch.equal(local.cpu(), host)
        idx = rng.integers(0, rows, n)
        if n > 3:
            idx[1] = -1  # negative index: output row stays as it was
        got = torch.full((n, dim), 7, dtype=out_dtype or dtype, device="cuda")
        w_i = wg.env.wrap_torch_tensor(torch.from_numpy(idx).to(idx_dtype).cuda())
        w_o = wg.env.wrap_torch_tensor(got)
        wg._lib.check(wg._lib.lib().wholememory_gather(t.c, w_i.c, w_o.c, wg.env.get_wholegraph_env_fns(),
                                                        wg.env.get_stream(), -1), "wholememory_gather")
        torch.cuda.synchronize()
        want = host.to(out_dtype or dtype)[torch.from_numpy(np.where(idx < 0, 0, idx))]
        if n > 3:
            want[1] = 7
        assert torch.equal(got.cpu(), want)
        if dtype == torch.float32 and n:
            ref = oracle.gather_rows(host.numpy(), np.where(idx < 0, 0, idx).astype(np.int64))
            keep = idx >= 0
            assert np.array_equal(got.cpu().numpy()[keep], ref[keep])
        # the method form
        if n:
            out = t.gather(torch.from_numpy(np.abs(idx)).to(idx_dtype).cuda(), force_dtype=out_dtype)
            assert torch.equal(out.cpu(), host.to(out_dtype or dtype)[torch.from_numpy(np.abs(idx))])
    finally:
        wg.destroy_wholememory_tensor(t)


def test_handle_1d_and_partition_queries(comm):
    import wholegraph_amd as wg
    L = wg._lib
    t = wg.create_wholememory_tensor(comm, "distributed", "cuda", [4097], torch.int64, [1])
    try:
        h = ctypes.c_void_p(L.lib().wholememory_tensor_get_memory_handle(t.c))
        assert L.lib().wholememory_tensor_has_handle(t.c)
        assert L.lib().wholememory_get_total_size(h) == 4097 * 8
        assert L.lib().wholememory_get_data_granularity(h) == 8
        assert L.lib().wholememory_get_memory_type(h) == L.MT_DISTRIBUTED
        assert L.lib().wholememory_get_memory_location(h) == L.ML_DEVICE
        sizes, offs = (ctypes.c_size_t * 1)(), (ctypes.c_size_t * 2)()
        L.check(L.lib().wholememory_get_rank_partition_sizes(sizes, h), "sizes")
        L.check(L.lib().wholememory_get_rank_partition_offsets(offs, h), "offsets")
        assert sizes[0] == 4097 * 8 and list(offs) == [0, 4097 * 8]
        vals = torch.arange(4097, dtype=torch.int64, device="cuda") * 3
        t.scatter(vals, torch.arange(4097, device="cuda"))
        idx = torch.tensor([0, 4096, 17, 17, 2048], device="cuda")
        assert torch.equal(t.gather(idx), idx * 3)
        # local tensor mapped as a plain-pointer wholememory tensor
        lt = ctypes.c_void_p()
        L.check(L.lib().wholememory_tensor_map_local_tensor(t.c, ctypes.byref(lt)), "map_local_tensor")
        assert not L.lib().wholememory_tensor_has_handle(lt)
        assert L.lib().wholememory_tensor_get_data_pointer(lt) == t.get_local_tensor()[0].data_ptr()
        L.lib().wholememory_destroy_tensor(lt)
    finally:
        wg.destroy_wholememory_tensor(t)


def test_unsupported_memory_types_are_refused(comm):
    import wholegraph_amd as wg
    with pytest.raises(wg.WholeMemoryError):
        wg.create_wholememory_tensor(comm, "hierarchy", "cuda", [16, 4], torch.float32, [4, 1])
    with pytest.raises(wg.WholeMemoryError):  # partition that does not add up
        wg.create_wholememory_tensor(comm, "distributed", "cuda", [16, 4], torch.float32, [4, 1], [15])


@pytest.mark.parametrize("dtype,dim", [(torch.float32, 100), (torch.float16, 33), (torch.int64, 1)])
def test_handle_file_roundtrip(comm, tmp_path, dtype, dim):
    """wholememory_store_to_file / wholememory_load_from_file: headerless row-major binary files; a list of files
    is one concatenated array (reference file_io.cpp:1893-2160), also split across files at arbitrary rows."""
    import wholegraph_amd as wg
    rows = 70001  # > one 16 MB staging chunk for the wide dtypes? no: several chunks at dim 100 fp32 = 28 MB
    rng = np.random.default_rng(5)
    host = (rng.standard_normal((rows, dim)).astype(np.float32) if dtype.is_floating_point
            else rng.integers(-1000, 1000, (rows, dim)))
    host = torch.from_numpy(host).to(dtype)
    t = wg.create_wholememory_tensor(comm, "distributed", "cuda", [rows, dim], dtype, [dim, 1])
    t2 = wg.create_wholememory_tensor(comm, "distributed", "cuda", [rows + 10, dim], dtype, [dim, 1])
    try:
        t.get_local_tensor()[0].copy_(host)
        t.to_file_prefix(str(tmp_path / "tab"))
        part = tmp_path / "tab_part_0_of_1"
        assert part.stat().st_size == rows * dim * host.element_size()
        raw = np.fromfile(part, dtype=np.int8)
        assert np.array_equal(raw, host.view(torch.int8).numpy().reshape(-1))
        # reload from three files cut at arbitrary rows (one of them empty) into a LARGER table: the tail stays
        cuts = [0, 12345, 12345, rows]
        names = []
        for i in range(3):
            f = tmp_path / f"piece{i}.bin"
            host[cuts[i]:cuts[i + 1]].view(torch.int8).numpy().tofile(f)
            names.append(str(f))
        t2.get_local_tensor()[0].fill_(7)
        t2.from_filelist(names)
        got = t2.get_local_tensor()[0].cpu()
        assert torch.equal(got[:rows], host) and bool((got[rows:] == 7).all())
        # errors: file size not a multiple of the row size; more rows than the table holds
        bad = tmp_path / "bad.bin"
        bad.write_bytes(b"x" * (dim * host.element_size() + 1))
        with pytest.raises(wg.WholeMemoryError):
            t.from_filelist(str(bad))
        with pytest.raises(wg.WholeMemoryError):
            t.from_filelist(names + names)
        with pytest.raises(wg.WholeMemoryError):
            t.from_filelist(str(tmp_path / "missing.bin"))
    finally:
        wg.destroy_wholememory_tensor(t)
        wg.destroy_wholememory_tensor(t2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16, torch.float64, torch.int32, torch.int64,
                                   torch.int16, torch.int8])
@pytest.mark.parametrize("entries,dim", [(13, 7), (0, 5), (1000, 128)])
def test_env_test_op_exercises_every_allocation_type(dtype, entries, dim):
    """wholememory_env_test_op (wholememory_op.h:61-68, wholememory_test_op.cu:53-140): out[i, j] = (T)(float)i + input[j]
    through temporary memory into the fixed output and one output per allocation type (device / pinned / host) — the
    self-test `wholememory_env_test_cython_op` of the reference binding runs on its allocator callbacks."""
    import wholegraph_amd as wg
    from wholegraph_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(entries + dim)
    inp = (torch.randint(-5, 5, (dim,), generator=g).to(dtype) if not dtype.is_floating_point
           else torch.randn(dim, generator=g).to(dtype)).cuda()
    fixed = torch.zeros(entries, dim + 3, dtype=dtype, device="cuda")[:, :dim]      # strided rows
    ctxs = [wg.env.TorchMemoryContext() for _ in range(3)]
    w_i, w_f = wg.env.wrap_torch_tensor(inp), wg.env.wrap_torch_tensor(fixed)
    L.check(lib.wholememory_env_test_op(w_i.c, w_f.c, *[c.get_c_context() for c in ctxs], entries,
                                        wg.env.get_wholegraph_env_fns(), wg.env.get_stream()), "env_test_op")
    tag = torch.arange(entries, dtype=torch.float32).to(dtype)[:, None]
    want = (tag + inp.cpu()[None, :]).to(dtype)
    assert torch.equal(fixed.cpu(), want)
    kinds = ["cuda", "cpu", "cpu"]
    for c, kind in zip(ctxs, kinds):
        t = c.get_tensor()
        assert t is not None and tuple(t.shape) == (entries, dim) and t.dtype == dtype and t.device.type == kind
        assert torch.equal(t.cpu(), want)
    assert ctxs[1].get_tensor().is_pinned() or entries == 0


def test_a_stale_tensor_cannot_free_the_handle_that_took_its_address(hiplib):
    """A communicator that is destroyed force-releases the handles it still had; a tensor that owned one of them is destroyed
    LATER.  By then the allocator may have handed the same address to a new, unrelated handle: the stale destroy must leave it
    alone (handles carry a serial number that is never reused; `wholememory_destroy_tensor` frees only the handle it created)."""
    import wholegraph_amd as wg
    c1 = wg.create_group_communicator()
    old = [wg.create_wholememory_tensor(c1, "distributed", "cuda", [100, 8], torch.float32, [8, 1]) for _ in range(6)]
    c1.destroy()                                      # the six handles go with it; `old` still points at their addresses
    c2 = wg.create_group_communicator()
    new = [wg.create_wholememory_tensor(c2, "distributed", "cuda", [100, 8], torch.float32, [8, 1]) for _ in range(12)]
    for k, t in enumerate(new):
        t.get_local_tensor()[0].fill_(float(k))
    for t in old:
        wg.destroy_wholememory_tensor(t)              # stale: must not touch anything live
    idx = torch.arange(0, 100, 3, device="cuda")
    for k, t in enumerate(new):
        assert bool((t.gather(idx) == float(k)).all()), "a live handle was released by a stale tensor"
    for t in new:
        wg.destroy_wholememory_tensor(t)
    c2.destroy()
