"""N>1 path on CPU: world_size-2 gloo run of the range-partitioned feature store
(bucket ids -> all-to-all ids -> local gather -> all-to-all rows -> un-permute;
reference algorithm /root/reference/cpp/src/wholememory_ops/gather_op_impl_nccl.cu:23-171).

The product's local row kernels are HIP; here the TEST injects the oracle's CPU row kernels as
``local_ops`` so the host logic (partitioning, bucketing, exchange, permutation) runs without a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class OracleLocalOps:
    """test-only stand-in for wholegraph_amd.tensor.HipLocalOps, backed by the oracle"""

    @staticmethod
    def _np(t, other):
        # numpy has no bfloat16: same-dtype row copies move the 16-bit patterns
        return t.view(torch.int16).numpy() if t.dtype == torch.bfloat16 and other.dtype == torch.bfloat16 else t.numpy()

    @staticmethod
    def gather(table, idx, out):
        import oracle
        oracle.gather(OracleLocalOps._np(table, out), idx.numpy(), out=OracleLocalOps._np(out, table))
        return out

    @staticmethod
    def scatter(inp, idx, table):
        import oracle
        oracle.scatter(OracleLocalOps._np(inp, table), idx.numpy(), OracleLocalOps._np(table, inp))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, offsets, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "cugraph-gnn_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from wholegraph_amd import WholeMemoryTensor, create_wholememory_tensor, equal_entry_partition
        V, F = 1003, 7
        full = (torch.arange(V).view(-1, 1) * 10 + torch.arange(F).view(1, -1)).float()  # KAT: row i = 10*i + j
        offs = offsets if offsets is not None else equal_entry_partition(V, world)
        wm = create_wholememory_tensor((V, F), torch.float32, device="cpu", partition_offsets=offs, local_ops=OracleLocalOps)
        assert wm.shape == (V, F) and wm.local_tensor.shape[0] == offs[rank + 1] - offs[rank]
        # load through the distributed scatter: each rank contributes an interleaved half of the rows
        mine = torch.arange(rank, V, world)
        wm.scatter(full[mine], mine)
        dist.barrier()
        assert torch.equal(wm.local_tensor, full[offs[rank]:offs[rank + 1]])
        assert wm.get_local_tensor()[1] == offs[rank]
        # gather arbitrary (rank-dependent, repeated, negative) indices
        g = torch.Generator().manual_seed(100 + rank)
        idx = torch.randint(0, V, (517 + 31 * rank,), generator=g)
        idx[5] = -1
        idx[6] = idx[7]
        out = wm.gather(idx)
        ref = full[idx.clamp(min=0)]
        ref[5] = 0  # skipped row: reference leaves it untouched; our output buffer starts undefined -> compare others
        mask = torch.ones(len(idx), dtype=torch.bool)
        mask[5] = False
        assert torch.equal(out[mask], ref[mask])
        # int32 indices, empty request on one rank (every rank must still take part in the collectives)
        idx32 = torch.randint(0, V, (0 if rank == 0 else 64,), generator=g).int()
        out32 = wm.gather(idx32)
        assert torch.equal(out32, full[idx32.long()])
        # 1-D table
        wm1 = WholeMemoryTensor(torch.arange(offs[rank], offs[rank + 1]) * 3, global_rows=V, partition_offsets=offs,
                                local_ops=OracleLocalOps)
        assert torch.equal(wm1.gather(idx[mask]), idx[mask] * 3)
        # binary file I/O: store per-rank part files, reload them (plain and round-robin sharded) into fresh tables
        import tempfile
        tmp = [tempfile.mkdtemp() if rank == 0 else None]
        dist.broadcast_object_list(tmp, src=0)
        prefix = os.path.join(tmp[0], "feat")
        wm.to_file_prefix(prefix)
        dist.barrier()
        assert os.path.getsize("%s_part_%d_of_%d" % (prefix, rank, world)) == (offs[rank + 1] - offs[rank]) * F * 4
        wm2 = create_wholememory_tensor((V, F), torch.float32, device="cpu", partition_offsets=offs, local_ops=OracleLocalOps)
        wm2.local_tensor.fill_(-1)
        wm2.from_file_prefix(prefix)                      # part files of a DIFFERENT split than the reader's are fine
        assert torch.equal(wm2.local_tensor, full[offs[rank]:offs[rank + 1]])
        rr_rows = [sum(min(50, V - g0) for g0 in range(r * 50, V, world * 50)) for r in range(world)]
        eq = [sum(rr_rows[:r]) for r in range(world + 1)]   # a round-robin shard must fit the rank's rows
        wm3 = create_wholememory_tensor((V, F), torch.float32, device="cpu", partition_offsets=eq, local_ops=OracleLocalOps)
        wm3.local_tensor.fill_(-1)
        wm3.from_filelist(["%s_part_%d_of_%d" % (prefix, r, world) for r in range(world)], round_robin_size=50)
        n_local = eq[rank + 1] - eq[rank]
        rows = [g0 + j for k in range(V) for g0 in [(k * world + rank) * 50] if g0 < V for j in range(min(50, V - g0))]
        assert len(rows) <= n_local and torch.equal(wm3.local_tensor[:len(rows)], full[rows])
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("offsets", [None, [0, 17, 1003], [0, 1003, 1003]])
def test_partitioned_feature_store_world2(oracle_mod, offsets):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, offsets, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_equal_entry_partition_matches_reference_formula():
    # per = ceil(V/W); rank r owns [min(r*per,V), min((r+1)*per,V))  (memory_handle.cpp:1613-1629)
    import sys
    from wholegraph_amd import equal_entry_partition
    assert equal_entry_partition(10, 3) == [0, 4, 8, 10]
    assert equal_entry_partition(2449029, 8)[-1] == 2449029
    assert equal_entry_partition(3, 8) == [0, 1, 2, 3, 3, 3, 3, 3, 3]
    for V, W in ((111059956, 8), (67108864, 8), (7, 2)):
        o = equal_entry_partition(V, W)
        assert len(o) == W + 1 and o[0] == 0 and o[-1] == V and all(b >= a for a, b in zip(o, o[1:]))


def _store_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "cugraph-gnn_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cugraph_pyg_amd.data import FeatureStore, GraphStore
        from cugraph_pyg_amd.tensor import DistTensor
        DistTensor.default_local_ops = OracleLocalOps          # test-only injection (no GPU here)
        # every rank contributes its own slice of the edges (graph_store.py: "each worker should have a slice")
        g = torch.Generator().manual_seed(0)
        full = torch.stack([torch.randint(0, 50, (400,), generator=g), torch.randint(0, 50, (400,), generator=g)])
        mine = full[:, :150] if rank == 0 else full[:, 150:]    # uneven slices, rank order = global edge-id order
        gs = GraphStore()
        gs.put_edge_index(mine, ("n", "e", "n"), "coo", False, (50, 50))
        assert gs.is_multi_gpu and gs.is_homogeneous
        csr = gs._graph
        # reference CSR from the full edge list on one process
        order = torch.sort(full[1], stable=True).indices
        assert torch.equal(csr.col, full[0][order]) and torch.equal(csr.edge_id, order)
        assert csr.row_ptr.tolist() == [0] + torch.cumsum(torch.bincount(full[1], minlength=50), 0).tolist()
        hg = gs._hetero_graphs[("n", "e", "n")]
        assert torch.equal(hg.col, csr.col) and torch.equal(hg.edge_id, csr.edge_id)
        # size inference needs the MAX over ranks
        gs2 = GraphStore()
        gs2.put_edge_index(torch.tensor([[rank * 7], [rank * 9]]), ("n", "e", "n"), "coo")
        assert gs2._num_vertices() == {"n": 10}
        # FeatureStore: slices concatenated in rank order, global gathers from any rank
        fs = FeatureStore()
        x = torch.arange(30 * 4, dtype=torch.float32).view(30, 4)
        fs["n", "x", None] = x[:12] if rank == 0 else x[12:]
        assert tuple(fs.get_tensor_size("n", "x", None)) == (30, 4)
        idx = torch.tensor([29, 0, 12, 11, 5, 5])
        assert torch.equal(fs["n", "x", None][idx], x[idx])
        assert torch.equal(fs["n", "x", idx], x[idx])
        y = torch.arange(30)
        fs["n", "y", None] = y[:12] if rank == 0 else y[12:]
        assert torch.equal(fs["n", "y", None][idx], y[idx])
        names = sorted(a.attr_name for a in fs.get_all_tensor_attrs())
        assert names == ["x", "y"]
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_pyg_stores_world2(oracle_mod):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_store_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"


def _dist_tensor_worker(rank, world, port, q, tmpdir):
    """Mirror of the reference's python/cugraph-pyg/cugraph_pyg/tests/tensor/test_dist_tensor_mg.py:19-170 and
    test_dist_matrix_mg.py:13-120 (creation from a tensor / a .pt / a .npy / binary part files, gathers from any rank,
    invalid cases, COO matrix get / set, even local shares) on two gloo ranks with the oracle's row kernels."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "cugraph-gnn_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        from cugraph_pyg_amd.tensor import DistEmbedding, DistMatrix, DistTensor
        DistTensor.default_local_ops = OracleLocalOps          # test-only injection (no GPU here)
        for clx in (DistTensor, DistEmbedding):
            for dtype in (torch.float32, torch.float16, torch.bfloat16):
                for device in ("cpu", "cuda"):
                    g = torch.Generator().manual_seed(3)
                    features = torch.randn(world * 100 * 10, generator=g).to(dtype).reshape((-1, 10))
                    t = clx.from_tensor(tensor=features, device=device)
                    assert t.shape == features.shape and t.dtype == features.dtype and t.device == device
                    assert t.dim == 2 and t.dim() == 2 and len(t) == features.shape[0]
                    assert t.get_local_tensor().shape[0] == 100 and t.get_local_offset() == 100 * rank
                    ix = torch.randint(0, features.shape[0], (10,), generator=torch.Generator().manual_seed(rank))
                    assert torch.equal(features[ix], t[ix])
                    assert clx.__name__ in repr(t)
            # .pt and .npy sources: every rank keeps its own rows of the file
            features = torch.arange(0, world * 1000).reshape((-1, 100)).to(torch.float32)
            pt, npy = os.path.join(tmpdir, "f.pt"), os.path.join(tmpdir, "f.npy")
            if rank == 0:
                torch.save(features, pt)
                np.save(npy, features.numpy())
            dist.barrier()
            for path in (pt, npy):
                t = clx.from_file(path, device="cuda")
                assert t.shape == features.shape and t.dtype == features.dtype
                ix = torch.randperm(features.shape[0], generator=torch.Generator().manual_seed(5 + rank))[:10]
                assert torch.equal(features[ix], t[ix])
            # binary part files that do not line up with the row partition (utils.py:96-170)
            parts = [os.path.join(tmpdir, "bin_part_%d_of_3" % i) for i in range(3)]
            if rank == 0:
                for f, blk in zip(parts, torch.split(features, [3, 11, 6], dim=0)):
                    blk.numpy().tofile(f)
            dist.barrier()
            t = clx(src=parts, shape=list(features.shape), dtype=torch.float32, partition_book=[13, 7])
            assert t.get_local_offset() == (0 if rank == 0 else 13)
            assert torch.equal(t.get_local_tensor(), features[:13] if rank == 0 else features[13:])
            # __setitem__ from every rank, then a global read-back; load_from_local_tensor
            t2 = clx(shape=[20, 100], dtype=torch.float32)
            mine = torch.arange(rank, 20, world)
            t2[mine] = features[mine]
            dist.barrier()
            assert torch.equal(t2[torch.arange(20)], features)
            t2.load_from_local_tensor(torch.full((10, 100), float(rank)))
            dist.barrier()
            assert torch.equal(t2[torch.tensor([0, 19])][:, 0], torch.tensor([0.0, 1.0]))
            for bad in (torch.zeros(3, 100), torch.zeros((10, 100), dtype=torch.float64)):
                try:
                    t2.load_from_local_tensor(bad)
                    raise AssertionError("shape / dtype mismatch accepted")
                except ValueError:
                    pass
            dist.barrier()
        # invalid cases (test_dist_tensor_mg.py:137-160)
        for kwargs in (dict(shape=[1, 2, 3], dtype=torch.float32), dict(), dict(src="invalid.txt"), dict(shape=[4]),
                       dict(src=["a", "b"])):
            try:
                DistTensor(**kwargs)
                raise AssertionError("accepted %r" % (kwargs,))
            except ValueError:
                pass
        try:
            DistEmbedding(shape=[4, 4], dtype=torch.float32, cache_policy=object())
            raise AssertionError("cache policy accepted")
        except RuntimeError:     # a cached embedding is a handle of the HIP library: no GPU, no cache (it says so)
            pass
        assert DistEmbedding(shape=[4, 4], dtype=torch.float32, name="emb").name == "emb"
        # COO matrix
        g = torch.Generator().manual_seed(11)
        col, row = torch.randint(0, 100, (1001,), generator=g), torch.randint(0, 100, (1001,), generator=g)
        m = DistMatrix(src=(col, row), device="cuda", format="coo")
        assert m.shape == (1001, 1001) and m.dtype == torch.long and m._format == "coo"
        idx = torch.randint(0, 1001, (10,), generator=torch.Generator().manual_seed(rank))
        res = m[idx]
        assert res.shape == (2, 10) and torch.equal(res[0], col[idx]) and torch.equal(res[1], row[idx])
        lo = 0 if rank == 0 else 501                      # 1001 entries over 2 ranks: 501 + 500
        assert torch.equal(m.local_col, col[lo:lo + (501 if rank == 0 else 500)])
        assert torch.equal(m.local_coo, torch.stack([col, row])[:, lo:lo + (501 if rank == 0 else 500)])
        e = DistMatrix(shape=(30, 30), dtype=torch.int64, format="coo")
        mine = torch.arange(rank, 30, world)
        e[mine] = torch.stack([mine * 2, mine * 3])
        dist.barrier()
        assert torch.equal(e[torch.arange(30)], torch.stack([torch.arange(30) * 2, torch.arange(30) * 3]))
        e[mine] = (mine * 5, mine * 7)
        dist.barrier()
        assert torch.equal(e[torch.tensor([29])], torch.tensor([[145], [203]]))
        for bad in (dict(src=(col,)), dict(src="x.bin"), dict(), dict(src=(col, row[:5])), dict(src=("a", "b")), dict(src=5)):
            try:
                DistMatrix(**bad)
                raise AssertionError("accepted %r" % (list(bad),))
            except (ValueError, NotImplementedError):
                pass
        for bad_val in (torch.zeros(3, 2, dtype=torch.int64), torch.zeros(2, dtype=torch.int64)):
            try:
                e[torch.tensor([0, 1])] = bad_val
                raise AssertionError("bad value accepted")
            except ValueError:
                pass
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_dist_tensor_embedding_matrix_world2(oracle_mod, tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dist_tensor_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"
