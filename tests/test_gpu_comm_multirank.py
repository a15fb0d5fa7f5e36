"""World sizes 2..5 of the C-level DISTRIBUTED gather / scatter (csrc/wg_comm.hip) on ONE GPU.

RCCL cannot put two ranks on one device, so the test process loads tests/shim/libfake_rccl.so (an in-process
stand-in for the nine RCCL calls, selected through WGAMD_RCCL_LIBRARY) and runs every rank as a thread.
Everything else — owner histogram, bucketing, id localisation, local row kernels, un-permute, uneven and empty
partitions, negative indices, dtype conversion — is the product code.  Runs in a subprocess because the RCCL
choice is made once per process.
"""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_DIR = os.path.join(ROOT, "tests", "shim")
SHIM = os.path.join(SHIM_DIR, "build", "libfake_rccl.so")

WORKER = textwrap.dedent(r"""
    import ctypes, sys, threading
    import numpy as np, torch
    sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/cugraph-gnn_amd")
    import wholegraph_amd as wg
    from wholegraph_amd import _lib as L
    from wholegraph_amd.comm import WholeMemoryCommunicator
    W, rows, dim, n = (int(v) for v in sys.argv[2:6])
    part = [int(v) for v in sys.argv[6].split(",")] if len(sys.argv) > 6 and sys.argv[6] else None
    mtype = sys.argv[7] if len(sys.argv) > 7 else "distributed"
    tdt, odt = torch.float16, torch.float32
    lib = L.lib()
    uid = L.UniqueId()
    L.check(lib.wholememory_create_unique_id(ctypes.byref(uid)), "uid")
    rng = np.random.default_rng(W * 1000 + rows)
    table = torch.from_numpy(rng.standard_normal((rows, dim)).astype(np.float32)).to(tdt)
    errors, results = [], [None] * W
    import tempfile
    tmpdir = tempfile.mkdtemp()

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            c = ctypes.c_void_p()
            L.check(lib.wholememory_create_communicator(ctypes.byref(c), uid, r, W), "create_communicator")
            comm = WholeMemoryCommunicator(c.value)
            assert comm.get_rank() == r and comm.get_size() == W
            assert comm.support_type_location(mtype, "cuda") and not comm.support_type_location("hierarchy", "cuda")
            t = wg.create_wholememory_tensor(comm, mtype, "cuda", [rows, dim], tdt, [dim, 1], part)
            assert ("peer-mapped" in t.fetch_path()) == (mtype != "distributed"), t.fetch_path()
            local, start = t.get_local_tensor()
            offs = np.concatenate([[0], np.cumsum(part)]) if part else np.minimum(
                np.arange(W + 1) * -(-rows // W), rows)
            assert start == offs[r] and local.shape[0] == offs[r + 1] - offs[r], (start, local.shape, offs)
            local.zero_()
            comm.barrier()
            # rank r writes the rows with id % W == r (wherever they live), in a shuffled order, int32 ids
            mine = np.arange(r, rows, W)
            np.random.default_rng(r).shuffle(mine)
            t.scatter(table[torch.from_numpy(mine)].cuda(), torch.from_numpy(mine).int().cuda())
            comm.barrier()
            assert torch.equal(local.cpu(), table[offs[r]:offs[r + 1]]), "scatter landed wrong"
            if mtype != "distributed":
                # the chunked view: every rank's partition is addressable from here, mine is my own allocation
                ptrs = (ctypes.c_void_p * W)()
                handle = ctypes.c_void_p(lib.wholememory_tensor_get_memory_handle(t.c))
                L.check(lib.wgamd_get_peer_pointers(ptrs, handle), "wgamd_get_peer_pointers")
                for q in range(W):
                    assert (ptrs[q] is not None) == (offs[q + 1] > offs[q]), (q, ptrs[q])
                assert (ptrs[r] or 0) == (local.data_ptr() if local.numel() else 0)
            # the handle / tensor queries the reference's Cython binding makes (wholememory_binding.pyx:31-262,501-565)
            handle = ctypes.c_void_p(lib.wholememory_tensor_get_memory_handle(t.c))
            eo, ep = (ctypes.c_size_t * (W + 1))(), (ctypes.c_size_t * W)()
            L.check(lib.wholememory_tensor_get_entry_offsets(eo, t.c), "entry_offsets")
            L.check(lib.wholememory_tensor_get_entry_partition_sizes(ep, t.c), "entry_partition_sizes")
            assert list(eo) == [int(v) for v in offs] and list(ep) == [int(b - a) for a, b in zip(offs, offs[1:])]
            ls, lo = ctypes.c_size_t(), ctypes.c_size_t()
            L.check(lib.wholememory_get_local_size(ctypes.byref(ls), handle), "local_size")
            L.check(lib.wholememory_get_local_offset(ctypes.byref(lo), handle), "local_offset")
            row_b = dim * 2
            assert ls.value == (offs[r + 1] - offs[r]) * row_b and lo.value == offs[r] * row_b
            for q in range(W):
                pq, sq, oq = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_size_t()
                rc = lib.wholememory_get_rank_memory(ctypes.byref(pq), ctypes.byref(sq), ctypes.byref(oq), q, handle)
                if q == r or mtype != "distributed":
                    assert rc == 0 and sq.value == (offs[q + 1] - offs[q]) * row_b and oq.value == offs[q] * row_b
                    if q == r:
                        assert (pq.value or 0) == (local.data_ptr() if local.numel() else 0)
                else:
                    assert rc == L.WHOLEMEMORY_INVALID_INPUT      # a DISTRIBUTED handle cannot address a peer's rows
            assert lib.wholememory_get_rank_memory(ctypes.byref(pq), ctypes.byref(sq), ctypes.byref(oq), W, handle) != 0
            gp = ctypes.c_void_p()
            assert lib.wholememory_get_global_pointer(ctypes.byref(gp), handle) == L.WHOLEMEMORY_INVALID_INPUT or W == 1 \
                or (offs[r + 1] - offs[r]) == rows
            sub_c = ctypes.c_void_p()
            assert lib.wholememory_get_local_communicator(ctypes.byref(sub_c), handle) == L.WHOLEMEMORY_NOT_SUPPORTED
            assert lib.wholememory_get_cross_communicator(ctypes.byref(sub_c), handle) == L.WHOLEMEMORY_NOT_SUPPORTED
            assert lib.wholememory_is_intranode_communicator(c) and not lib.wholememory_is_intra_mnnvl_communicator(c)
            assert lib.wholememory_communicator_get_distributed_backend(c) == 1
            assert lib.wholememory_communicator_set_distributed_backend(c, 2) == L.WHOLEMEMORY_NOT_SUPPORTED
            assert lib.wholememory_communicator_set_distributed_backend(c, 1) == 0
            ci = L.CliqueInfo()
            L.check(lib.wholememory_communicator_get_clique_info(ctypes.byref(ci), c), "clique_info")
            assert ci.is_in_clique == 0 and ci.clique_num == 0
            # split (collective): even / odd ranks, ordered by DESCENDING parent rank (key = -r); the last rank of an odd
            # world sits out (WHOLEMEMORY_SPLIT_NOCOLOR) and gets NULL
            sits_out = W % 2 == 1 and r == W - 1 and W > 1
            color = -1 if sits_out else r % 2
            L.check(lib.wholememory_split_communicator(ctypes.byref(sub_c), c, color, -r), "split")
            if sits_out:
                assert not sub_c.value
            else:
                group = [q for q in range(W) if q % 2 == color and not (W % 2 == 1 and q == W - 1 and W > 1)][::-1]
                sr, ss = ctypes.c_int(), ctypes.c_int()
                L.check(lib.wholememory_communicator_get_rank(ctypes.byref(sr), sub_c), "rank")
                L.check(lib.wholememory_communicator_get_size(ctypes.byref(ss), sub_c), "size")
                assert ss.value == len(group) and group[sr.value] == r, (r, group, sr.value, ss.value)
                # the new communicator carries a table of its own: partitioned over the group only
                ts = wg.create_wholememory_tensor(WholeMemoryCommunicator(sub_c.value), "distributed", "cuda",
                                                  [64, 4], torch.float32, [4, 1])
                lt, st = ts.get_local_tensor()
                per = -(-64 // len(group))
                assert st == min(64, per * sr.value) and lt.shape[0] == min(64, per * (sr.value + 1)) - st
                lt.copy_(torch.arange(st, st + lt.shape[0], device="cuda", dtype=torch.float32)[:, None].expand(-1, 4))
                L.check(lib.wholememory_communicator_barrier(sub_c), "sub barrier")
                got = ts.gather(torch.arange(63, -1, -1, device="cuda"))
                assert torch.equal(got[:, 0].cpu(), torch.arange(63, -1, -1, dtype=torch.float32)), "gather over the split"
                L.check(lib.wholememory_communicator_barrier(sub_c), "sub barrier")
                wg.destroy_wholememory_tensor(ts)
                L.check(lib.wholememory_destroy_communicator(sub_c), "destroy sub")
            comm.barrier()
            # gather with duplicates, negatives, fp16 -> fp32 conversion, a different count on every rank
            k = n + 37 * r if n else (0 if r % 2 == 0 else 5)
            idx = np.random.default_rng(100 + r).integers(0, rows, k)
            if k > 4:
                idx[::5] = -1
            out = torch.full((k, dim), 3.0, dtype=odt, device="cuda")
            w_i, w_o = wg.env.wrap_torch_tensor(torch.from_numpy(idx).cuda()), wg.env.wrap_torch_tensor(out)
            L.check(lib.wholememory_gather(t.c, w_i.c, w_o.c, wg.env.get_wholegraph_env_fns(), wg.env.get_stream(), -1),
                    "wholememory_gather")
            want = table.to(odt)[torch.from_numpy(np.where(idx < 0, 0, idx))]
            want[torch.from_numpy(idx < 0)] = 3.0
            assert torch.equal(out.cpu(), want), f"rank {r}: gather mismatch"
            # same dtype in and out: the rows this rank owns are copied straight from its partition (never exchanged)
            out2 = torch.full((k, dim), 3.0, dtype=tdt, device="cuda")
            w_o2 = wg.env.wrap_torch_tensor(out2)
            L.check(lib.wholememory_gather(t.c, w_i.c, w_o2.c, wg.env.get_wholegraph_env_fns(), wg.env.get_stream(), -1),
                    "wholememory_gather")
            assert torch.equal(out2.cpu(), want.to(tdt)), f"rank {r}: same-dtype gather mismatch"
            # every DISTINCT row through the exchange once, expanded locally (gather(dedup=True)): same rows, and the rows
            # of negative ids are left alone; a rank may take this path while another takes the plain one (one collective each)
            from wholegraph_amd.tensor import dedup_pays
            idx_d = torch.from_numpy(idx).cuda()
            out3 = torch.full((k, dim), 3.0, dtype=odt, device="cuda")
            t.gather(idx_d, force_dtype=odt, out=out3, dedup=(r % 2 == 0))
            assert torch.equal(out3.cpu(), want), f"rank {r}: de-duplicated gather mismatch"
            out4 = torch.full((k, dim), 3.0, dtype=tdt, device="cuda")
            t.gather(idx_d.int(), out=out4, dedup="auto")
            assert torch.equal(out4.cpu(), want.to(tdt)), f"rank {r}: auto gather mismatch"
            # an id past the table on ONE rank: it still enters the collective (bad ids left out) and raises AFTER it; the
            # other ranks finish their fetch instead of hanging in the exchange (tensor.gather_distinct)
            idx_bad = idx_d.clone()
            if r == 0 and k > 0:
                idx_bad[k - 1] = rows + 5
            try:
                t.gather(idx_bad, force_dtype=odt, out=out3, dedup=True)
                raised = False
            except IndexError:
                raised = True
            assert raised == (r == 0 and k > 0), (r, k, raised)
            if not raised:
                assert torch.equal(out3.cpu(), want), f"rank {r}: gather next to a failing rank"
            assert dedup_pays(10_900_000, 2_449_029, 8) and not dedup_pays(10_900_000, 2_449_029, 1)
            assert not dedup_pays(1000, 2_449_029, 8) and not dedup_pays(10**7, 1 << 31, 8)
            comm.barrier()
            # file I/O: every rank stores its rows; reload (a) the part files in order into a table with a DIFFERENT
            # partition, (b) round-robin sharded (blocks of 16 rows dealt to the ranks in turn)
            import os, tempfile
            prefix = os.path.join(tmpdir, "tab")
            t.to_file_prefix(prefix)
            names = [f"{prefix}_part_{i}_of_{W}" for i in range(W)]
            assert os.path.getsize(names[r]) == (offs[r + 1] - offs[r]) * dim * 2
            t2 = wg.create_wholememory_tensor(comm, "distributed", "cuda", [rows, dim], tdt, [dim, 1])
            t2.from_filelist(names)
            l2, s2 = t2.get_local_tensor()
            assert torch.equal(l2.cpu(), table[s2:s2 + l2.shape[0]]), "reload mismatch"
            if rows >= 16 * W:
                rr_rows = [sum(min(16, rows - g0) for g0 in range(q * 16, rows, W * 16)) for q in range(W)]
                t3 = wg.create_wholememory_tensor(comm, "distributed", "cuda", [rows, dim], tdt, [dim, 1], rr_rows)
                t3.from_filelist(names, round_robin_size=16)
                want = [g0 + j for g0 in range(r * 16, rows, W * 16) for j in range(min(16, rows - g0))]
                assert torch.equal(t3.get_local_tensor()[0].cpu(), table[want]), "round-robin reload mismatch"
                wg.destroy_wholememory_tensor(t3)
            wg.destroy_wholememory_tensor(t2)
            wg.destroy_wholememory_tensor(t)
            comm.destroy()
            results[r] = "ok"
        except BaseException as e:  # noqa
            import traceback; traceback.print_exc()
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(W)]
    for th in threads: th.start()
    for th in threads: th.join(120)
    alive = [i for i, th in enumerate(threads) if th.is_alive()]
    if alive or errors or any(v != "ok" for v in results):
        print("FAILED", alive, errors, results); sys.stdout.flush()
        import os; os._exit(1)
    print("ALL_RANKS_OK")
""")


@pytest.fixture(scope="module")
def shim():
    if not os.path.exists(SHIM):
        subprocess.run(["make", "-C", SHIM_DIR, "-s"], check=True)
    return SHIM


@pytest.mark.parametrize("W,rows,dim,n,part", [
    (2, 1000, 64, 3000, ""),
    (3, 1001, 7, 500, ""),
    (4, 50000, 128, 20000, ""),
    (4, 10, 4, 64, "0,7,0,3"),        # empty partitions
    (5, 777, 33, 0, "100,200,77,300,100"),  # some ranks gather nothing
    (8, 4096, 100, 4096, ""),
])
@pytest.mark.parametrize("mtype", ["distributed", "chunked"])
def test_world_gt1_threads_over_fake_rccl(shim, W, rows, dim, n, part, mtype):
    """mtype = "chunked": the PEER-MAPPED memory type (reference: gather_op_impl_mapped.cu:18-67) — same-dtype gather and
    scatter are one kernel that addresses every rank's partition directly (ranks of one process share the pointers; ranks
    in different processes map them through HIP IPC, tests/test_gpu_ipc_two_processes.py); a converting gather still
    goes through the exchange."""
    env = dict(os.environ, WGAMD_RCCL_LIBRARY=shim)
    p = subprocess.run([sys.executable, "-c", WORKER, ROOT, str(W), str(rows), str(dim), str(n), part, mtype],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ALL_RANKS_OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]


def test_continuous_is_peer_mapped_too(shim):
    env = dict(os.environ, WGAMD_RCCL_LIBRARY=shim)
    p = subprocess.run([sys.executable, "-c", WORKER, ROOT, "3", "5000", "100", "2000", "", "continuous"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ALL_RANKS_OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]


FAIL_WORKER = textwrap.dedent(r"""
    import ctypes, os, sys, threading
    import torch
    sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/cugraph-gnn_amd")
    import wholegraph_amd as wg
    from wholegraph_amd import _lib as L
    from wholegraph_amd.comm import WholeMemoryCommunicator
    W, location = int(sys.argv[2]), sys.argv[3]
    lib = L.lib()
    uid = L.UniqueId()
    L.check(lib.wholememory_create_unique_id(ctypes.byref(uid)), "uid")
    results = [None] * W

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            c = ctypes.c_void_p()
            L.check(lib.wholememory_create_communicator(ctypes.byref(c), uid, r, W), "create_communicator")
            comm = WholeMemoryCommunicator(c.value)
            try:
                wg.create_wholememory_tensor(comm, "chunked", location, [3000, 16], torch.float32, [16, 1])
                results[r] = "allocated"
            except L.WholeMemoryError as e:
                results[r] = "error %d" % e.code
            # the communicator is still usable by everybody: the next collective allocation goes through
            os.environ.pop("WGAMD_TEST_FAIL_MALLOC_RANK", None)
            comm.barrier()
            t = wg.create_wholememory_tensor(comm, "chunked", location, [3000, 16], torch.float32, [16, 1])
            wg.destroy_wholememory_tensor(t)
            comm.destroy()
        except BaseException as e:  # noqa
            import traceback; traceback.print_exc()
            results[r] = "crash " + repr(e)

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(W)]
    for th in threads: th.start()
    for th in threads: th.join(90)
    alive = [i for i, th in enumerate(threads) if th.is_alive()]
    print("RESULTS", alive, results); sys.stdout.flush()
    os._exit(0 if not alive and all(v and v.startswith("error") for v in results) else 1)
""")


@pytest.mark.parametrize("location", ["cuda", "cpu"])
def test_peer_mapped_malloc_fails_on_every_rank_when_one_partition_cannot_be_allocated(shim, location):
    """A peer-mapped allocation is collective: when ONE rank cannot allocate its partition, every rank meets it in the exchange
    and all of them get the error (OUT_OF_MEMORY) — nobody is left waiting in an allgather — and the communicator stays usable."""
    env = dict(os.environ, WGAMD_RCCL_LIBRARY=shim, WGAMD_TEST_FAIL_MALLOC_RANK="1")
    p = subprocess.run([sys.executable, "-c", FAIL_WORKER, ROOT, "3", location], env=env, capture_output=True, text=True, timeout=200)
    assert p.returncode == 0 and "RESULTS [] ['error" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
