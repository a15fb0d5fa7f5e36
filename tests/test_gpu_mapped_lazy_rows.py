"""Fetch-in-the-layer over a PEER-MAPPED feature table (CHUNKED / CONTINUOUS handle partitioned over several ranks): the
one-kernel SAGE layer reads remote rows itself through byte offsets over this process's mapping of every partition
(wgamd_mapped_row_offsets + src_ids_dtype = WGAMD_IDS_BYTE_OFFSETS), forward and weight gradient — bit for bit the result of
gathering the rows first (wholememory_gather over the same mapping) and running the layer on the gathered matrix.
Reference behaviour: the mapped gather addresses the partitions through global references
(/root/reference/cpp/src/wholememory_ops/functions/gather_scatter_func.cuh:242-505, gather_op_impl_mapped.cu).

Worlds 2 and 8 on ONE GPU: ranks are threads over the in-process RCCL stand-in (tests/test_gpu_comm_multirank.py); ranks of one
process share the partition pointers, ranks of different processes map them through HIP IPC
(tests/test_gpu_ipc_two_processes.py)."""
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
    sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/cugraph-gnn_amd"); sys.path.insert(0, sys.argv[1] + "/tests")
    import wholegraph_amd as wg
    from wholegraph_amd import _lib as L, nn
    from wholegraph_amd.comm import WholeMemoryCommunicator
    W, rows, F, N, mtype = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    lib = L.lib()
    uid = L.UniqueId()
    L.check(lib.wholememory_create_unique_id(ctypes.byref(uid)), "uid")
    rng = np.random.default_rng(W * 1000 + rows)
    table = torch.from_numpy(rng.standard_normal((rows, F)).astype(np.float32))
    errors, results = [], [None] * W
    lock = threading.Lock()

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            c = ctypes.c_void_p()
            L.check(lib.wholememory_create_communicator(ctypes.byref(c), uid, r, W), "create_communicator")
            comm = WholeMemoryCommunicator(c.value)
            t = wg.create_wholememory_tensor(comm, mtype, "cuda", [rows, F], torch.float32, [F, 1])
            assert "peer-mapped" in t.fetch_path()
            local, start = t.get_local_tensor()
            local.copy_(table[start:start + local.shape[0]])
            torch.cuda.synchronize()
            comm.barrier()
            g = torch.Generator(device="cuda").manual_seed(100 + r)
            n_src, n_dst = 5000, 1200 + 37 * r
            ids = torch.randint(0, rows, (n_src,), generator=g, device="cuda", dtype=torch.int64 if r % 2 else torch.int32)
            ids[:3] = torch.tensor([0, rows - 1, rows // 2], device="cuda").to(ids.dtype)        # partition edges
            deg = torch.randint(0, 30, (n_dst,), generator=g, device="cuda")
            rp = torch.zeros(n_dst + 1, dtype=torch.int32, device="cuda"); rp[1:] = torch.cumsum(deg, 0)
            col = torch.randint(0, n_src, (int(rp[-1]),), generator=g, device="cuda", dtype=torch.int32)
            self_rows = torch.randperm(n_src, generator=g, device="cuda")[:n_dst].contiguous()
            w_t = torch.randn((2 * F, N), generator=g, device="cuda") * 0.1
            bias = torch.randn(N, generator=g, device="cuda")
            # (a) gather, then the layer on the gathered rows
            x = t.gather(ids)
            assert torch.equal(x.cpu(), table[ids.cpu().long()])
            agg_a = torch.empty((n_dst, F), device="cuda")
            out_a = nn.sage_layer_fused_forward(rp, col, x, self_rows, w_t, bias, relu=True, mean=True, agg_out=agg_a)
            # (b) the layer reads the mapped partitions itself
            lazy = nn.mapped_lazy_rows(t, ids)
            assert isinstance(lazy.table, nn.MappedTable) and lazy.ids.dtype == torch.int64 and len(lazy) == n_src
            agg_b = torch.empty((n_dst, F), device="cuda")
            out_b = nn.sage_layer_fused_forward(rp, col, lazy.table, self_rows, w_t, bias, relu=True, mean=True, src_ids=lazy.ids,
                                                agg_out=agg_b)
            assert torch.equal(out_a, out_b) and torch.equal(agg_a, agg_b), f"rank {r}: layer over the mapping differs"
            assert torch.equal(lazy.materialize(), x)
            # weight gradient: self rows through the same offsets
            gout = torch.randn((n_dst, N), generator=g, device="cuda")
            ga = [torch.empty((N, F), device="cuda"), torch.empty((N, F), device="cuda"), torch.empty(N, device="cuda")]
            gb = [torch.empty((N, F), device="cuda"), torch.empty((N, F), device="cuda"), torch.empty(N, device="cuda")]
            with lock:      # (the weight-gradient scratch is one buffer per device: the thread ranks of this test share a GPU)
                nn.sage_wgrad(agg_a, x, self_rows, gout, *ga, act_out=out_a)
                nn.sage_wgrad(agg_b, lazy.table, self_rows, gout, *gb, act_out=out_b, src_ids=lazy.ids)
                torch.cuda.synchronize()
            assert all(torch.equal(p, q) for p, q in zip(ga, gb)), f"rank {r}: weight gradient over the mapping differs"
            # the module: SAGEConv over a LayerGraph with the mapped LazyRows, autograd on
            conv = nn.SAGEConv(F, N).cuda()
            lg = nn.LayerGraph([nn.HopGraph(rp, col, self_rows)])
            with lock:
                y = conv(lazy, lg, act="relu"); y.backward(gout)
                g_lazy = [p.grad.clone() for p in conv.parameters()]
                for p in conv.parameters(): p.grad = None
                y2 = conv(x, lg, act="relu"); y2.backward(gout)
                torch.cuda.synchronize()
            assert torch.equal(y, y2) and all(torch.equal(p.grad, q) for p, q in zip(conv.parameters(), g_lazy))
            comm.barrier()
            wg.destroy_wholememory_tensor(t)
            comm.destroy()
            results[r] = "ok"
        except BaseException as e:  # noqa
            import traceback; traceback.print_exc()
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(W)]
    for th in threads: th.start()
    for th in threads: th.join(180)
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


@pytest.mark.parametrize("W,rows,F,N,mtype", [(2, 40001, 100, 256, "chunked"), (8, 100003, 100, 256, "chunked"),
                                              (8, 9001, 256, 47, "chunked"), (3, 20000, 128, 128, "continuous")])
def test_layer_reads_a_peer_mapped_table_itself(shim, W, rows, F, N, mtype):
    env = dict(os.environ, WGAMD_RCCL_LIBRARY=shim)
    p = subprocess.run([sys.executable, "-c", WORKER, ROOT, str(W), str(rows), str(F), str(N), mtype], env=env,
                       capture_output=True, text=True, timeout=400)
    assert p.returncode == 0 and "ALL_RANKS_OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]


def test_mapped_row_offsets_refuses_what_it_cannot_address(hiplib):
    """A DISTRIBUTED handle has no pointer to a peer's rows; a single-partition handle is read through its local tensor."""
    import ctypes
    import torch
    import wholegraph_amd as wg
    from wholegraph_amd import _lib as L
    comm = wg.get_global_communicator()
    for mtype in ("distributed", "chunked"):
        t = wg.create_wholememory_tensor(comm, mtype, "cuda", [100, 8], torch.float32, [8, 1])
        ids = torch.arange(10, device="cuda")
        offs = torch.empty(10, dtype=torch.int64, device="cuda")
        base = ctypes.c_void_p()
        rc = L.lib().wgamd_mapped_row_offsets(t.c, ids.data_ptr(), L.DT_INT64, 10, offs.data_ptr(), ctypes.byref(base), None)
        assert rc == L.WHOLEMEMORY_LOGIC_ERROR
        wg.destroy_wholememory_tensor(t)
