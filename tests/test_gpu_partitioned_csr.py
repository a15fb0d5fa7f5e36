"""Sampling over a CSR that is itself partitioned over the GPUs of a communicator (DISTRIBUTED / CHUNKED / CONTINUOUS
handles for csr_row_ptr and csr_col): `wholegraph_csr_unweighted_sample_without_replacement` fetches the row offsets and the
picked columns from their owners (csrc/wg_sample.hip run_partitioned; reference
cpp/src/wholegraph_ops/unweighted_sample_without_replacement_nccl_func.cuh:213-372) and must return, on every rank, exactly
what the replicated-CSR op / the oracle returns for that rank's centres and seed.  Ranks are threads over the in-process
RCCL stand-in (tests/test_gpu_comm_multirank.py); a world of one runs the same path in-process.
"""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_DIR = os.path.join(ROOT, "tests", "shim")
SHIM = os.path.join(SHIM_DIR, "build", "libfake_rccl.so")

WORKER = textwrap.dedent(r"""
    import ctypes, sys, threading
    import numpy as np, torch
    sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/cugraph-gnn_amd"); sys.path.insert(0, sys.argv[1] + "/tests")
    import oracle
    import wholegraph_amd as wg
    from graphgen import powerlaw_csr
    from wholegraph_amd import _lib as L, wholegraph_ops
    from wholegraph_amd.comm import WholeMemoryCommunicator
    W, V, M = (int(v) for v in sys.argv[2:5])
    mtype, col_dt = sys.argv[5], (np.int64 if sys.argv[6] == "int64" else np.int32)
    oracle.build()
    lib = L.lib()
    uid = L.UniqueId()
    L.check(lib.wholememory_create_unique_id(ctypes.byref(uid)), "uid")
    row_ptr, col = powerlaw_csr(V, 14, seed=V + M, col_dtype=col_dt, max_deg=2500)
    E = col.shape[0]
    results = [None] * W

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            c = ctypes.c_void_p()
            L.check(lib.wholememory_create_communicator(ctypes.byref(c), uid, r, W), "create_communicator")
            comm = WholeMemoryCommunicator(c.value)
            t_ptr = wg.create_wholememory_tensor(comm, mtype, "cuda", [V + 1], torch.int64, [1])
            t_col = wg.create_wholememory_tensor(comm, mtype, "cuda", [E], torch.from_numpy(col[:1]).dtype, [1])
            for t, full in ((t_ptr, row_ptr), (t_col, col)):
                local, start = t.get_local_tensor()
                local.copy_(torch.from_numpy(full[start:start + local.shape[0]]).cuda())
            torch.cuda.synchronize()
            comm.barrier()
            rng = np.random.default_rng(100 + r)
            for rnd, seed_dt in enumerate((np.int64, np.int32, np.int64)):
                n = 0 if (rnd == 1 and r == W - 1) else 700 + 300 * r     # ranks bring different amounts, once nothing
                centres = rng.integers(0, V, n).astype(seed_dt)
                seed = 1000 * rnd + r
                got = wholegraph_ops.unweighted_sample_without_replacement(t_ptr, t_col, torch.from_numpy(centres).cuda(), M,
                                                                           seed, True, True)
                want = oracle.unweighted_sample(row_ptr, col, centres, M, seed)
                for name, g, w_ in zip(("offsets", "dest", "center_localid", "edge_gid"), got, want):
                    assert np.array_equal(g.cpu().numpy(), w_), (r, rnd, name)
                # the plain call (no optional outputs) returns the same two arrays
                got2 = wholegraph_ops.unweighted_sample_without_replacement(t_ptr, t_col, torch.from_numpy(centres).cuda(), M, seed)
                assert len(got2) == 2 and torch.equal(got2[0], got[0]) and torch.equal(got2[1], got[1])
            # the reference's multi-hop walk (graph_structure.py:136-196) over the partitioned CSR, per rank
            if M > 0 and M <= 40:
                gs = wg.GraphStructure()
                gs.set_csr_graph(t_ptr, t_col)
                seeds = rng.permutation(V)[:200 + 50 * r].astype(np.int64)
                rs = [77 + r, 99 + r]
                got = gs.multilayer_sample_without_replacement(torch.from_numpy(seeds).cuda(), [M, 5], random_seeds=rs)
                want = oracle.multilayer_sample(row_ptr, col, seeds, [M, 5], rs)
                for name, g_l, w_l in zip(("target_gids", "edge_indice", "csr_row_ptr", "csr_col_ind"), got, want):
                    for lvl, (x, y) in enumerate(zip(g_l, w_l)):
                        assert np.array_equal(x.cpu().numpy(), y), (r, name, lvl)
                try:
                    gs.multilayer_sample_nosync(torch.from_numpy(seeds).cuda(), [M, 5])
                    raise AssertionError("no-sync walk accepted a partitioned CSR")
                except NotImplementedError:
                    pass
            comm.barrier()
            wg.destroy_wholememory_tensor(t_ptr); wg.destroy_wholememory_tensor(t_col)
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


@pytest.mark.parametrize("W,V,M,mtype,col_dt", [
    (1, 5000, 10, "distributed", "int64"),
    (2, 20000, 25, "distributed", "int32"),
    (3, 9000, 5, "distributed", "int64"),
    (4, 30000, 40, "chunked", "int32"),       # M > 32: the block kernel
    (2, 12000, -1, "continuous", "int64"),    # sample everything
    (2, 8000, 300, "distributed", "int32"),   # M > 256: the reservoir path
])
def test_partitioned_csr_sampling_matches_oracle(shim, W, V, M, mtype, col_dt):
    env = dict(os.environ, WGAMD_RCCL_LIBRARY=shim)
    p = subprocess.run([sys.executable, "-c", WORKER, ROOT, str(W), str(V), str(M), mtype, col_dt],
                       env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode == 0 and "ALL_RANKS_OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]


def test_weighted_op_refuses_a_partitioned_csr(hiplib):
    import ctypes
    import torch
    import wholegraph_amd as wg
    from wholegraph_amd import _lib as L
    from wholegraph_amd.env import TorchMemoryContext, get_wholegraph_env_fns, wrap_torch_tensor
    comm = wg.create_group_communicator()
    t_ptr = wg.create_wholememory_tensor(comm, "distributed", "cuda", [101], torch.int64, [1])
    t_col = wg.create_wholememory_tensor(comm, "distributed", "cuda", [400], torch.int64, [1])
    w = torch.ones(400, device="cuda")
    seeds = torch.zeros(4, dtype=torch.int64, device="cuda")
    off = torch.empty(5, dtype=torch.int32, device="cuda")
    ctx = TorchMemoryContext()
    rc = L.lib().wholegraph_csr_weighted_sample_without_replacement(
        t_ptr.c, t_col.c, wrap_torch_tensor(w).c, wrap_torch_tensor(seeds).c, 5, wrap_torch_tensor(off).c,
        ctx.get_c_context(), None, None, 1, get_wholegraph_env_fns(), None)
    assert rc == L.WHOLEMEMORY_INVALID_INPUT
    wg.destroy_wholememory_tensor(t_ptr)
    wg.destroy_wholememory_tensor(t_col)
    comm.destroy()
