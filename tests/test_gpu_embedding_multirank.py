"""World sizes 2..4 of the embedding gradient path on ONE GPU (ranks are threads over tests/shim/libfake_rccl.so, see
tests/test_gpu_comm_multirank.py).  Every rank contributes its own (indices, gradients) each step and the same row is hit
from several ranks.  The arrival order at an owner is (peers in rank order, then the owner's own pairs), so the fp32
gradient sums are formed in a different order than the oracle's (which sees the pairs of all ranks concatenated in rank
order): the gradients are multiples of 1/64 so that every order gives the same fp32 sum, and the comparison keeps the
reference test's 1e-5 tolerance (wholememory_embedding_gradient_apply_tests.cu:759).  Random row partitions as in its `use_random_partition` cases."""
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
    from oracle import embedding_optimizer as eo
    W, n, dim, k = (int(v) for v in sys.argv[2:6])
    kind = sys.argv[6]
    part = [int(v) for v in sys.argv[7].split(",")] if len(sys.argv) > 7 and sys.argv[7] else None
    lib = L.lib()
    uid = L.UniqueId()
    L.check(lib.wholememory_create_unique_id(ctypes.byref(uid)), "uid")
    rng = np.random.default_rng(W * 100 + dim)
    table = rng.uniform(-10, 10, (n, dim)).astype(np.float32)
    steps = 3
    idx = rng.integers(0, n, (steps, W, k))
    idx[:, :, ::11] = -1
    grads = (rng.integers(-320, 321, (steps, W, k, dim)) / 64.0).astype(np.float32)  # sums are exact in any order
    ref = table.copy()
    cpu = eo.SparseOptimizer(kind, n, dim)
    for s in range(steps):
        cpu.step(ref, idx[s].reshape(-1), grads[s].reshape(-1, dim), 0.1)
    errors, results = [], [None] * W
    barrier = threading.Barrier(W)

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            c = ctypes.c_void_p()
            L.check(lib.wholememory_create_communicator(ctypes.byref(c), uid, r, W), "create_communicator")
            comm = WholeMemoryCommunicator(c.value)
            emb = wg.create_embedding(comm, "distributed", "cuda", torch.float32, [n, dim], embedding_entry_partition=part)
            opt = wg.create_wholememory_optimizer(emb, kind, {})
            local, first = emb.get_embedding_tensor().get_local_tensor()
            local.copy_(torch.from_numpy(table[first:first + local.shape[0]]).cuda())
            comm.barrier()
            for s in range(steps):
                ki = k if (s + r) % 2 == 0 else k // 2      # ranks bring different amounts; one brings nothing at step 1
                if s == 1 and r == W - 1:
                    ki = 0
                # what a rank does not bring this step is brought by rank 0 instead (keeps the oracle's pair set)
                emb.add_gradients(torch.from_numpy(idx[s, r, :ki]).cuda(), torch.from_numpy(grads[s, r, :ki]).cuda())
                if r == 0:
                    for q in range(W):
                        kq = k if (s + q) % 2 == 0 else k // 2
                        if s == 1 and q == W - 1:
                            kq = 0
                        if kq < k:
                            emb.add_gradients(torch.from_numpy(idx[s, q, kq:]).cuda(), torch.from_numpy(grads[s, q, kq:]).cuda())
                emb.need_apply = True
                opt.step(0.1)
            got = local.cpu().numpy()
            want = ref[first:first + local.shape[0]]
            err = np.abs(got - want)
            ok = (err <= 1e-5) | (err <= 1e-5 * np.maximum(np.abs(got), np.abs(want)))
            assert ok.all(), f"rank {r}: max err {err.max()}"
            for name, w in cpu.states.items():
                g = emb.get_optimizer_state(name).get_local_tensor()[0].cpu().numpy()
                w = w[first:first + local.shape[0]]
                e2 = np.abs(g - w)
                assert ((e2 <= 1e-5) | (e2 <= 1e-5 * np.maximum(np.abs(g), np.abs(w)))).all(), f"rank {r}: state {name}"
            # forward after training: every rank gathers rows of every partition
            q = np.random.default_rng(r).integers(0, n, 333)
            out = emb.gather(torch.from_numpy(q).cuda()).cpu().numpy()
            e3 = np.abs(out - ref[q])
            assert ((e3 <= 1e-5) | (e3 <= 1e-5 * np.abs(ref[q]))).all(), f"rank {r}: gather after training"
            comm.barrier()
            wg.destroy_embedding(emb)
            wg.destroy_wholememory_optimizer(opt)
            comm.destroy()
            results[r] = "ok"
        except BaseException as e:  # noqa
            import traceback; traceback.print_exc()
            print("FAILED rank", r, repr(e)); sys.stdout.flush(); sys.stderr.flush()
            import os; os._exit(1)   # the other ranks would wait for this one in the next collective

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(W)]
    for th in threads: th.start()
    for th in threads: th.join(240)
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


@pytest.mark.parametrize("W,n,dim,k,kind,part", [
    (2, 5000, 64, 4000, "sgd", ""),
    (2, 5000, 127, 4000, "lazy_adam", "1234,3766"),
    (3, 3001, 32, 2000, "adagrad", ""),
    (4, 2048, 128, 3000, "rmsprop", "0,1000,48,1000"),   # an empty partition
    (4, 20000, 100, 9000, "lazy_adam", ""),
])
def test_embedding_training_world_gt1(shim, W, n, dim, k, kind, part):
    env = dict(os.environ, WGAMD_RCCL_LIBRARY=shim)
    p = subprocess.run([sys.executable, "-c", WORKER, ROOT, str(W), str(n), str(dim), str(k), kind, part],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ALL_RANKS_OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
