"""GPU: the range-partitioned feature store over RCCL (backend "nccl") with the HIP row kernels.  The
GPU box has one GPU, so this is a world_size-1 process group: every collective and kernel of the
N > 1 path executes (bucket -> all_to_all_single -> local HIP gather -> all_to_all_single -> HIP
un-permute); the multi-rank routing itself is covered by tests/test_dist_gloo.py."""
import os
import socket

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partitioned_store_over_rccl_world1(hiplib):
    import torch
    import torch.distributed as dist
    from wholegraph_amd import WholeMemoryTensor, create_wholememory_tensor
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        V, F = 200_003, 100
        full = torch.randn(V, F, device="cuda")
        wm = create_wholememory_tensor((V, F), torch.float32, partition_offsets=[0, V])
        assert wm.is_distributed and wm.shape == (V, F)
        ids = torch.arange(V, device="cuda")
        wm.scatter(full, ids)                                   # distributed scatter (loads the store)
        assert torch.equal(wm.local_tensor, full)
        idx = torch.randint(0, V, (100_005,), device="cuda")
        idx[7] = -1
        out = wm.gather(idx)
        ok = torch.ones(idx.numel(), dtype=torch.bool, device="cuda")
        ok[7] = False
        assert torch.equal(out[ok], full[idx[ok]])
        assert torch.equal(wm.gather(idx[ok].int()), full[idx[ok]])     # int32 indices
        assert wm.gather(idx[:0]).shape == (0, F)                       # empty request still joins the collectives
        half = wm.gather(idx[ok], force_dtype=torch.float16)            # converting gather on the remote path
        assert torch.equal(half, full[idx[ok]].half())
        one_d = WholeMemoryTensor(torch.arange(V, device="cuda") * 3, global_rows=V, partition_offsets=[0, V])
        assert torch.equal(one_d.gather(idx[ok]), idx[ok] * 3)
        # the cugraph_pyg-shaped stores on top of it
        from cugraph_pyg_amd.data import FeatureStore
        fs = FeatureStore()
        fs["paper", "x", None] = full
        assert torch.equal(fs["paper", "x", None][idx[ok]], full[idx[ok]])
        assert tuple(fs.get_tensor_size("paper", "x", None)) == (V, F)
    finally:
        dist.destroy_process_group()
