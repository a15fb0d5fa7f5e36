"""HIP IPC between two PROCESSES on one GPU — the mechanism the peer-mapped memory types (CHUNKED / CONTINUOUS,
csrc/wg_comm.hip: wholememory_malloc) use for ranks that are separate processes: the owner exports a 64-byte handle of its
hipMalloc'ed partition (wgamd_ipc_export), the peer maps it (wgamd_ipc_open) and reads / writes it with ordinary kernels.
(The threads-as-ranks tests cover the kernels; a process cannot open its own handle, so this test covers the mapping.)"""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent(r"""
    import ctypes, sys
    import torch
    sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/cugraph-gnn_amd")
    from wholegraph_amd import _lib as L
    from wholegraph_amd.tensor import _DevicePointerView
    rows, dim = int(sys.argv[3]), int(sys.argv[4])
    handle = (ctypes.c_char * 64).from_buffer_copy(bytes.fromhex(sys.argv[2]))
    torch.cuda.init()
    ptr = ctypes.c_void_p()
    L.check(L.lib().wgamd_ipc_open(handle, ctypes.byref(ptr)), "wgamd_ipc_open")
    keep = object()
    view = torch.as_tensor(_DevicePointerView(ptr.value, (rows, dim), "<f4", keep), device="cuda")
    want = (torch.arange(rows, device="cuda").view(-1, 1) * 1000 + torch.arange(dim, device="cuda")).float()
    assert torch.equal(view, want), "the mapped partition does not hold the owner's rows"
    # a gather kernel of the library reading the PEER's memory through the mapping
    import wholegraph_amd as wg
    idx = torch.tensor([rows - 1, 0, 7, 7, 3], device="cuda")
    got = wg.tensor.local_gather(view, idx, torch.empty((idx.numel(), dim), device="cuda"))
    assert torch.equal(got, want[idx])
    view[5] = -1.0                      # and a store the owner must see
    torch.cuda.synchronize()
    L.check(L.lib().wgamd_ipc_close(ptr), "wgamd_ipc_close")
    print("CHILD_OK")
""")


def test_export_open_read_write_across_processes(hiplib):
    import ctypes
    import torch
    from wholegraph_amd import _lib as L
    rows, dim = 4096, 100
    owner = (torch.arange(rows, device="cuda").view(-1, 1) * 1000 + torch.arange(dim, device="cuda")).float()
    # hipIpcGetMemHandle wants the base of an allocation: a block of its own straight from the driver (a tensor of torch's caching
    # allocator may be a slice of a larger cached segment, depending on what ran before in the process)
    from wholegraph_amd.tensor import _DevicePointerView
    hip = ctypes.CDLL("libamdhip64.so")
    base = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(base), ctypes.c_size_t(rows * dim * 4)) == 0
    block = torch.as_tensor(_DevicePointerView(base.value, (rows, dim), "<f4", base), device="cuda")
    block.copy_(owner)
    torch.cuda.synchronize()
    handle = (ctypes.c_char * 64)()
    L.check(L.lib().wgamd_ipc_export(ctypes.c_void_p(block.data_ptr()), handle), "wgamd_ipc_export")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT, bytes(handle).hex(), str(rows), str(dim)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "CHILD_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
    torch.cuda.synchronize()
    ok = bool(torch.all(block[5] == -1.0)) and torch.equal(block[6], owner[6])
    del block
    torch.cuda.synchronize()
    hip.hipFree(base)
    assert ok
