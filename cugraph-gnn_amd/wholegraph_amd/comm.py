"""WholeMemory communicators over RCCL, created through the C ABI (include/wgamd_comm.h).

Mirrors the slice of ``pylibwholegraph.torch.comm`` a training script touches
(/root/reference/python/pylibwholegraph/pylibwholegraph/torch/comm.py:62-225):
``WholeMemoryCommunicator`` (rank / size / barrier / support_type_location / destroy),
``create_group_communicator``, ``get_global_communicator``, ``destroy_communicator``.

Bootstrap is the reference's: rank 0 of the group creates the 128-byte unique id, it travels by
``torch.distributed.broadcast`` and every rank calls ``wholememory_create_communicator``.  One process per
GPU; node-local / device-local / MNNVL communicators of the reference have no role on a single xGMI node
and are not provided.
"""
import ctypes

import torch

from . import _lib as L

_MEMORY_TYPES = {"none": L.MT_NONE, "continuous": L.MT_CONTINUOUS, "chunked": L.MT_CHUNKED,
                 "distributed": L.MT_DISTRIBUTED, "hierarchy": L.MT_HIERARCHY}
_LOCATIONS = {"none": L.ML_NONE, "cuda": L.ML_DEVICE, "cpu": L.ML_HOST}

_global_communicators = {}


def memory_type_code(name) -> int:
    return name if isinstance(name, int) else _MEMORY_TYPES[name]


def memory_location_code(name) -> int:
    return name if isinstance(name, int) else _LOCATIONS[name]


class WholeMemoryCommunicator(object):
    """Owner of one ``wholememory_comm_t``.  Use ``create_group_communicator`` /
    ``get_global_communicator`` rather than constructing it directly."""

    def __init__(self, c_comm: int):
        self.c_comm = ctypes.c_void_p(c_comm)

    def get_rank(self) -> int:
        v = ctypes.c_int(-1)
        L.check(L.lib().wholememory_communicator_get_rank(ctypes.byref(v), self.c_comm), "communicator_get_rank")
        return v.value

    def get_size(self) -> int:
        v = ctypes.c_int(-1)
        L.check(L.lib().wholememory_communicator_get_size(ctypes.byref(v), self.c_comm), "communicator_get_size")
        return v.value

    def barrier(self):
        """Device-side 1-int all-reduce + host sync on the default stream (comm.py:95-104)."""
        L.check(L.lib().wholememory_communicator_barrier(self.c_comm), "communicator_barrier")

    def rccl_info(self):
        """(ranks, version) as RCCL itself reports them for this communicator (ncclCommCount / ncclGetVersion; -1 where the
        loaded library has no such symbol)."""
        n, v = ctypes.c_int(-1), ctypes.c_int(-1)
        L.check(L.lib().wgamd_communicator_rccl_info(self.c_comm, ctypes.byref(n), ctypes.byref(v)), "communicator_rccl_info")
        return n.value, v.value

    def support_type_location(self, memory_type: str, memory_location: str) -> bool:
        rc = L.lib().wholememory_communicator_support_type_location(
            self.c_comm, memory_type_code(memory_type), memory_location_code(memory_location))
        return rc == L.WHOLEMEMORY_SUCCESS

    def destroy(self):
        if self.c_comm is not None:
            L.check(L.lib().wholememory_destroy_communicator(self.c_comm), "destroy_communicator")
            self.c_comm = None

    @property
    def distributed_backend(self):
        return "nccl"  # RCCL


def _broadcast_unique_id(uid: "L.UniqueId", src_global_rank: int, group, device):
    """Move the unique id from the group's first rank to the others (comm.py:159-166)."""
    import torch.distributed as dist
    buf = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).clone()
    use_dev = dist.get_backend(group) == "nccl"
    if use_dev:
        buf = buf.to(device)
    dist.broadcast(buf, src=src_global_rank, group=group)
    raw = bytes(buf.cpu().numpy().tobytes())
    ctypes.memmove(ctypes.byref(uid), raw, ctypes.sizeof(uid))


def create_group_communicator(group_size: int = -1, comm_stride: int = 1, *, device=None):
    """Communicator over ranks {base + i*comm_stride}: consecutive blocks of ``group_size * comm_stride``
    world ranks are split into ``comm_stride`` interleaved groups (comm.py:132-170).  ``group_size=-1`` = all
    ranks.  Without an initialised process group a world of 1 is assumed."""
    import torch.distributed as dist
    lib = L.lib()
    have_pg = dist.is_available() and dist.is_initialized()
    world_rank = dist.get_rank() if have_pg else 0
    world_size = dist.get_world_size() if have_pg else 1
    if group_size == -1:
        group_size = world_size
    strided = group_size * comm_stride
    assert world_size % strided == 0, "world size must be a multiple of group_size * comm_stride"
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    uid = L.UniqueId()
    my_comm = None
    # every rank walks all groups (new_group is collective over the world), keeps its own
    for block in range(world_size // strided):
        for lane in range(comm_stride):
            members = [block * strided + lane + i * comm_stride for i in range(group_size)]
            if world_size == 1 or not have_pg:
                L.check(lib.wholememory_create_unique_id(ctypes.byref(uid)), "create_unique_id")
                mine, rank_in_group = True, 0
            else:
                pg = dist.group.WORLD if len(members) == world_size else dist.new_group(members)
                mine = world_rank in members
                if mine:
                    rank_in_group = members.index(world_rank)
                    if rank_in_group == 0:
                        L.check(lib.wholememory_create_unique_id(ctypes.byref(uid)), "create_unique_id")
                    _broadcast_unique_id(uid, members[0], pg, device)
            if mine:
                with torch.cuda.device(device):
                    c = ctypes.c_void_p()
                    L.check(lib.wholememory_create_communicator(ctypes.byref(c), uid, rank_in_group, group_size),
                            "create_communicator")
                my_comm = WholeMemoryCommunicator(c.value)
    return my_comm


def destroy_communicator(wm_comm: WholeMemoryCommunicator):
    if wm_comm is not None:
        wm_comm.destroy()


def get_global_communicator(distributed_backend="nccl"):
    """The all-ranks communicator, created on first use (comm.py:202-225)."""
    assert distributed_backend == "nccl", "only the RCCL backend exists on MI355X"
    if "nccl" not in _global_communicators:
        _global_communicators["nccl"] = create_group_communicator()
    return _global_communicators["nccl"]


def get_local_node_communicator():
    """One communicator per node (comm.py:228-246); the node size is ``LOCAL_WORLD_SIZE`` (torchrun), else the world."""
    import os
    import torch.distributed as dist
    if "local_node" not in _global_communicators:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        local = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        _global_communicators["local_node"] = create_group_communicator(local if world % local == 0 else world)
    return _global_communicators["local_node"]


def get_local_device_communicator():
    """A communicator of this rank alone (comm.py:249-266)."""
    if "local_device" not in _global_communicators:
        _global_communicators["local_device"] = create_group_communicator(1)
    return _global_communicators["local_device"]


def reset_communicators():
    for c in _global_communicators.values():
        c.destroy()
    _global_communicators.clear()
