"""``WholeMemoryTensor`` — the feature/embedding table handle (the "WholeGraph kv-store").

Mirrors the slice of ``pylibwholegraph.torch.tensor.WholeMemoryTensor`` the hot path uses
(/root/reference/python/pylibwholegraph/pylibwholegraph/torch/tensor.py:24-90,200-319):
``shape/dtype/dim``, ``gather(indice, force_dtype=)``, ``scatter(input, indice)``,
``get_local_tensor()``.

Layout on MI355X: a table is EITHER one device tensor (single GPU / a
replicated CSR array) OR a node-local range partition — rank r owns rows
``[offsets[r], offsets[r+1])`` in its own HBM — and remote rows are fetched with the RCCL
all-to-all pipeline in ``dist.py``.  The reference's mapped memory types (CUDA VMM / cudaIpc /
NVSHMEM; memory_handle.cpp) are deliberately not reproduced.
"""
from typing import Optional, Sequence, Union

import torch

from . import _lib as L
from . import dist as _dist
from .env import get_stream, get_wholegraph_env_fns, wrap_torch_tensor


def local_gather(table: torch.Tensor, indice: torch.Tensor, output: torch.Tensor):
    """output[i,:] = convert(table[indice[i],:]) through ``wholememory_gather`` (HIP kernel)."""
    w_t, w_i, w_o = wrap_torch_tensor(table), wrap_torch_tensor(indice), wrap_torch_tensor(output)
    L.check(L.lib().wholememory_gather(w_t.c, w_i.c, w_o.c, get_wholegraph_env_fns(), get_stream(), -1),
            "wholememory_gather")
    return output


def local_scatter(input_tensor: torch.Tensor, indice: torch.Tensor, table: torch.Tensor):
    """table[indice[i],:] = convert(input[i,:]) through ``wholememory_scatter`` (HIP kernel)."""
    w_in, w_i, w_t = wrap_torch_tensor(input_tensor), wrap_torch_tensor(indice), wrap_torch_tensor(table)
    L.check(L.lib().wholememory_scatter(w_in.c, w_i.c, w_t.c, get_wholegraph_env_fns(), get_stream(), -1),
            "wholememory_scatter")


class HipLocalOps:
    """Local row kernels used by the distributed pipeline (product default: the HIP library)."""

    gather = staticmethod(local_gather)
    scatter = staticmethod(local_scatter)


class WholeMemoryTensor(object):
    r"""WholeMemory Tensor (single device tensor, or node-local range partition)."""

    def __init__(self, local_tensor: torch.Tensor, *, global_rows: Optional[int] = None,
                 partition_offsets: Optional[Sequence[int]] = None, group=None, local_ops=HipLocalOps):
        assert local_tensor.dim() in (1, 2)
        self.local_tensor = local_tensor
        self.group = group
        self.local_ops = local_ops
        if partition_offsets is None:
            self.partition_offsets = None
            self._rows = local_tensor.shape[0]
        else:
            self.partition_offsets = [int(v) for v in partition_offsets]
            self._rows = int(global_rows if global_rows is not None else self.partition_offsets[-1])
            assert self.partition_offsets[0] == 0 and self.partition_offsets[-1] == self._rows

    # ---- metadata ------------------------------------------------------------------------
    @property
    def dtype(self):
        return self.local_tensor.dtype

    def dim(self):
        return self.local_tensor.dim()

    @property
    def shape(self):
        return (self._rows,) + tuple(self.local_tensor.shape[1:])

    def stride(self):
        return self.local_tensor.stride()

    def storage_offset(self):
        return 0

    @property
    def is_distributed(self):
        return self.partition_offsets is not None

    def get_local_tensor(self, host_view: bool = False):
        """(local tensor, first global row held locally) — tensor.py:106-123."""
        start = 0
        if self.is_distributed:
            start = self.partition_offsets[_dist.rank(self.group)]
        return (self.local_tensor.cpu() if host_view else self.local_tensor), start

    # ---- ops -----------------------------------------------------------------------------
    def gather(self, indice: torch.Tensor, *, force_dtype: Union[torch.dtype, None] = None):
        assert indice.dim() == 1
        embedding_dim = self.shape[1] if self.dim() == 2 else 1
        output_dtype = force_dtype if force_dtype is not None else self.dtype
        output_tensor = torch.empty([indice.shape[0], embedding_dim], device=indice.device, dtype=output_dtype,
                                    requires_grad=False)
        table2d = self.local_tensor if self.dim() == 2 else self.local_tensor.unsqueeze(1)
        if self.is_distributed:
            _dist.distributed_gather(table2d, self.partition_offsets, indice, output_tensor, group=self.group,
                                     local_ops=self.local_ops)
        else:
            self.local_ops.gather(table2d, indice, output_tensor)
        return output_tensor.view(-1) if self.dim() == 1 else output_tensor

    def scatter(self, input_tensor: torch.Tensor, indice: torch.Tensor):
        assert indice.dim() == 1
        assert input_tensor.dim() == self.dim()
        assert indice.shape[0] == input_tensor.shape[0]
        if self.dim() == 2:
            assert input_tensor.shape[1] == self.shape[1]
            table2d = self.local_tensor
        else:
            input_tensor = input_tensor.unsqueeze(1)
            table2d = self.local_tensor.unsqueeze(1)
        if self.is_distributed:
            _dist.distributed_scatter(input_tensor, indice, table2d, self.partition_offsets, group=self.group,
                                      local_ops=self.local_ops)
        else:
            self.local_ops.scatter(input_tensor, indice, table2d)


def equal_entry_partition(total_rows: int, world_size: int):
    """Row offsets of the equal range partition: per = ceil(V/W), rank r owns
    [min(r*per, V), min((r+1)*per, V))
    (/root/reference/cpp/src/wholememory/memory_handle.cpp:1613-1629; public
    wholememory_equal_entry_partition_plan, cpp/include/wholememory/wholememory.h:380)."""
    per = (total_rows + world_size - 1) // world_size
    return [min(r * per, total_rows) for r in range(world_size + 1)]


def create_wholememory_tensor(shape, dtype, *, device=None, group=None, partition_offsets=None,
                              local_ops=HipLocalOps):
    """Allocate a (possibly range-partitioned) table.  With a process group of size > 1 every rank
    allocates only its own slice (tensor.py:200-247 with memory type 'distributed')."""
    shape = tuple(shape)
    device = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
    ws = _dist.world_size(group)
    if ws == 1 and partition_offsets is None:
        return WholeMemoryTensor(torch.empty(shape, dtype=dtype, device=device), local_ops=local_ops)
    offs = list(partition_offsets) if partition_offsets is not None else equal_entry_partition(shape[0], ws)
    r = _dist.rank(group)
    local = torch.empty((offs[r + 1] - offs[r],) + shape[1:], dtype=dtype, device=device)
    return WholeMemoryTensor(local, global_rows=shape[0], partition_offsets=offs, group=group, local_ops=local_ops)
