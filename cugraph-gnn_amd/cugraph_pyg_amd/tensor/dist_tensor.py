"""``DistTensor`` / ``DistEmbedding`` — the tensors a FeatureStore keeps
(/root/reference/python/cugraph-pyg/cugraph_pyg/tensor/dist_tensor.py:20-534: ``__getitem__`` gathers
by global row, ``__setitem__`` scatters, ``shape``/``dtype``), over ``wholegraph_amd.WholeMemoryTensor``:
one device tensor on a single GPU, a node-local range partition + RCCL all-to-all otherwise.
The reference's ``backend`` ("vmm"/"nccl") and host ``device`` options do not apply: storage is HBM."""
from typing import Optional, Sequence

import torch

from wholegraph_amd import dist as _dist
from wholegraph_amd.tensor import HipLocalOps, WholeMemoryTensor, equal_entry_partition


class DistTensor:
    """1-D (or N-D with dim-0 partitioning) distributed tensor."""

    _ndim = 1
    default_local_ops = HipLocalOps   # the product's row kernels; CPU tests inject the oracle's here

    def __init__(self, src: Optional[torch.Tensor] = None, shape: Optional[Sequence[int]] = None,
                 dtype: Optional[torch.dtype] = None, device: str = "cuda", backend: Optional[str] = None,
                 partition_offsets: Optional[Sequence[int]] = None, group=None, local_ops=None):
        if src is None and (shape is None or dtype is None):
            raise ValueError("Please specify shape and dtype for empty tensor.")
        local_ops = local_ops if local_ops is not None else DistTensor.default_local_ops
        self._group = group
        ws, rk = _dist.world_size(group), _dist.rank(group)
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        if src is not None and ws == 1:
            t = src.to(dev)
            t = t if t.dim() >= 1 else t.view(1)
            self._wm = WholeMemoryTensor(t.contiguous(), local_ops=local_ops)
        else:
            shape = tuple(int(s) for s in (shape if shape is not None else src.shape))
            dtype = dtype or src.dtype
            if ws == 1:
                self._wm = WholeMemoryTensor(torch.zeros(shape, dtype=dtype, device=dev), local_ops=local_ops)
            else:
                offs = list(partition_offsets) if partition_offsets is not None else equal_entry_partition(shape[0], ws)
                local = torch.zeros((offs[rk + 1] - offs[rk],) + shape[1:], dtype=dtype, device=dev)
                self._wm = WholeMemoryTensor(local, global_rows=shape[0], partition_offsets=offs, group=group,
                                             local_ops=local_ops)

    @property
    def shape(self):
        return torch.Size(self._wm.shape)

    @property
    def dtype(self):
        return self._wm.dtype

    @property
    def device(self):
        return self._wm.local_tensor.device

    def dim(self):
        return self._wm.dim()

    def __len__(self):
        return self._wm.shape[0]

    def get_local_tensor(self, host_view=False):
        return self._wm.get_local_tensor(host_view)[0]

    def get_local_offset(self):
        return self._wm.get_local_tensor()[1]

    def _idx(self, idx):
        if isinstance(idx, slice):
            idx = torch.arange(*idx.indices(self._wm.shape[0]))
        idx = torch.as_tensor(idx)
        if idx.dtype not in (torch.int32, torch.int64):
            idx = idx.long()
        return idx.to(self.device).contiguous().view(-1)

    def __getitem__(self, idx) -> torch.Tensor:
        return self._wm.gather(self._idx(idx))

    def __setitem__(self, idx, val: torch.Tensor):
        idx = self._idx(idx)
        val = val.to(device=self.device, dtype=self.dtype)
        if val.dim() < self._wm.dim():
            val = val.view((-1,) + tuple(self._wm.shape[1:]))
        self._wm.scatter(val.contiguous(), idx)


class DistEmbedding(DistTensor):
    """2-D [num_embeddings, embedding_dim] table (dist_tensor.py:283-534)."""

    _ndim = 2
