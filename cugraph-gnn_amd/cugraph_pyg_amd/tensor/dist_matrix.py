"""``DistMatrix`` — a COO edge list kept as two DistTensors
(/root/reference/python/cugraph-pyg/cugraph_pyg/tensor/dist_matrix.py:12-162): ``m[idx]`` returns the 2 x len(idx) block
of (col, row) pairs, ``m[idx] = (col, row)`` or a 2 x N tensor writes them, ``local_col`` / ``local_row`` / ``local_coo``
give this rank's even share of the entries."""
from typing import Optional, Tuple, Union

import torch

from wholegraph_amd import dist as _dist

from .dist_tensor import DistTensor


class DistMatrix:
    def __init__(self, src=None, shape: Optional[Union[list, tuple]] = None, dtype: Optional[torch.dtype] = None,
                 device: Optional[str] = "cuda", backend: Optional[str] = "nccl", format: Optional[str] = "coo", **kwargs):
        self._backend = backend
        self._format = format
        if isinstance(src, (tuple, list)):
            if len(src) > 0 and isinstance(src[0], str):
                raise NotImplementedError("Constructing from a file or list of files is not yet supported.")
            if len(src) != 2:
                raise ValueError("src must be a tuple of two tensors")
            as_dt = (lambda t: t if isinstance(t, DistTensor) else
                     DistTensor(src=t if dtype is None else t.to(dtype), device=device, backend=backend, **kwargs))
            self._col, self._row = as_dt(src[0]), as_dt(src[1])
            if self._format == "coo" and self._col.shape[0] != self._row.shape[0]:
                raise ValueError("col and row must have the same number of elements for COO format")
        elif src is None:
            if dtype is None or shape is None:
                raise ValueError("dtype and shape must be provided if src is None")
            if self._format != "coo":
                raise ValueError("Only COO format is supported for empty matrices")
            self._col = DistTensor(src=None, shape=(shape[0],), dtype=dtype, device=device, backend=backend, **kwargs)
            self._row = DistTensor(src=None, shape=(shape[1],), dtype=dtype, device=device, backend=backend, **kwargs)
        elif isinstance(src, str):
            raise NotImplementedError("Constructing from a file or list of files is not yet supported.")
        else:
            raise ValueError("Invalid src type")

    def __setitem__(self, idx, val):
        if isinstance(idx, slice):
            idx = torch.arange(self._col.shape[0])[idx]
        if self._format != "coo":
            raise ValueError("Updating is currently only supported for COO format")
        if isinstance(val, torch.Tensor):
            if val.dim() != 2:
                raise ValueError("val must be a 2D tensor")
            if val.shape[0] != 2:
                raise ValueError("val must be a 2xN tensor")
            if val.shape[1] != idx.shape[0]:
                raise ValueError("val and idx must have compatible shapes")
            self._col[idx] = val[0]
            self._row[idx] = val[1]
        elif isinstance(val, tuple):
            if len(val) != 2:
                raise ValueError("val must be a tuple of two tensors")
            self._col[idx] = val[0]
            self._row[idx] = val[1]

    def __getitem__(self, idx: torch.Tensor) -> torch.Tensor:
        if self._format != "coo":
            raise ValueError("Getting is currently only supported for COO format")
        if idx.dim() != 1:
            raise ValueError("idx must be a 1D tensor")
        return torch.stack([self._col[idx], self._row[idx]])

    def get_local_tensor(self) -> Tuple[torch.Tensor, torch.Tensor]:
        return (self._col.get_local_tensor(), self._row.get_local_tensor())

    def _even_share(self, t: DistTensor) -> torch.Tensor:
        """Entries [rank's even share) fetched through the distributed gather (dist_matrix.py:121-153: q = n // W entries
        per rank, the first n % W ranks take one more) — collective."""
        ws, rank = _dist.world_size(t.get_comm()), _dist.rank(t.get_comm())
        q, r = divmod(t.shape[0], ws)
        lo = q * rank + min(rank, r)
        return t[torch.arange(lo, lo + q + (1 if rank < r else 0))]

    @property
    def local_col(self) -> torch.Tensor:
        return self._even_share(self._col)

    @property
    def local_row(self) -> torch.Tensor:
        return self._even_share(self._row)

    @property
    def local_coo(self) -> torch.Tensor:
        return torch.stack([self.local_col, self.local_row])

    @property
    def shape(self) -> Tuple[int, int]:
        return (self._col.shape[0], self._row.shape[0])

    @property
    def dtype(self) -> torch.dtype:
        return self._col.dtype
