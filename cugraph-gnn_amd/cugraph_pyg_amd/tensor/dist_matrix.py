"""``DistMatrix`` — a distributed COO edge list: two equally long ``DistTensor`` columns (minor ids, major ids) addressed by
entry number.

Behavioural contract (reference: python/cugraph-pyg/cugraph_pyg/tensor/dist_matrix.py:12-162, exercised by
tests/tensor/test_dist_matrix_mg.py): ``m[idx]`` is the ``2 x len(idx)`` block of entries ``idx``; ``m[idx] = block`` writes a
``2 x N`` tensor or a ``(col, row)`` pair; ``local_col / local_row / local_coo`` are this rank's share of an even split of the
entry range (the first ``n % W`` ranks take one entry more); ``shape`` is the pair of column lengths and ``dtype`` the id type.
Only the COO layout carries data — any other ``format`` is refused on every access and on empty construction.  File sources
are not part of this class (``DistTensor`` reads files; a matrix is built from two of them).

Written from that contract: the two columns live in ``self._axes`` and every accessor goes through the three helpers below
instead of repeating the checks per method.
"""
from typing import Optional, Sequence, Tuple, Union

import torch

from wholegraph_amd import dist as _dist

from .dist_tensor import DistTensor

_UNSUPPORTED_SOURCE = "Constructing from a file or list of files is not yet supported."


def _looks_like_path(x) -> bool:
    return isinstance(x, (str, bytes))


class DistMatrix:
    def __init__(self, src=None, shape: Optional[Sequence[int]] = None, dtype: Optional[torch.dtype] = None,
                 device: Optional[str] = "cuda", backend: Optional[str] = "nccl", format: Optional[str] = "coo", **kwargs):
        self._backend, self._format = backend, format
        mk = dict(device=device, backend=backend, **kwargs)
        if src is None:
            # an empty matrix needs its extent and id type up front, and only COO can be filled entry by entry
            if shape is None or dtype is None:
                raise ValueError("dtype and shape must be provided if src is None")
            self._need_coo("Only COO format is supported for empty matrices")
            axes = [DistTensor(src=None, shape=(int(n),), dtype=dtype, **mk) for n in (shape[0], shape[1])]
        elif _looks_like_path(src) or (isinstance(src, (tuple, list)) and any(_looks_like_path(s) for s in src)):
            raise NotImplementedError(_UNSUPPORTED_SOURCE)
        elif isinstance(src, (tuple, list)):
            if len(src) != 2:
                raise ValueError("src must be a tuple of two tensors")
            axes = [s if isinstance(s, DistTensor) else DistTensor(src=s if dtype is None else s.to(dtype), **mk) for s in src]
            if format == "coo" and axes[0].shape[0] != axes[1].shape[0]:
                raise ValueError("col and row must have the same number of elements for COO format")
        else:
            raise ValueError("Invalid src type")
        self._axes: Tuple[DistTensor, DistTensor] = (axes[0], axes[1])

    # the reference exposes the two columns under these names (graph_store.py reads them)
    @property
    def _col(self) -> DistTensor:
        return self._axes[0]

    @property
    def _row(self) -> DistTensor:
        return self._axes[1]

    # ---- helpers ------------------------------------------------------------------------------------------------------
    def _need_coo(self, message: str) -> None:
        if self._format != "coo":
            raise ValueError(message)

    def _entries(self, idx) -> torch.Tensor:
        """Entry numbers as a 1-D index tensor (a slice addresses the whole entry range of the first column)."""
        if isinstance(idx, slice):
            return torch.arange(*idx.indices(self._axes[0].shape[0]))
        return idx

    @staticmethod
    def _as_pair(val, n: int):
        """(col values, row values) out of a 2 x n tensor or a pair; anything else of tensor type is a shape error."""
        if isinstance(val, torch.Tensor):
            if val.dim() != 2:
                raise ValueError("val must be a 2D tensor")
            if val.shape[0] != 2:
                raise ValueError("val must be a 2xN tensor")
            if val.shape[1] != n:
                raise ValueError("val and idx must have compatible shapes")
            return val[0], val[1]
        if isinstance(val, tuple):
            if len(val) != 2:
                raise ValueError("val must be a tuple of two tensors")
            return val
        return None          # other value types are ignored, as in the reference

    def _my_entries(self, axis: DistTensor) -> torch.Tensor:
        """Entry numbers of this rank's share of an even split of ``axis`` (collective when gathered)."""
        comm = axis.get_comm()
        world, rank = _dist.world_size(comm), _dist.rank(comm)
        base, extra = divmod(axis.shape[0], world)
        first = rank * base + min(rank, extra)
        return torch.arange(first, first + base + (rank < extra))

    # ---- element access -----------------------------------------------------------------------------------------------
    def __setitem__(self, idx, val) -> None:
        where = self._entries(idx)
        self._need_coo("Updating is currently only supported for COO format")
        pair = self._as_pair(val, where.shape[0])
        if pair is not None:
            for axis, values in zip(self._axes, pair):
                axis[where] = values

    def __getitem__(self, idx: torch.Tensor) -> torch.Tensor:
        self._need_coo("Getting is currently only supported for COO format")
        if idx.dim() != 1:
            raise ValueError("idx must be a 1D tensor")
        return torch.stack([axis[idx] for axis in self._axes])

    def get_local_tensor(self) -> Tuple[torch.Tensor, torch.Tensor]:
        return tuple(axis.get_local_tensor() for axis in self._axes)

    @property
    def local_col(self) -> torch.Tensor:
        return self._axes[0][self._my_entries(self._axes[0])]

    @property
    def local_row(self) -> torch.Tensor:
        return self._axes[1][self._my_entries(self._axes[1])]

    @property
    def local_coo(self) -> torch.Tensor:
        return torch.stack([self.local_col, self.local_row])

    @property
    def shape(self) -> Tuple[int, int]:
        return tuple(axis.shape[0] for axis in self._axes)

    @property
    def dtype(self) -> torch.dtype:
        return self._axes[0].dtype
