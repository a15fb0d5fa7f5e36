"""``DistTensor`` / ``DistEmbedding`` — the tensors a FeatureStore keeps
(/root/reference/python/cugraph-pyg/cugraph_pyg/tensor/dist_tensor.py:20-534; helpers of tensor/utils.py:14-170):
``__getitem__`` gathers by global row, ``__setitem__`` scatters, sources are a tensor, a ``.pt`` / ``.npy`` file or a list
of headerless binary part files.  Storage is ``wholegraph_amd.WholeMemoryTensor``: one device tensor on a single GPU, a
node-local range partition + RCCL all-to-all otherwise (also runs over gloo for the CPU tests).

``device``: "cuda" (the default here: 288 GB of HBM per GPU hold every BASELINE table) keeps a rank's rows in HBM; "cpu"
is the reference's host-pinned placement (dist_tensor.py:60-75: pinned host memory the GPU reads through UVA) — the rows live
in PINNED HOST memory and the same HIP row kernels read and write them in place over PCIe (for tables that outgrow HBM;
on a box without a GPU it is plain host memory).  ``backend`` ("vmm" / "nccl" / "nvshmem" / "chunked") is accepted,
remembered and reported back, but does not select a memory type.  ``DistEmbedding(cache_policy=...)`` forwards the policy to
``wholegraph_amd.embedding.create_embedding`` like the reference (dist_tensor.py:385-399): the table is then a handle of the HIP
library (device or pinned-host partitions) behind its READONLY / READWRITE device cache.

Pinned-host placement and the stream: the row kernels write the pinned tensor on the current HIP stream; every method that
hands the host memory out, or copies into it from the host, waits for that stream first."""
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from wholegraph_amd import dist as _dist
from wholegraph_amd.tensor import HipLocalOps, WholeMemoryTensor, equal_entry_partition


class _Dim(int):
    """``tensor.dim`` is a property in the reference (dist_tensor.py:312-314) and a method on torch tensors: serve both."""

    def __call__(self):
        return int(self)


def _offsets_from_book(partition_book, rows, ws):
    if partition_book is None:
        return equal_entry_partition(rows, ws)
    book = [int(v) for v in partition_book]
    if len(book) != ws or sum(book) != rows or any(v < 0 for v in book):
        raise ValueError("partition_book must hold one entry count per rank and sum to shape[0]")
    return [0] + np.cumsum(book).tolist()


class DistTensor:
    """1-D or 2-D distributed tensor, range-partitioned over dim 0."""

    default_local_ops = HipLocalOps   # the product's row kernels; CPU tests inject the oracle's here

    def __init__(self, src: Optional[Union[torch.Tensor, str, List[str]]] = None, shape: Optional[Sequence[int]] = None,
                 dtype: Optional[torch.dtype] = None, device: Optional[str] = "cuda",
                 partition_book: Optional[Sequence[int]] = None, backend: Optional[str] = "nccl", *,
                 partition_offsets: Optional[Sequence[int]] = None, group=None, local_ops=None, **kwargs):
        self._tensor = None
        self._requested_device = device
        # "cpu" with a GPU present: pinned host rows, indices and results stay on the GPU
        self._host_rows = str(device).startswith("cpu") and torch.cuda.is_available()
        self._backend = backend
        self._group = group
        self._local_ops = local_ops if local_ops is not None else DistTensor.default_local_ops
        self._book = partition_book
        self._offsets_arg = partition_offsets
        if src is None:
            if shape is None:
                raise ValueError("Please specify the shape of the tensor.")
            if dtype is None:
                raise ValueError("Please specify the dtype of the tensor.")
            if len(shape) not in (1, 2):
                raise ValueError("The shape of the tensor must be 1D or 2D.")
            self._create(shape, dtype)
        elif isinstance(src, (list, tuple)):
            if shape is None or dtype is None:
                raise ValueError("For now, reading from multiple files is only supported with binary format.")
            self._create(shape, dtype)      # dist_tensor.py:82-90, utils.py:96-170
            self._tensor.from_filelist(list(src), int(kwargs.get("round_robin_size", 0) or 0))
        else:
            self._init_from_single_source(src)

    # ---- construction -----------------------------------------------------------------------------------------
    def _create(self, shape, dtype):
        shape = tuple(int(s) for s in shape)
        ws, rk = _dist.world_size(self._group), _dist.rank(self._group)
        dev = "cuda" if torch.cuda.is_available() else "cpu"

        def zeros(sz):
            return torch.zeros(sz, dtype=dtype, pin_memory=True) if self._host_rows else torch.zeros(sz, dtype=dtype, device=dev)
        if ws == 1:
            self._tensor = WholeMemoryTensor(zeros(shape), local_ops=self._local_ops)
        else:
            offs = list(self._offsets_arg) if self._offsets_arg is not None else _offsets_from_book(self._book, shape[0], ws)
            local = zeros((offs[rk + 1] - offs[rk],) + shape[1:])
            self._tensor = WholeMemoryTensor(local, global_rows=shape[0], partition_offsets=offs, group=self._group,
                                             local_ops=self._local_ops)
        self._dtype = dtype

    def _init_from_single_source(self, src):
        """A tensor every rank holds in full, or a ``.pt`` / ``.npy`` file (dist_tensor.py:98-157): each rank copies
        its own row range."""
        if isinstance(src, torch.Tensor):
            host_tensor = src if src.dim() >= 1 else src.view(1)
        elif isinstance(src, str) and src.endswith(".pt"):
            host_tensor = torch.load(src, mmap=True)
        elif isinstance(src, str) and src.endswith(".npy"):
            host_tensor = torch.from_numpy(np.load(src, mmap_mode="c"))
        else:
            raise ValueError("Unsupported source type. Please provide a torch.Tensor, a file path, or a list of file paths.")
        if host_tensor.dim() not in (1, 2):
            raise ValueError("The shape of the tensor must be 1D or 2D.")
        if _dist.world_size(self._group) == 1:
            # one GPU holds everything: adopt the tensor (no second copy of a table that may fill most of the HBM)
            dev = "cuda" if torch.cuda.is_available() else "cpu"
            rows = host_tensor.cpu().contiguous().pin_memory() if self._host_rows else host_tensor.to(dev).contiguous()
            self._tensor = WholeMemoryTensor(rows, local_ops=self._local_ops)
            self._dtype = host_tensor.dtype
            return
        self._create(host_tensor.shape, host_tensor.dtype)
        self.load_from_global_tensor(host_tensor)

    def load_from_global_tensor(self, tensor):
        """Every rank passes the WHOLE tensor and keeps its own rows (utils.py:14-19)."""
        if self._tensor is None:
            raise ValueError("Please create WholeGraph tensor first.")
        local, start = self._tensor.get_local_tensor()
        if tuple(tensor.shape) != tuple(self._tensor.shape):
            raise ValueError("The shape of the tensor does not match the shape of the distributed tensor.")
        self._host_fence()
        local.copy_(tensor[start:start + local.shape[0]].to(local.dtype))

    def load_from_local_tensor(self, tensor):
        """Every rank passes exactly its own rows (dist_tensor.py:172-190)."""
        if self._tensor is None:
            raise ValueError("Please create WholeGraph tensor first.")
        local = self._tensor.get_local_tensor()[0]
        if tuple(local.shape) != tuple(tensor.shape):
            raise ValueError("The shape of the tensor does not match the shape of the local tensor.")
        if self.dtype != tensor.dtype:
            raise ValueError("The dtype of the tensor does not match the dtype of the local tensor.")
        self._host_fence()
        local.copy_(tensor)

    @classmethod
    def from_tensor(cls, tensor: torch.Tensor, device: Optional[str] = "cuda", partition_book=None,
                    backend: Optional[str] = "nccl", **kwargs):
        return cls(src=tensor, device=device, partition_book=partition_book, backend=backend, **kwargs)

    @classmethod
    def from_file(cls, file_path: str, device: Optional[str] = "cuda", partition_book=None,
                  backend: Optional[str] = "nccl", **kwargs):
        return cls(src=file_path, device=device, partition_book=partition_book, backend=backend, **kwargs)

    # ---- access (collective over the group when partitioned) -------------------------------------------------------
    @property
    def _wm(self):   # the name the FeatureStore used before
        return self._tensor

    def _idx(self, idx):
        if isinstance(idx, slice):
            idx = torch.arange(*idx.indices(self._tensor.shape[0]))
        idx = torch.as_tensor(idx)
        if idx.dtype not in (torch.int32, torch.int64):
            idx = idx.long()
        where = "cuda" if self._host_rows else self._tensor.local_tensor.device   # pinned host rows: the kernels run on the GPU
        return idx.to(where).contiguous().view(-1)

    def __getitem__(self, idx) -> torch.Tensor:
        assert self._tensor is not None, "Please create WholeGraph tensor first."
        return self._tensor.gather(self._idx(idx))

    def gather_into(self, idx, out: torch.Tensor) -> torch.Tensor:
        """``out[:] = self[idx]`` without a fresh allocation (``out``: contiguous, ``len(idx)`` rows, the tensor's dtype) —
        what the loaders use to fill one preallocated call-group buffer piece by piece.  Collective like ``__getitem__``."""
        assert self._tensor is not None, "Please create WholeGraph tensor first."
        return self._tensor.gather(self._idx(idx), out=out)

    def __setitem__(self, idx, val: torch.Tensor):
        assert self._tensor is not None, "Please create WholeGraph tensor first."
        idx = self._idx(idx)
        val = val.to(device=idx.device, dtype=self.dtype)
        if val.dim() < self._tensor.dim():
            val = val.view((-1,) + tuple(self._tensor.shape[1:]))
        self._tensor.scatter(val.contiguous(), idx)

    def _host_fence(self):
        """Pinned-host rows are written by kernels queued on the current stream (``__setitem__``) and read or copied by the
        HOST here: wait for the stream, or a host read sees rows from before a scatter and a host copy races a queued
        kernel.  (Rows in HBM need nothing: torch orders every access on the stream.)"""
        if self._host_rows and torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()

    def get_local_tensor(self, host_view=False):
        self._host_fence()
        return self._tensor.get_local_tensor(host_view)[0]

    def get_local_offset(self):
        return self._tensor.get_local_tensor()[1]

    def get_comm(self):
        """The process group the rows are partitioned over (``None`` = the default group)."""
        assert self._tensor is not None, "Please create WholeGraph tensor first."
        return self._group

    @property
    def dim(self):
        return _Dim(self._tensor.dim())

    @property
    def shape(self):
        return torch.Size(self._tensor.shape)

    @property
    def device(self):
        return self._requested_device

    @property
    def dtype(self):
        return self._dtype

    def __len__(self):
        return self._tensor.shape[0]

    def __repr__(self):
        if self._tensor is None:
            return "<DistTensor: No tensor loaded>"
        return f"DistTensor(shape={tuple(self._tensor.shape)}, dtype={self.dtype}, device='{self.device}')"


class DistEmbedding(DistTensor):
    """2-D [num_embeddings, embedding_dim] table with a name (dist_tensor.py:340-534)."""

    def __init__(self, src=None, shape=None, dtype=None, device: Optional[str] = "cuda", partition_book=None,
                 backend: Optional[str] = "nccl", cache_policy=None, gather_sms: Optional[int] = -1,
                 round_robin_size: int = 0, name: Optional[str] = None, **kwargs):
        self._name = name
        self._gather_sms = gather_sms
        self._embedding = None
        if cache_policy is None:
            super().__init__(src, shape, dtype, device, partition_book, backend, round_robin_size=round_robin_size, **kwargs)
            return
        # A cached embedding is a handle of the HIP library (reference dist_tensor.py:385-399 hands the policy to
        # create_wg_dist_tensor -> pylibwholegraph create_embedding, tensor/utils.py:21-93): partitions in HBM or pinned host
        # memory, lookups through the policy's device cache, writes as scatters into the table.
        from wholegraph_amd import embedding as wge
        from wholegraph_amd.comm import get_global_communicator
        if not torch.cuda.is_available():
            raise RuntimeError("DistEmbedding(cache_policy=...) needs the HIP library's communicator (a GPU)")
        self._tensor, self._requested_device, self._backend, self._group = None, device, backend, kwargs.get("group")
        self._host_rows = False        # (the library owns the placement; indices and results live on the GPU)
        self._local_ops, self._book, self._offsets_arg = None, partition_book, None
        host = None
        if isinstance(src, (list, tuple)):
            if shape is None or dtype is None:
                raise ValueError("For now, reading from multiple files is only supported with binary format.")
        elif src is not None:
            if isinstance(src, torch.Tensor):
                host = src
            elif isinstance(src, str) and src.endswith(".pt"):
                host = torch.load(src, mmap=True)
            elif isinstance(src, str) and src.endswith(".npy"):
                host = torch.from_numpy(np.load(src, mmap_mode="c"))
            else:
                raise ValueError("Unsupported source type. Please provide a torch.Tensor, a file path, or a list of file paths.")
            shape, dtype = tuple(host.shape), host.dtype
        if shape is None or dtype is None:
            raise ValueError("Please specify the shape and dtype of the embedding.")
        if len(shape) != 2:
            raise ValueError("an embedding is a 2-D table")
        memory_type = "chunked" if backend in ("vmm", "chunked") else "distributed"
        location = "cpu" if str(device).startswith("cpu") else "cuda"
        self._embedding = wge.create_embedding(get_global_communicator(), memory_type, location, dtype, list(shape),
                                               cache_policy=cache_policy, embedding_entry_partition=partition_book,
                                               gather_sms=-1 if gather_sms is None else int(gather_sms),
                                               round_robin_size=int(round_robin_size or 0))
        self._tensor = self._embedding.get_embedding_tensor()
        self._dtype = dtype
        if isinstance(src, (list, tuple)):
            self._tensor.from_filelist(list(src), int(round_robin_size or 0))
        elif host is not None:
            local, start = self._tensor.get_local_tensor()
            local.copy_(host[start:start + local.shape[0]].to(local.dtype))
            if torch.cuda.is_available():
                torch.cuda.current_stream().synchronize()

    def __getitem__(self, idx) -> torch.Tensor:
        if self._embedding is None:
            return super().__getitem__(idx)
        return self._embedding.gather(self._idx(idx))          # through the policy's cache

    def gather_into(self, idx, out: torch.Tensor) -> torch.Tensor:
        if self._embedding is None:
            return super().gather_into(idx, out)
        out.copy_(self._embedding.gather(self._idx(idx)).view(out.shape))
        return out

    def __setitem__(self, idx, val: torch.Tensor):
        if self._embedding is None:
            return super().__setitem__(idx, val)
        idx = self._idx(idx)
        if self._embedding.wmb_cache_policy is not None:
            # BEFORE the scatter: dropping a read-write cache writes its modified lines back (wg_embedding.hip,
            # wholememory_embedding_drop_all_cache), so a line left dirty by an optimizer step would land on top of the rows
            # written here if the order were the other way round.  After the drop nothing is resident, nothing is dirty, and
            # no line can shadow the new rows.
            self._embedding.drop_all_cache()
        self._embedding.get_embedding_tensor().scatter(val.to(device=idx.device, dtype=self.dtype).contiguous(), idx)

    def _idx(self, idx):
        if self._embedding is None:
            return super()._idx(idx)
        if isinstance(idx, slice):
            idx = torch.arange(*idx.indices(self._tensor.shape[0]))
        idx = torch.as_tensor(idx)
        if idx.dtype not in (torch.int32, torch.int64):
            idx = idx.long()
        return idx.cuda().contiguous().view(-1)

    def get_local_tensor(self, host_view=False):
        if self._embedding is None:
            return super().get_local_tensor(host_view)
        return self._tensor.get_local_tensor(host_view)[0]

    def get_local_offset(self):
        if self._embedding is None:
            return super().get_local_offset()
        return self._tensor.get_local_tensor()[1]

    @classmethod
    def from_tensor(cls, tensor, device: Optional[str] = "cuda", partition_book=None, name: Optional[str] = None,
                    cache_policy=None, **kwargs):
        return cls(src=tensor, device=device, partition_book=partition_book, name=name, cache_policy=cache_policy, **kwargs)

    @classmethod
    def from_file(cls, file_path, device: Optional[str] = "cuda", partition_book=None, name: Optional[str] = None,
                  cache_policy=None, **kwargs):
        return cls(src=file_path, device=device, partition_book=partition_book, name=name, cache_policy=cache_policy,
                   **kwargs)

    @property
    def name(self):
        return self._name

    def __repr__(self):
        if self._tensor is None:
            return f"<DistEmbedding: No embedding loaded, Name: {self._name}>"
        head = f"DistEmbedding(name={self._name}, " if self._name else "DistEmbedding("
        return head + f"shape={tuple(self.shape)}, dtype={self.dtype}, device='{self.device}')"
