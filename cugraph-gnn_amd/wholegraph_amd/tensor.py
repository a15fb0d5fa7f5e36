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
import os
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


def unique_bounded(indice: torch.Tensor, id_bound: int, *, report_out_of_bound: bool = False):
    """``-> (distinct, inverse)``: the distinct non-negative ids of ``indice`` ASCENDING (int64) and, for every entry, its
    position in that list (int32; -1 for a negative id = a row to skip) — ``wgamd_unique_bounded`` (mark, scan over the
    bound, compact, look up; include/wgamd_ext.h).  One host synchronisation (the count).  An id >= ``id_bound`` raises —
    unless ``report_out_of_bound``: then such ids are left out of ``distinct`` (their ``inverse`` is -1) and the call
    returns ``(distinct, inverse, any_out_of_bound)``, for callers that must reach a collective before they may raise."""
    from .env import torch_dtype_to_wm
    assert indice.is_cuda and indice.dim() == 1 and indice.dtype in (torch.int32, torch.int64) and indice.is_contiguous()
    lib, dev, n = L.lib(), indice.device, int(indice.shape[0])
    nbytes = lib.wgamd_unique_bounded_workspace_bytes(int(id_bound))
    if nbytes == 0:
        raise ValueError("unique_bounded: id_bound %d is outside (0, 2^31 - 4096)" % id_bound)
    buf = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)   # (the caching allocator hands the same block back)
    ws = (buf, (-buf.data_ptr()) % 256, nbytes)
    distinct = torch.empty(min(n, int(id_bound)), dtype=torch.int64, device=dev)
    inverse = torch.empty(n, dtype=torch.int32, device=dev)
    info = torch.empty(2, dtype=torch.int32, device=dev)    # {distinct count, any id out of bound}
    L.check(lib.wgamd_unique_bounded(indice.data_ptr(), torch_dtype_to_wm(indice.dtype), n, int(id_bound), distinct.data_ptr(),
                                     inverse.data_ptr(), info.data_ptr(), info.data_ptr() + 4, ws[0].data_ptr() + ws[1], ws[2],
                                     get_stream()), "wgamd_unique_bounded")
    n_d, bad = (int(v) for v in info.cpu())
    if report_out_of_bound:
        return distinct[:n_d], inverse, bool(bad)
    if bad:
        raise IndexError("gather: an index is >= the table's %d rows" % id_bound)
    return distinct[:n_d], inverse


def unique_bounded_nosync(indice: torch.Tensor, n_live_dev: torch.Tensor, id_bound: int):
    """``unique_bounded`` over a CAPACITY-sized id list whose live length is a device int (``n_live_dev``: a 1-element int32
    view, e.g. ``counts[k][1:2]`` of a no-sync walk), WITHOUT a host synchronisation: ``-> (distinct [min(cap, id_bound)] int64,
    inverse [cap] int32, info [2] int32 = {distinct count, any id out of bound})`` — entries past the live counts are
    unwritten.  Enqueue it behind the walk, copy ``info`` back with the walk's own sizes."""
    from .env import torch_dtype_to_wm
    assert indice.is_cuda and indice.dim() == 1 and indice.dtype in (torch.int32, torch.int64) and indice.is_contiguous()
    assert n_live_dev.is_cuda and n_live_dev.dtype == torch.int32 and n_live_dev.numel() == 1
    lib, dev, n = L.lib(), indice.device, int(indice.shape[0])
    nbytes = lib.wgamd_unique_bounded_workspace_bytes(int(id_bound))
    if nbytes == 0:
        raise ValueError("unique_bounded: id_bound %d is outside (0, 2^31 - 4096)" % id_bound)
    buf = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    distinct = torch.empty(min(n, int(id_bound)), dtype=torch.int64, device=dev)
    inverse = torch.empty(n, dtype=torch.int32, device=dev)
    info = torch.empty(2, dtype=torch.int32, device=dev)
    L.check(lib.wgamd_unique_bounded_live(indice.data_ptr(), torch_dtype_to_wm(indice.dtype), n, n_live_dev.data_ptr(), int(id_bound),
                                          distinct.data_ptr(), inverse.data_ptr(), info.data_ptr(), info.data_ptr() + 4,
                                          buf.data_ptr() + (-buf.data_ptr()) % 256, nbytes, get_stream()), "wgamd_unique_bounded_live")
    return distinct, inverse, info


def dedup_pays(n: int, rows: int, world: int) -> bool:
    """The ``dedup="auto"`` rule of the partitioned gathers: more than one rank (a repeat costs wire bytes only then) and
    an id list that is large next to the table (a call group of mini-batches: 10.9 M ids into the 2.45 M rows of products —
    at most a quarter of them can be distinct; the scan over the bound is then noise next to the exchange it shortens)."""
    import os
    force = os.environ.get("WGAMD_GATHER_DEDUP")   # "0" / "1": measurement switch, overrides the rule for every rank
    if force in ("0", "1"):
        return force == "1" and 0 < rows < (1 << 31) - 4096 and n > 0
    return world > 1 and 0 < rows < (1 << 31) - 4096 and n >= max(rows // 8, 1024)


def gather_distinct(gather_rows, indice: torch.Tensor, rows: int, out: torch.Tensor):
    """``out[i] = table[indice[i]]`` fetching every DISTINCT row once: ``gather_rows(ids) -> [len(ids), dim]`` is the
    (collective) fetch, the expansion ``out[i] = fetched[inverse[i]]`` a local row copy (negative ids leave their row alone)."""
    # ``gather_rows`` is a COLLECTIVE on a partitioned table: a rank holding a bad id must still enter it (with the bad ids
    # left out), or every other rank waits in the exchange for ever; it raises afterwards, like the plain path, which
    # also meets the error inside / behind the collective.
    distinct, inverse, bad = unique_bounded(indice, rows, report_out_of_bound=True)
    fetched = gather_rows(distinct)
    if bad:
        raise IndexError("gather: an index is >= the table's %d rows" % rows)
    if indice.shape[0] > 0 and fetched.shape[0] > 0:
        local_gather(fetched if fetched.dim() == 2 else fetched.unsqueeze(1), inverse, out if out.dim() == 2 else out.unsqueeze(1))
    return out


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
    def gather(self, indice: torch.Tensor, *, force_dtype: Union[torch.dtype, None] = None, dedup="auto",
               out: Optional[torch.Tensor] = None):
        """``dedup`` (partitioned tables): fetch every DISTINCT row once and expand locally — True / False / "auto"
        (``dedup_pays``).  Same result either way; every rank still makes exactly one collective fetch.  ``out``: a
        contiguous ``[len(indice), dim]`` (or ``[len(indice)]``) tensor to fill instead of a fresh allocation."""
        assert indice.dim() == 1
        embedding_dim = self.shape[1] if self.dim() == 2 else 1
        output_dtype = force_dtype if force_dtype is not None else self.dtype
        if out is not None:
            assert out.is_contiguous() and out.shape[0] == indice.shape[0] and out.numel() == indice.shape[0] * embedding_dim
            output_tensor = out.view(indice.shape[0], embedding_dim)
        else:
            output_tensor = torch.empty([indice.shape[0], embedding_dim], device=indice.device, dtype=output_dtype,
                                        requires_grad=False)
        table2d = self.local_tensor if self.dim() == 2 else self.local_tensor.unsqueeze(1)
        if self.is_distributed and indice.is_cuda and self.local_ops is HipLocalOps and (
                dedup is True or (dedup == "auto" and dedup_pays(indice.shape[0], self._rows, _dist.world_size(self.group)))):
            gather_distinct(lambda ids: self.gather(ids, force_dtype=force_dtype, dedup=False), indice.contiguous(), self._rows,
                            output_tensor)
            return output_tensor.view(-1) if self.dim() == 1 else output_tensor
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


def get_part_file_name(prefix: str, part_id: int, part_count: int):
    """Part-file naming of the reference (torch/utils.py:180-188)."""
    return "%s_part_%d_of_%d" % (prefix, part_id, part_count)


def get_part_file_list(prefix: str, part_count: int):
    return [get_part_file_name(prefix, i, part_count) for i in range(part_count)]


class _DevicePointerView:
    """``__cuda_array_interface__`` carrier so torch can view memory owned by a wholememory handle."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner  # keeps the handle alive as long as any view of it lives


_TYPESTR = {torch.float32: "<f4", torch.float64: "<f8", torch.float16: "<f2", torch.bfloat16: "<i2",
            torch.int32: "<i4", torch.int64: "<i8", torch.int16: "<i2", torch.int8: "|i1"}


class DistributedWholeMemoryTensor(object):
    r"""A table whose storage is a C-level ``WHOLEMEMORY_MT_DISTRIBUTED`` handle (include/wgamd_comm.h):
    rank r holds a contiguous row range in its own HBM; ``gather``/``scatter`` run the RCCL all-to-all
    pipeline INSIDE the library (csrc/wg_comm.hip) and are collective over the communicator.  Same methods as
    ``pylibwholegraph.torch.tensor.WholeMemoryTensor`` (tensor.py:24-123)."""

    def __init__(self, c_tensor, comm, dtype=None, shape=None, owner=True):
        import ctypes
        from .env import wm_dtype_to_torch
        self.c = ctypes.c_void_p(c_tensor)
        self.comm = comm
        d = L.lib().wholememory_tensor_get_tensor_description(self.c).contents
        self._dtype = dtype if dtype is not None else wm_dtype_to_torch(d.dtype)
        self._shape = tuple(int(d.sizes[i]) for i in range(d.dim)) if shape is None else tuple(int(v) for v in shape)
        # a view of a wider table (embedding tables are padded to 16 B; optimizer states share one table)
        self._row_stride = int(d.strides[0]) if d.dim == 2 else 1
        self._col0 = int(d.storage_offset) % self._row_stride if d.dim == 2 else 0
        assert d.dim == 1 or int(d.storage_offset) < self._row_stride, "row-offset views are not exposed to torch"
        self._owner = owner  # False: the C tensor belongs to an embedding (destroyed with it)
        self._local_view = None

    @property
    def dtype(self):
        return self._dtype

    def dim(self):
        return len(self._shape)

    @property
    def shape(self):
        return self._shape

    def stride(self):
        return (self._shape[1], 1) if self.dim() == 2 else (1,)

    def storage_offset(self):
        return 0

    @property
    def is_distributed(self):
        return True

    def get_comm(self):
        return self.comm

    @property
    def local_tensor(self):
        return self.get_local_tensor()[0]

    def get_sub_tensor(self, starts, ends):
        """View [starts, ends) of the same storage (-1 = whole dim) — wholememory_tensor_get_subtensor,
        binding ``PyWholeMemoryTensor.get_sub_tensor`` (wholememory_binding.pyx:1381-1404).  Column ranges only (every
        rank keeps its row range); destroy the view before the root tensor."""
        import ctypes
        assert len(starts) == len(ends) == self.dim()
        assert int(starts[0]) in (-1, 0) and int(ends[0]) in (-1, self._shape[0]), "row ranges are not supported"
        st = (ctypes.c_int64 * self.dim())(*[max(int(v), 0) for v in starts])
        en = (ctypes.c_int64 * self.dim())(*[int(v) for v in ends])
        c = ctypes.c_void_p()
        L.check(L.lib().wholememory_tensor_get_subtensor(self.c, st, en, ctypes.byref(c)), "wholememory_tensor_get_subtensor")
        return DistributedWholeMemoryTensor(c.value, self.comm)

    def get_local_tensor(self, host_view: bool = False):
        """(torch view of this rank's rows, first global row held here) — tensor.py:106-123."""
        import ctypes
        lib = L.lib()
        n, start = ctypes.c_size_t(0), ctypes.c_size_t(0)
        L.check(lib.wholememory_tensor_get_local_entry_count(ctypes.byref(n), self.c), "tensor_get_local_entry_count")
        L.check(lib.wholememory_tensor_get_local_entry_start(ctypes.byref(start), self.c),
                "tensor_get_local_entry_start")
        if self._local_view is None:
            ptr, size, off = ctypes.c_void_p(), ctypes.c_size_t(0), ctypes.c_size_t(0)
            handle = ctypes.c_void_p(lib.wholememory_tensor_get_memory_handle(self.c))
            L.check(lib.wholememory_get_local_memory(ctypes.byref(ptr), ctypes.byref(size), ctypes.byref(off), handle),
                    "wholememory_get_local_memory")
            shape = (n.value,) + self._shape[1:]
            if n.value == 0:
                self._local_view = torch.empty(shape, dtype=self._dtype, device="cuda")
            elif lib.wholememory_get_memory_location(handle) == L.ML_HOST:
                # pinned host partition (WHOLEMEMORY_ML_HOST): a CPU tensor over the very bytes the GPU reads in place
                full = (n.value, self._row_stride) if self.dim() == 2 else shape
                es = torch.empty((), dtype=self._dtype).element_size()
                count = full[0] * (full[1] if self.dim() == 2 else 1)
                raw = (ctypes.c_char * (count * es)).from_address(ptr.value)
                view = torch.frombuffer(raw, dtype=self._dtype, count=count).view(full)
                view._wg_owner = self
                self._local_view = view[:, self._col0:self._col0 + shape[1]] if self.dim() == 2 else view
            else:
                full = (n.value, self._row_stride) if self.dim() == 2 else shape
                view = torch.as_tensor(_DevicePointerView(ptr.value, full, _TYPESTR[self._dtype], self),
                                       device="cuda")
                view = view.view(self._dtype) if view.dtype != self._dtype else view
                self._local_view = view[:, self._col0:self._col0 + shape[1]] if self.dim() == 2 else view
        t = self._local_view
        return (t.cpu() if host_view else t), start.value

    def memory_location(self) -> str:
        import ctypes
        handle = ctypes.c_void_p(L.lib().wholememory_tensor_get_memory_handle(self.c))
        return "cpu" if L.lib().wholememory_get_memory_location(handle) == L.ML_HOST else "cuda"

    def memory_type(self) -> str:
        import ctypes
        handle = ctypes.c_void_p(L.lib().wholememory_tensor_get_memory_handle(self.c))
        code = L.lib().wholememory_get_memory_type(handle)
        return {L.MT_CONTINUOUS: "continuous", L.MT_CHUNKED: "chunked", L.MT_DISTRIBUTED: "distributed",
                L.MT_HIERARCHY: "hierarchy"}.get(code, "none")

    def fetch_path(self) -> str:
        """How a remote row reaches this rank (reported by bench.py next to the partitioned result)."""
        if self.comm.get_size() == 1:
            return "local rows only (single-rank communicator)"
        if self.memory_type() in ("chunked", "continuous"):
            return "peer-mapped loads over xGMI (HIP IPC, one kernel, no host sync)"
        return "all-to-all-v (RCCL send/recv groups, one host sync)"

    def gather(self, indice: torch.Tensor, *, force_dtype: Union[torch.dtype, None] = None, out: torch.Tensor = None,
               dedup="auto"):
        """``dedup``: fetch every DISTINCT row through the exchange once and expand locally (``gather_distinct``) — True /
        False / "auto" (``dedup_pays``: more than one rank and an id list that is large next to the table, i.e. a call
        group).  Same result; still exactly one collective ``wholememory_gather`` per rank and call."""
        assert indice.dim() == 1
        embedding_dim = self._shape[1] if self.dim() == 2 else 1
        if out is None:
            out = torch.empty([indice.shape[0], embedding_dim] if self.dim() == 2 else [indice.shape[0]],
                              device=indice.device, dtype=force_dtype if force_dtype is not None else self._dtype)
        if dedup is True or (dedup == "auto" and dedup_pays(indice.shape[0], self._shape[0], self.comm.get_size())):
            return gather_distinct(lambda ids: self.gather(ids, force_dtype=out.dtype, dedup=False), indice.contiguous(),
                                   self._shape[0], out)
        w_i, w_o = wrap_torch_tensor(indice), wrap_torch_tensor(out)
        L.check(L.lib().wholememory_gather(self.c, w_i.c, w_o.c, get_wholegraph_env_fns(), get_stream(), -1),
                "wholememory_gather")
        return out

    def scatter(self, input_tensor: torch.Tensor, indice: torch.Tensor):
        assert indice.dim() == 1 and input_tensor.dim() == self.dim()
        assert indice.shape[0] == input_tensor.shape[0]
        w_in, w_i = wrap_torch_tensor(input_tensor), wrap_torch_tensor(indice)
        L.check(L.lib().wholememory_scatter(w_in.c, w_i.c, self.c, get_wholegraph_env_fns(), get_stream(), -1),
                "wholememory_scatter")

    # ---- binary file I/O (tensor.py:153-198; format: headerless row-major entries) -------------------------
    def _entry_bytes(self):
        row = self._shape[1] if self.dim() == 2 else 1
        return row * torch.empty((), dtype=self._dtype).element_size()

    def _memory_layout(self):
        """(byte offset of column 0 inside a stored row, stored row bytes, bytes of one row of THIS tensor)."""
        es = torch.empty((), dtype=self._dtype).element_size()
        return self._col0 * es, self._row_stride * es, self._entry_bytes()

    def from_filelist(self, filelist, round_robin_size: int = 0):
        """Collective: the files, read as one concatenated array of rows, fill the tensor (rank-local rows only
        are read by each rank); ``round_robin_size`` > 0 deals blocks of that many rows to the ranks in turn."""
        import ctypes
        if isinstance(filelist, str):
            filelist = [filelist]
        names = (ctypes.c_char_p * len(filelist))(*[os.fsencode(f) for f in filelist])
        handle = ctypes.c_void_p(L.lib().wholememory_tensor_get_memory_handle(self.c))
        off, stride, eb = self._memory_layout()
        L.check(L.lib().wholememory_load_from_file(handle, off, stride, eb, names, len(filelist), int(round_robin_size)),
                "wholememory_load_from_file")

    def from_file_prefix(self, file_prefix: str, part_count: Union[int, None] = None):
        if part_count is None:
            part_count = self.comm.get_size()
        self.from_filelist(get_part_file_list(file_prefix, part_count))

    def local_to_file(self, filename: str):
        """Collective: every rank writes its own rows to its own file."""
        import ctypes
        handle = ctypes.c_void_p(L.lib().wholememory_tensor_get_memory_handle(self.c))
        off, stride, eb = self._memory_layout()
        L.check(L.lib().wholememory_store_to_file(handle, off, stride, eb, os.fsencode(filename)),
                "wholememory_store_to_file")

    def to_file_prefix(self, file_prefix: str):
        self.local_to_file(get_part_file_name(file_prefix, self.comm.get_rank(), self.comm.get_size()))

    def destroy(self):
        if self.c is not None and self.c.value:
            self._local_view = None
            if self._owner:
                L.check(L.lib().wholememory_destroy_tensor(self.c), "wholememory_destroy_tensor")
            self.c = None


def _create_handle_tensor(comm, memory_type, memory_location, sizes, dtype, strides, tensor_entry_partition):
    """Reference signature create_wholememory_tensor(comm, memory_type, memory_location, sizes, dtype,
    strides, tensor_entry_partition) — tensor.py:200-247."""
    import ctypes
    from .comm import memory_location_code, memory_type_code
    from .env import torch_dtype_to_wm
    sizes = [int(v) for v in sizes]
    assert len(sizes) in (1, 2), "sizes should be 1D or 2D"
    if strides is None:
        strides = [sizes[1], 1] if len(sizes) == 2 else [1]
    desc = L.TensorDescription()
    L.lib().wholememory_initialize_tensor_desc(ctypes.byref(desc))
    desc.dim = len(sizes)
    for i, (n, st) in enumerate(zip(sizes, strides)):
        desc.sizes[i], desc.strides[i] = n, int(st)
    desc.dtype = torch_dtype_to_wm(dtype)
    part = None
    if tensor_entry_partition is not None:
        part = (ctypes.c_size_t * len(tensor_entry_partition))(*[int(v) for v in tensor_entry_partition])
    c = ctypes.c_void_p()
    L.check(L.lib().wholememory_create_tensor(ctypes.byref(c), ctypes.byref(desc), comm.c_comm,
                                              memory_type_code(memory_type), memory_location_code(memory_location),
                                              part),
            "wholememory_create_tensor")
    return DistributedWholeMemoryTensor(c.value, comm, dtype, sizes)


def destroy_wholememory_tensor(wm_tensor):
    """tensor.py:322-328."""
    if isinstance(wm_tensor, DistributedWholeMemoryTensor):
        wm_tensor.destroy()


def _torch_from_filelist(self, filelist, round_robin_size: int = 0):
    """File loading for the torch-backed table (same format and sharding rules as the C entry point; each rank
    reads only its own rows with numpy memory maps)."""
    import numpy as np
    if isinstance(filelist, str):
        filelist = [filelist]
    local = self.local_tensor
    row_shape = tuple(local.shape[1:])
    np_dtype = torch.empty((), dtype=torch.int16 if self.dtype == torch.bfloat16 else self.dtype).numpy().dtype
    per_row = int(np.prod(row_shape)) if row_shape else 1
    maps = [np.memmap(f, dtype=np_dtype, mode="r").reshape(-1, per_row) if os.path.getsize(f) else
            np.empty((0, per_row), np_dtype) for f in filelist]  # an empty part (a rank without rows) cannot be mapped
    first = np.concatenate([[0], np.cumsum([m.shape[0] for m in maps])])
    total = int(first[-1])
    assert total <= self._rows, f"the files hold {total} rows, the tensor only {self._rows}"
    ws, rk = _dist.world_size(self.group), _dist.rank(self.group)
    start = self.partition_offsets[rk] if self.is_distributed else 0

    def read(lo, hi):  # rows [lo, hi) of the concatenated array
        parts = []
        while lo < hi:
            f = int(np.searchsorted(first, lo, side="right")) - 1
            take = min(hi, int(first[f + 1])) - lo
            parts.append(np.asarray(maps[f][lo - int(first[f]): lo - int(first[f]) + take]))
            lo += take
        arr = np.concatenate(parts) if parts else np.empty((0, per_row), np_dtype)
        t = torch.from_numpy(arr.reshape((-1,) + row_shape))
        return t.view(self.dtype) if self.dtype == torch.bfloat16 else t

    if round_robin_size == 0:
        lo, hi = min(start, total), min(start + local.shape[0], total)
        if hi > lo:
            local[: hi - lo] = read(lo, hi).to(local.device)
    else:
        rr, k = int(round_robin_size), 0
        while (k * ws + rk) * rr < total:
            g = (k * ws + rk) * rr
            cnt = min(rr, total - g)
            assert k * rr + cnt <= local.shape[0], f"round-robin shard of rank {rk} does not fit its {local.shape[0]} rows"
            local[k * rr: k * rr + cnt] = read(g, g + cnt).to(local.device)
            k += 1


def _torch_local_to_file(self, filename: str):
    t = self.local_tensor.detach().cpu().contiguous()
    (t.view(torch.int16) if t.dtype == torch.bfloat16 else t).numpy().tofile(filename)


WholeMemoryTensor.from_filelist = _torch_from_filelist
WholeMemoryTensor.from_file_prefix = lambda self, prefix, part_count=None: _torch_from_filelist(
    self, get_part_file_list(prefix, part_count if part_count is not None else _dist.world_size(self.group)))
WholeMemoryTensor.local_to_file = _torch_local_to_file
WholeMemoryTensor.to_file_prefix = lambda self, prefix: _torch_local_to_file(
    self, get_part_file_name(prefix, _dist.rank(self.group), _dist.world_size(self.group)))


def equal_entry_partition(total_rows: int, world_size: int):
    """Row offsets of the equal range partition: per = ceil(V/W), rank r owns
    [min(r*per, V), min((r+1)*per, V))
    (/root/reference/cpp/src/wholememory/memory_handle.cpp:1613-1629; public
    wholememory_equal_entry_partition_plan, cpp/include/wholememory/wholememory.h:380)."""
    per = (total_rows + world_size - 1) // world_size
    return [min(r * per, total_rows) for r in range(world_size + 1)]


def create_wholememory_tensor(*args, device=None, group=None, partition_offsets=None, local_ops=HipLocalOps,
                              **kwargs):
    """Allocate a (possibly range-partitioned) table.

    Two call forms:
      * the reference's ``(comm, memory_type, memory_location, sizes, dtype, strides,
        tensor_entry_partition=None)`` with a ``comm.WholeMemoryCommunicator`` first (tensor.py:200-247): the
        storage is a C-level DISTRIBUTED handle and gather/scatter exchange rows inside the library;
      * ``(shape, dtype, *, device, group, partition_offsets)``: torch tensors per rank, exchange over
        ``torch.distributed`` (``dist.py``; works with the gloo backend on CPU for tests).
    With more than one rank every rank allocates only its own slice."""
    from .comm import WholeMemoryCommunicator
    first = args[0] if args else kwargs.get("comm")
    if isinstance(first, WholeMemoryCommunicator):
        names = ["comm", "memory_type", "memory_location", "sizes", "dtype", "strides", "tensor_entry_partition"]
        bound = dict(zip(names, args))
        bound.update(kwargs)
        return _create_handle_tensor(bound["comm"], bound["memory_type"], bound["memory_location"], bound["sizes"],
                                     bound["dtype"], bound.get("strides"), bound.get("tensor_entry_partition"))
    shape = args[0] if args else kwargs["shape"]
    dtype = args[1] if len(args) > 1 else kwargs["dtype"]
    shape = tuple(shape)
    device = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
    ws = _dist.world_size(group)
    if ws == 1 and partition_offsets is None:
        return WholeMemoryTensor(torch.empty(shape, dtype=dtype, device=device), local_ops=local_ops)
    offs = list(partition_offsets) if partition_offsets is not None else equal_entry_partition(shape[0], ws)
    r = _dist.rank(group)
    local = torch.empty((offs[r + 1] - offs[r],) + shape[1:], dtype=dtype, device=device)
    return WholeMemoryTensor(local, global_rows=shape[0], partition_offsets=offs, group=group, local_ops=local_ops)
