"""Trainable embedding tables with sparse optimizers.

Public surface = that of /root/reference/python/pylibwholegraph/pylibwholegraph/torch/embedding.py
(``create_embedding``, ``create_embedding_from_filelist``, ``create_wholememory_optimizer``,
``create_wholememory_cache_policy``, ``create_builtin_cache_policy``, ``WholeMemoryEmbeddingModule``, the ``destroy_*``
functions), written from that contract over the C entry points of include/wgamd_embedding.h:

* the table is a handle of the library — rows range-partitioned over the GPUs of the communicator, all of it in HBM;
  ``gather`` is the feature-fetch exchange, and a training step is ONE collective call
  (``wholememory_embedding_gather_gradient_apply``: every (row, gradient) pair goes to the row's owner, duplicates are
  summed there in a fixed order and the optimizer formula runs in the same kernel, csrc/wg_embedding.hip);
* what autograd hands back between two optimizer steps is parked in a ``_PendingGradients`` buffer — one pre-sized
  (indices, rows) pair that contributions are copied into, a single contribution is kept as it came — and leaves it
  in one piece at ``step``;
* a READONLY cache policy is a NAME for "keep hot remote rows in my own HBM": the resolution from the reference's
  builtin names to (communicator, memory type) is the table ``_BUILTIN_CACHES``;
* a READWRITE policy on the table's own communicator is the reference's device cache in front of a host-resident table
  (``memory_location="cpu"``: pinned host memory the GPU reads in place): every rank keeps a write-back cache of ITS OWN
  rows — embedding row and optimizer state behind one tag — in HBM; ``writeback_all_cache`` / ``drop_all_cache`` flush it
  (csrc/wg_embedding.hip, "READWRITE device cache").
"""
import ctypes
import os
from typing import Callable, Dict, List, NamedTuple, Optional, Sequence, Union

import torch

from . import _lib as L
from .comm import (WholeMemoryCommunicator, get_global_communicator, get_local_device_communicator,
                   get_local_node_communicator, memory_location_code, memory_type_code)
from .env import get_stream, get_wholegraph_env_fns, torch_dtype_to_wm, wrap_torch_tensor
from .tensor import DistributedWholeMemoryTensor

# wholememory_optimizer_type_t / wholememory_access_type_t (include/wgamd_embedding.h; reference embedding.h:33-39)
_OPTIMIZER_CODE = {"sgd": 1, "adam": 2, "lazy_adam": 2, "rmsprop": 3, "adagrad": 4}
_ACCESS_CODE = {"readonly": 1, "readwrite": 2}
_MEMORY_TYPES = ("continuous", "chunked", "distributed", "hierarchy")
_MEMORY_LOCATIONS = ("cpu", "cuda")


def _stream() -> int:
    return int(get_stream().value or 0)


def _call(name: str, *args):
    """One C entry point, return code turned into WholeMemoryError."""
    L.check(getattr(L.lib(), name)(*args), name)


# ------------------------------------------------------------------------------------------------------------------
# gradients waiting for the next optimizer step
# ------------------------------------------------------------------------------------------------------------------
class _PendingGradients:
    """(row ids, fp32 gradient rows) collected since the last step.

    The usual training step contributes once per embedding; that pair is kept by reference and handed to the library as
    it is.  From the second contribution on, everything lives in one pre-sized buffer pair that doubles when it runs out
    — no Python list of tensors, no concatenation at ``step`` time."""

    def __init__(self, width: int):
        self.width = int(width)
        self._single = None                 # (ids, rows) of a lone contribution, not copied
        self._ids = self._rows = None       # the buffers
        self._used = 0

    def __len__(self):
        return self._used if self._single is None else int(self._single[0].shape[0])

    def __bool__(self):
        return self._single is not None or self._used > 0

    def clear(self):
        self._single, self._used = None, 0

    def _append(self, ids: torch.Tensor, rows: torch.Tensor):
        n, width = int(ids.shape[0]), int(rows.shape[1])
        if self._used and width != int(self._rows.shape[1]):
            raise ValueError("gradient rows of different widths in one step")
        need = self._used + n
        id_dtype = ids.dtype if (not self._used or self._ids.dtype == ids.dtype) else torch.int64
        fits = (self._ids is not None and int(self._ids.shape[0]) >= need and self._ids.dtype == id_dtype
                and int(self._rows.shape[1]) == width)
        if not fits:   # (a wrong width is the library's error to report at apply time, not a copy failure here)
            cap = max(need, 2 * (0 if self._ids is None else int(self._ids.shape[0])), 1024)
            new_ids = torch.empty(cap, dtype=id_dtype, device=ids.device)
            new_rows = torch.empty((cap, width), dtype=torch.float32, device=ids.device)
            if self._used:
                new_ids[:self._used] = self._ids[:self._used]
                new_rows[:self._used] = self._rows[:self._used]
            self._ids, self._rows = new_ids, new_rows
        self._ids[self._used:need] = ids
        self._rows[self._used:need] = rows
        self._used = need

    def add(self, ids: torch.Tensor, rows: torch.Tensor):
        assert ids.dim() == 1 and rows.dim() == 2 and ids.shape[0] == rows.shape[0]
        if self._single is None and self._used == 0:
            self._single = (ids, rows)
            return
        if self._single is not None:
            first, self._single = self._single, None
            self._append(first[0], first[1])
        self._append(ids, rows)

    def take(self, device):
        """(ids, fp32 rows) of everything collected; an empty pair when nothing was."""
        if self._single is not None:
            ids, rows = self._single
            return ids.contiguous(), rows.to(torch.float32).contiguous()
        if self._used:
            return self._ids[:self._used], self._rows[:self._used]
        return (torch.empty(0, dtype=torch.int64, device=device),
                torch.empty((0, self.width), dtype=torch.float32, device=device))


# ------------------------------------------------------------------------------------------------------------------
# cache policies
# ------------------------------------------------------------------------------------------------------------------
class WholeMemoryCachePolicy:
    """Handle of a cache policy (reference embedding.py:71-80).  Built by :func:`create_wholememory_cache_policy` or
    :func:`create_builtin_cache_policy`."""

    def __init__(self, c_policy, access_type: str):
        self.c_policy, self.access_type = c_policy, access_type


def create_wholememory_cache_policy(cache_comm: WholeMemoryCommunicator, *, memory_type: str = "chunked",
                                    memory_location: str = "cuda", access_type: str = "readonly", ratio: float = 0.5):
    """Reference embedding.py:83-110.  Only records the request; :func:`create_embedding` decides what it means here
    (READONLY: a private per-GPU cache of ``ratio * entries`` rows; READWRITE, on the table's communicator with
    ``memory_location="cuda"``: the write-back cache of every rank's own rows)."""
    handle = ctypes.c_void_p()
    _call("wholememory_create_embedding_cache_policy", ctypes.byref(handle), cache_comm.c_comm,
          memory_type_code(memory_type), memory_location_code(memory_location), _ACCESS_CODE[access_type], float(ratio))
    return WholeMemoryCachePolicy(handle, access_type)


def destroy_wholememory_cache_policy(cache_policy: Optional[WholeMemoryCachePolicy]):
    """Reference embedding.py:113-121; ``None`` and an already destroyed policy are fine."""
    if cache_policy is None or cache_policy.c_policy is None:
        return
    _call("wholememory_destroy_embedding_cache_policy", cache_policy.c_policy)
    cache_policy.c_policy = None


class _BuiltinCache(NamedTuple):
    communicator: Callable[[], WholeMemoryCommunicator]   # which ranks the reference would spread the cache over
    memory_type: Callable[[str, str], str]                # (requested cache type or "", embedding type) -> cache type


# reference embedding.py:124-216, as a table: builtin name -> how its communicator and memory type are picked
_BUILTIN_CACHES: Dict[str, _BuiltinCache] = {
    "all_devices": _BuiltinCache(get_global_communicator, lambda asked, emb: asked or emb),
    "local_node": _BuiltinCache(get_local_node_communicator, lambda asked, emb: asked or "chunked"),
    "local_device": _BuiltinCache(get_local_device_communicator, lambda asked, emb: "continuous"),
}


def create_builtin_cache_policy(builtin_cache_type: str, embedding_memory_type: str, embedding_memory_location: str,
                                access_type: str, cache_ratio: float, *, cache_memory_type: str = "",
                                cache_memory_location: str = ""):
    """``"none"`` gives ``None``; the other builtin names resolve through ``_BUILTIN_CACHES`` (on this target every GPU
    keeps its own cache lines whichever communicator the name stands for)."""
    if embedding_memory_type not in _MEMORY_TYPES:
        raise ValueError(f"embedding_memory_type={embedding_memory_type} is not valid")
    if embedding_memory_location not in _MEMORY_LOCATIONS:
        raise ValueError(f"embedding_memory_location={embedding_memory_location} is not valid")
    if builtin_cache_type == "none":
        return None
    if cache_memory_location not in ("",) + _MEMORY_LOCATIONS:
        raise ValueError(f"cache_memory_location is {cache_memory_location}, should be empty or cpu, cuda")
    entry = _BUILTIN_CACHES.get(builtin_cache_type)
    if entry is None:
        raise ValueError(f"builtin_cache_type={builtin_cache_type} not supported, should be none, "
                         + ", ".join(sorted(_BUILTIN_CACHES)))
    return create_wholememory_cache_policy(entry.communicator(),
                                           memory_type=entry.memory_type(cache_memory_type, embedding_memory_type),
                                           memory_location=cache_memory_location or "cuda", access_type=access_type,
                                           ratio=cache_ratio)


# ------------------------------------------------------------------------------------------------------------------
# the embedding
# ------------------------------------------------------------------------------------------------------------------
class WholeMemoryEmbedding:
    """One embedding table (reference embedding.py:275-407: ``gather``, ``add_gradients``, ``apply_gradients``, the tensor
    and optimizer-state views, ``save`` / ``load``, cache control)."""

    def __init__(self, c_embedding, comm: WholeMemoryCommunicator, cache_policy: Optional[WholeMemoryCachePolicy] = None):
        self.c_embedding, self.comm = c_embedding, comm
        self.wmb_cache_policy = cache_policy
        self.adjust_cache = cache_policy is not None      # a cached embedding learns from its misses by default
        self.wm_optimizer: Optional["WholeMemoryOptimizer"] = None
        # autograd needs SOME leaf that requires grad for backward() to reach the lookup (the indices are integers)
        self.dummy_input = torch.nn.Parameter(torch.zeros(1), requires_grad=False)
        self._views: Dict[str, DistributedWholeMemoryTensor] = {}     # "" = the table, else an optimizer state by name
        self._pending: Optional[_PendingGradients] = None
        self._touched = False       # a training-mode gather happened since the last step

    # ---- the table and its states as tensors -------------------------------------------------------------------
    def _view(self, key: str) -> DistributedWholeMemoryTensor:
        if key not in self._views:
            lib = L.lib()
            c = (lib.wholememory_embedding_get_embedding_tensor(self.c_embedding) if key == "" else
                 lib.wholememory_embedding_get_optimizer_state(self.c_embedding, key.encode()))
            if not c:
                raise KeyError(key)
            self._views[key] = DistributedWholeMemoryTensor(c, self.comm, owner=False)
        return self._views[key]

    def get_embedding_tensor(self) -> DistributedWholeMemoryTensor:
        return self._view("")

    def get_optimizer_state(self, state_name: str) -> DistributedWholeMemoryTensor:
        return self._view(state_name)

    def get_optimizer_state_names(self) -> List[str]:
        names = L.lib().wholememory_embedding_get_optimizer_state_names(self.c_embedding)   # NULL-terminated char**
        out = []
        while names and names[len(out)]:
            out.append(names[len(out)].decode())
        return out

    def dim(self):
        return self.get_embedding_tensor().dim()

    @property
    def shape(self):
        return self.get_embedding_tensor().shape

    # ---- cache -------------------------------------------------------------------------------------------------
    def set_adjust_cache(self, adjust_cache: bool):
        self.adjust_cache = bool(adjust_cache) and self.wmb_cache_policy is not None

    def cache_stats(self):
        """(hits, valid lookups, lines) of THIS rank's cache since creation / the last ``drop_all_cache``."""
        hits, looked, lines = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _call("wgamd_embedding_cache_stats", self.c_embedding, ctypes.byref(hits), ctypes.byref(looked), ctypes.byref(lines))
        return hits.value, looked.value, lines.value

    def writeback_all_cache(self):
        _call("wholememory_embedding_writeback_cache", self.c_embedding, _stream())

    def drop_all_cache(self):
        _call("wholememory_embedding_drop_all_cache", self.c_embedding, _stream())

    # ---- lookup and training -----------------------------------------------------------------------------------
    def need_grad(self) -> bool:
        return self.wm_optimizer is not None

    @property
    def need_apply(self) -> bool:
        """Is there anything for the next ``WholeMemoryOptimizer.step`` to do for this embedding?"""
        return self._touched or bool(self._pending)

    @need_apply.setter
    def need_apply(self, value: bool):
        self._touched = bool(value)

    def gather(self, indice: torch.Tensor, *, is_training: bool = False, force_dtype: Optional[torch.dtype] = None):
        assert indice.dim() == 1
        table = self.get_embedding_tensor()
        train = is_training and self.need_grad()
        out = torch.empty((indice.shape[0], table.shape[1]), device=indice.device,
                          dtype=table.dtype if force_dtype is None else force_dtype, requires_grad=train)
        self._touched = self._touched or train
        ids, rows = wrap_torch_tensor(indice), wrap_torch_tensor(out)
        _call("wholememory_embedding_gather", self.c_embedding, ids.c, rows.c, self.adjust_cache, get_wholegraph_env_fns(),
              _stream())
        return out

    def add_gradients(self, indice: torch.Tensor, grad_outputs: torch.Tensor):
        """Park the gradient rows of looked-up ids until the optimizer's next step."""
        if self._pending is None:
            self._pending = _PendingGradients(int(grad_outputs.shape[1]))
        self._pending.add(indice, grad_outputs)

    def discard_gradients(self):
        """Forget what ``add_gradients`` collected (e.g. after a step that failed validation)."""
        if self._pending is not None:
            self._pending.clear()
        self._touched = False

    def apply_gradients(self, lr: float):
        """Collective over the embedding's communicator: EVERY rank calls it, with or without gradients of its own."""
        device = torch.device("cuda", torch.cuda.current_device())
        pending = self._pending if self._pending is not None else _PendingGradients(int(self.shape[1]))
        ids, rows = pending.take(device)
        try:
            w_ids, w_rows = wrap_torch_tensor(ids), wrap_torch_tensor(rows)
            _call("wholememory_embedding_gather_gradient_apply", self.c_embedding, w_ids.c, w_rows.c, self.adjust_cache,
                  float(lr), get_wholegraph_env_fns(), _stream())
        finally:
            self.discard_gradients()

    # ---- persistence (reference embedding.py:378-407) ------------------------------------------------------------
    def _parts(self):
        """(file-name suffix, tensor) of the table and every optimizer state."""
        yield "embedding_tensor", self.get_embedding_tensor()
        for name in self.get_optimizer_state_names():
            yield name, self.get_optimizer_state(name)

    def _readwrite_cache(self) -> bool:
        return self.wmb_cache_policy is not None and self.wmb_cache_policy.access_type == "readwrite"

    def save(self, file_prefix: str):
        """(A READWRITE cache holds the newest rows and optimizer states in its dirty lines: they are written back first, or
        the files would hold rows from before the last steps.)"""
        if self._readwrite_cache():
            self.writeback_all_cache()
        for suffix, tensor in self._parts():
            tensor.to_file_prefix(f"{file_prefix}_{suffix}")

    def load(self, file_prefix: str, *, ignore_embedding: bool = False, part_count: Optional[int] = None):
        """(Resident cache lines would shadow the loaded rows, and dirty ones overwrite them at the next write-back: the cache
        is written back and emptied before the load and emptied again after it.)"""
        if self.wmb_cache_policy is not None:
            if self._readwrite_cache():
                self.writeback_all_cache()
            self.drop_all_cache()
        for suffix, tensor in self._parts():
            if suffix == "embedding_tensor" and ignore_embedding:
                continue
            tensor.from_file_prefix(f"{file_prefix}_{suffix}", part_count)
        if self.wmb_cache_policy is not None:
            self.drop_all_cache()

    def _release_views(self):
        for tensor in self._views.values():
            tensor.destroy()
        self._views = {}


def _matrix_description(rows: int, cols: int, dtype: torch.dtype) -> "L.TensorDescription":
    desc = L.TensorDescription()
    L.lib().wholememory_initialize_tensor_desc(ctypes.byref(desc))
    desc.dim = 2
    desc.sizes[0], desc.sizes[1] = int(rows), int(cols)
    desc.strides[0], desc.strides[1] = int(cols), 1
    desc.dtype = torch_dtype_to_wm(dtype)
    return desc


def create_embedding(comm: WholeMemoryCommunicator, memory_type: str, memory_location: str, dtype: torch.dtype,
                     sizes: Sequence[int], *, cache_policy: Optional[WholeMemoryCachePolicy] = None,
                     embedding_entry_partition: Optional[Sequence[int]] = None, random_init: bool = False,
                     gather_sms: int = -1, round_robin_size: int = 0) -> WholeMemoryEmbedding:
    """Reference embedding.py:410-495.  ``memory_location`` "cuda" or "cpu" (pinned host memory); ``cache_policy``:
    ``None``, a READONLY policy, or a READWRITE policy on ``comm`` itself; ``random_init``: Xavier-uniform on every rank's
    own rows; collective (ends with a barrier)."""
    if len(sizes) != 2:
        raise ValueError("an embedding is a 2-D table: sizes = [entries, dim]")
    if embedding_entry_partition is not None and round_robin_size != 0:
        print("round_robin_size is ignored because embedding_entry_partition is specified")
        round_robin_size = 0
    partition = None
    if embedding_entry_partition is not None:
        partition = (ctypes.c_size_t * len(embedding_entry_partition))(*(int(v) for v in embedding_entry_partition))
    desc = _matrix_description(sizes[0], sizes[1], dtype)
    handle = ctypes.c_void_p()
    _call("wholememory_create_embedding", ctypes.byref(handle), ctypes.byref(desc), comm.c_comm,
          memory_type_code(memory_type), memory_location_code(memory_location),
          None if cache_policy is None else cache_policy.c_policy, partition, int(gather_sms), int(round_robin_size))
    embedding = WholeMemoryEmbedding(handle, comm, cache_policy)
    if random_init:
        mine, _ = embedding.get_embedding_tensor().get_local_tensor()
        if mine.numel():
            torch.nn.init.xavier_uniform_(mine)
    comm.barrier()
    return embedding


def create_embedding_from_filelist(comm: WholeMemoryCommunicator, memory_type: str, memory_location: str,
                                   filelist: Union[List[str], str], dtype: torch.dtype, last_dim_size: int, *,
                                   cache_policy: Optional[WholeMemoryCachePolicy] = None,
                                   embedding_entry_partition: Optional[Sequence[int]] = None, gather_sms: int = -1,
                                   round_robin_size: int = 0) -> WholeMemoryEmbedding:
    """Reference embedding.py:498-564: the entry count is what the (headerless, row-major) files hold together."""
    files = [filelist] if isinstance(filelist, str) else list(filelist)
    if last_dim_size <= 0:
        raise ValueError("last_dim_size must be positive")
    row_bytes = torch.empty((), dtype=dtype).element_size() * int(last_dim_size)
    sizes = {name: os.path.getsize(name) for name in files}
    for name, size in sizes.items():
        if size % row_bytes:
            raise ValueError("File %s size is %d not mutlple of %d" % (name, size, row_bytes))
    embedding = create_embedding(comm, memory_type, memory_location, dtype, [sum(sizes.values()) // row_bytes, last_dim_size],
                                 cache_policy=cache_policy, embedding_entry_partition=embedding_entry_partition,
                                 gather_sms=gather_sms, round_robin_size=round_robin_size)
    embedding.get_embedding_tensor().from_filelist(files, round_robin_size)
    return embedding


def destroy_embedding(wm_embedding: WholeMemoryEmbedding):
    """Reference embedding.py:567-572; the tensor views handed out by the embedding die with it."""
    if wm_embedding.c_embedding is None:
        return
    wm_embedding._release_views()
    _call("wholememory_destroy_embedding", wm_embedding.c_embedding)
    wm_embedding.c_embedding = None


# ------------------------------------------------------------------------------------------------------------------
# sparse optimizer
# ------------------------------------------------------------------------------------------------------------------
class WholeMemoryOptimizer:
    """One optimizer (type + hyper-parameters) driving any number of embeddings (reference embedding.py:32-68).  Built by
    :func:`create_wholememory_optimizer`."""

    def __init__(self, global_comm: WholeMemoryCommunicator):
        self.c_opt = None
        self.global_comm = global_comm
        self.embeddings: List[WholeMemoryEmbedding] = []

    def create_optimizer(self, optimizer_type: str, param_dict: Optional[dict]):
        kind = _OPTIMIZER_CODE.get(optimizer_type.lower())
        if kind is None:
            raise ValueError(f"optimizer_type={optimizer_type}: expected one of {sorted(_OPTIMIZER_CODE)}")
        handle = ctypes.c_void_p()
        _call("wholememory_create_embedding_optimizer", ctypes.byref(handle), kind)
        self.c_opt = handle
        for name, value in (param_dict or {}).items():
            number = ctypes.c_float(float(value))
            L.check(L.lib().wholememory_optimizer_set_parameter(handle, name.encode(), ctypes.byref(number)),
                    "wholememory_optimizer_set_parameter(%s)" % name)

    def add_embedding(self, wm_embedding: WholeMemoryEmbedding):
        if not isinstance(wm_embedding, WholeMemoryEmbedding):
            raise TypeError("add_embedding takes a WholeMemoryEmbedding")
        if wm_embedding.wm_optimizer is not None:
            raise ValueError("optimizer can only be set once.")
        _call("wholememory_embedding_set_optimizer", wm_embedding.c_embedding, self.c_opt)
        wm_embedding.wm_optimizer = self
        wm_embedding.dummy_input.requires_grad_(True)
        self.embeddings.append(wm_embedding)

    def step(self, lr: float):
        """Apply what every embedding collected since the last step, then meet the other ranks (collective)."""
        for embedding in self.embeddings:
            if embedding.need_apply:
                embedding.apply_gradients(lr)
        self.global_comm.barrier()

    def zero_grad(self):
        for embedding in self.embeddings:
            embedding.discard_gradients()


def create_wholememory_optimizer(embeddings: Union[WholeMemoryEmbedding, Sequence[WholeMemoryEmbedding]],
                                 optimizer_type: str, param_dict: Optional[dict], *, global_comm=None):
    """Reference embedding.py:608-629.  ``global_comm`` (the barrier that ends a step) defaults to the first embedding's
    communicator."""
    members = [embeddings] if isinstance(embeddings, WholeMemoryEmbedding) else list(embeddings)
    optimizer = WholeMemoryOptimizer(global_comm if global_comm is not None else members[0].comm)
    optimizer.create_optimizer(optimizer_type, param_dict)
    for embedding in members:
        optimizer.add_embedding(embedding)
    return optimizer


def destroy_wholememory_optimizer(optimizer: WholeMemoryOptimizer):
    """Reference embedding.py:632-638."""
    if optimizer.c_opt is not None:
        L.lib().wholememory_destroy_embedding_optimizer(optimizer.c_opt)
        optimizer.c_opt = None


# ------------------------------------------------------------------------------------------------------------------
# autograd + nn.Module front end
# ------------------------------------------------------------------------------------------------------------------
class EmbeddingLookupFn(torch.autograd.Function):
    """Lookup whose backward parks (ids, output gradients) on the embedding for the optimizer's next step
    (reference embedding.py:220-247).  ``anchor`` is the embedding's ``dummy_input``: the only differentiable input."""

    @staticmethod
    def forward(ctx, indice, anchor, wm_embedding, is_training=False, force_dtype=None):
        rows = wm_embedding.gather(indice, is_training=is_training, force_dtype=force_dtype)
        ctx.embedding = wm_embedding if (is_training and wm_embedding.need_grad()) else None
        if ctx.embedding is not None:
            ctx.save_for_backward(indice, anchor)
        return rows

    @staticmethod
    def backward(ctx, grad_rows):
        embedding, ctx.embedding = ctx.embedding, None
        if embedding is None:
            return None, None, None, None, None
        indice, anchor = ctx.saved_tensors
        embedding.add_gradients(indice, grad_rows)
        return None, torch.zeros_like(anchor), None, None, None


class WholeMemoryEmbeddingModule(torch.nn.Module):
    """``module(indices)`` = rows of the table, differentiable w.r.t. the table in training mode
    (reference embedding.py:578-600)."""

    def __init__(self, wm_embedding: WholeMemoryEmbedding):
        super().__init__()
        self.wm_embedding = wm_embedding
        self.embedding_gather_fn = EmbeddingLookupFn.apply

    def forward(self, indice: torch.Tensor, force_dtype: Optional[torch.dtype] = None):
        e = self.wm_embedding
        return self.embedding_gather_fn(indice, e.dummy_input, e, self.training, force_dtype)
