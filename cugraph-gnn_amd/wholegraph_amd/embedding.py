"""Trainable embedding tables with sparse optimizers — the surface of
/root/reference/python/pylibwholegraph/pylibwholegraph/torch/embedding.py (``create_embedding``,
``create_embedding_from_filelist``, ``create_wholememory_optimizer``, ``WholeMemoryEmbeddingModule`` ...).

The table is a DISTRIBUTED handle of the library (rows range-partitioned over the GPUs of the communicator, in
HBM); ``gather`` is the all-to-all feature fetch, ``apply_gradients`` routes every (row, gradient) to the owner,
sums duplicates there and runs the optimizer update in one HIP kernel (csrc/wg_embedding.hip).  There is no
slower memory tier on this target, so the reference's READWRITE device cache (in front of a host table) does not exist:
asking for one raises instead of silently training without it.  A READONLY cache policy builds the reference's
"local cached global readonly embedding": hot rows owned by peer GPUs are kept in a set-associative cache in this
GPU's own HBM, and a gather returns the same bytes with or without it (include/wgamd_embedding.h).
"""
import ctypes
from typing import List, Union

import torch

from . import _lib as L
from .comm import (WholeMemoryCommunicator, get_global_communicator, get_local_device_communicator,
                   get_local_node_communicator, memory_location_code, memory_type_code)
from .env import get_stream, get_wholegraph_env_fns, torch_dtype_to_wm, wrap_torch_tensor
from .tensor import DistributedWholeMemoryTensor

_OPTIMIZER_TYPES = {"sgd": 1, "adam": 2, "lazy_adam": 2, "rmsprop": 3, "adagrad": 4}  # embedding.h:33-39; utils.py
_ACCESS_TYPES = {"readonly": 1, "readwrite": 2}


def _stream_int():
    return int(get_stream().value or 0)


class WholeMemoryOptimizer(object):
    """Sparse optimizer shared by any number of embeddings (embedding.py:32-68).  Use
    :func:`create_wholememory_optimizer`."""

    def __init__(self, global_comm: WholeMemoryCommunicator):
        self.c_opt = None
        self.embeddings = []
        self.global_comm = global_comm

    def create_optimizer(self, optimizer_type: str, param_dict: dict):
        c = ctypes.c_void_p()
        L.check(L.lib().wholememory_create_embedding_optimizer(ctypes.byref(c), _OPTIMIZER_TYPES[optimizer_type.lower()]),
                "wholememory_create_embedding_optimizer")
        self.c_opt = c
        for name, value in (param_dict or {}).items():
            v = ctypes.c_float(float(value))
            L.check(L.lib().wholememory_optimizer_set_parameter(c, name.encode(), ctypes.byref(v)),
                    "wholememory_optimizer_set_parameter(%s)" % name)

    def add_embedding(self, wm_embedding):
        assert isinstance(wm_embedding, WholeMemoryEmbedding)
        if wm_embedding.wm_optimizer is not None:
            raise ValueError("optimizer can only be set once.")
        L.check(L.lib().wholememory_embedding_set_optimizer(wm_embedding.c_embedding, self.c_opt),
                "wholememory_embedding_set_optimizer")
        wm_embedding.wm_optimizer = self
        wm_embedding.dummy_input.requires_grad_(True)
        self.embeddings.append(wm_embedding)

    def step(self, lr: float):
        """Apply the accumulated sparse gradients of every embedding (collective)."""
        for wm_embedding in self.embeddings:
            if wm_embedding.need_apply:
                wm_embedding.apply_gradients(lr)
        self.global_comm.barrier()


class WholeMemoryCachePolicy(object):
    """embedding.py:71-80.  Use :func:`create_wholememory_cache_policy` / :func:`create_builtin_cache_policy`."""

    def __init__(self, c_policy, access_type: str):
        self.c_policy = c_policy
        self.access_type = access_type


def create_wholememory_cache_policy(cache_comm, *, memory_type: str = "chunked", memory_location: str = "cuda",
                                    access_type: str = "readonly", ratio: float = 0.5):
    """embedding.py:83-110.  The policy is only recorded here; :func:`create_embedding` judges it (READONLY: a private
    per-GPU cache of ``ratio * entries`` rows; READWRITE: refused, there is no host tier to write back to)."""
    c = ctypes.c_void_p()
    L.check(L.lib().wholememory_create_embedding_cache_policy(ctypes.byref(c), cache_comm.c_comm,
                                                              memory_type_code(memory_type),
                                                              memory_location_code(memory_location),
                                                              _ACCESS_TYPES[access_type], float(ratio)),
            "wholememory_create_embedding_cache_policy")
    return WholeMemoryCachePolicy(c, access_type)


def destroy_wholememory_cache_policy(cache_policy):
    """embedding.py:113-121."""
    if cache_policy is not None and cache_policy.c_policy is not None:
        L.check(L.lib().wholememory_destroy_embedding_cache_policy(cache_policy.c_policy), "destroy_cache_policy")
        cache_policy.c_policy = None


def create_builtin_cache_policy(builtin_cache_type: str, embedding_memory_type: str, embedding_memory_location: str,
                                access_type: str, cache_ratio: float, *, cache_memory_type: str = "",
                                cache_memory_location: str = ""):
    """embedding.py:124-216: ``"none"`` -> ``None``; ``"all_devices"`` / ``"local_node"`` / ``"local_device"`` name the
    communicator the reference would spread the cache over (here every GPU keeps its own lines either way)."""
    if embedding_memory_type not in ("continuous", "chunked", "distributed", "hierarchy"):
        raise ValueError(f"embedding_memory_type={embedding_memory_type} is not valid")
    if embedding_memory_location not in ("cpu", "cuda"):
        raise ValueError(f"embedding_memory_location={embedding_memory_location} is not valid")
    if builtin_cache_type == "none":
        return None
    if cache_memory_location not in ("", "cpu", "cuda"):
        raise ValueError(f"cache_memory_location is {cache_memory_location}, should be empty or cpu, cuda")
    cache_memory_location = "cuda" if cache_memory_location == "" else cache_memory_location
    if builtin_cache_type == "all_devices":
        cache_memory_type = embedding_memory_type if cache_memory_type == "" else cache_memory_type
        comm = get_global_communicator()
    elif builtin_cache_type == "local_node":
        cache_memory_type = "chunked" if cache_memory_type == "" else cache_memory_type
        comm = get_local_node_communicator()
    elif builtin_cache_type == "local_device":
        cache_memory_type = "continuous"
        comm = get_local_device_communicator()
    else:
        raise ValueError(f"builtin_cache_type={builtin_cache_type} not supported, "
                         f"should be none, local_device, local_node or all_devices")
    return create_wholememory_cache_policy(comm, memory_type=cache_memory_type, memory_location=cache_memory_location,
                                           access_type=access_type, ratio=cache_ratio)


class EmbeddingLookupFn(torch.autograd.Function):
    """embedding.py:220-247: forward = gather; backward parks (indices, grads) on the embedding until
    ``WholeMemoryOptimizer.step``."""

    @staticmethod
    def forward(ctx, indice, dummy_input, wm_embedding, is_training=False, force_dtype=None):
        output_tensor = wm_embedding.gather(indice, is_training=is_training, force_dtype=force_dtype)
        if is_training and wm_embedding.need_grad():
            ctx.save_for_backward(indice, output_tensor, dummy_input)
            ctx.wm_embedding = wm_embedding
        return output_tensor

    @staticmethod
    def backward(ctx, grad_outputs):
        indice, output_tensor, dummy_input = ctx.saved_tensors
        wm_embedding = ctx.wm_embedding
        wm_embedding.add_gradients(indice, grad_outputs)
        ctx.wm_embedding = None
        return None, torch.zeros_like(dummy_input), None, None, None


class WholeMemoryEmbedding(object):
    """embedding.py:275-407."""

    def __init__(self, c_embedding, comm, cache_policy=None):
        self.c_embedding = c_embedding
        self.comm = comm
        self.embedding_tensor = None
        self.optimizer_states = dict()
        self.wmb_cache_policy = cache_policy
        self.adjust_cache = False
        self.wm_optimizer = None
        self.dummy_input = torch.nn.Parameter(torch.zeros(1), requires_grad=False)
        self.need_apply = False
        self.sparse_indices = []
        self.sparse_grads = []

    def dim(self):
        return self.get_embedding_tensor().dim()

    @property
    def shape(self):
        return self.get_embedding_tensor().shape

    def set_adjust_cache(self, adjust_cache: bool):
        self.adjust_cache = bool(adjust_cache) and self.wmb_cache_policy is not None

    def cache_stats(self):
        """(hits, valid lookups, lines) of THIS rank's cache since creation / the last ``drop_all_cache``."""
        h, n, lines = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        L.check(L.lib().wgamd_embedding_cache_stats(self.c_embedding, ctypes.byref(h), ctypes.byref(n), ctypes.byref(lines)),
                "wgamd_embedding_cache_stats")
        return h.value, n.value, lines.value

    def need_grad(self):
        return self.wm_optimizer is not None

    def gather(self, indice: torch.Tensor, *, is_training: bool = False, force_dtype: Union[torch.dtype, None] = None):
        assert indice.dim() == 1
        t = self.get_embedding_tensor()
        need_grad = self.need_grad() and is_training
        out = torch.empty([indice.shape[0], t.shape[1]], device=indice.device,
                          dtype=force_dtype if force_dtype is not None else t.dtype, requires_grad=need_grad)
        if need_grad:
            self.need_apply = True
        w_i, w_o = wrap_torch_tensor(indice), wrap_torch_tensor(out)
        L.check(L.lib().wholememory_embedding_gather(self.c_embedding, w_i.c, w_o.c, self.adjust_cache,
                                                     get_wholegraph_env_fns(), _stream_int()),
                "wholememory_embedding_gather")
        return out

    def add_gradients(self, indice: torch.Tensor, grad_outputs: torch.Tensor):
        self.sparse_indices.append(indice)
        self.sparse_grads.append(grad_outputs)

    def apply_gradients(self, lr: float):
        """Collective over the embedding's communicator: every rank calls it, with or without gradients of its own."""
        if self.sparse_indices:
            sparse_indices = torch.cat(self.sparse_indices)
            sparse_grads = torch.cat(self.sparse_grads).to(torch.float32).contiguous()
        else:
            dev = torch.device("cuda", torch.cuda.current_device())
            sparse_indices = torch.empty((0,), dtype=torch.int64, device=dev)
            sparse_grads = torch.empty((0, self.shape[1]), dtype=torch.float32, device=dev)
        w_i, w_g = wrap_torch_tensor(sparse_indices), wrap_torch_tensor(sparse_grads)
        L.check(L.lib().wholememory_embedding_gather_gradient_apply(self.c_embedding, w_i.c, w_g.c, self.adjust_cache,
                                                                    float(lr), get_wholegraph_env_fns(), _stream_int()),
                "wholememory_embedding_gather_gradient_apply")
        self.sparse_indices = []
        self.sparse_grads = []
        self.need_apply = False

    def writeback_all_cache(self):
        L.check(L.lib().wholememory_embedding_writeback_cache(self.c_embedding, _stream_int()), "writeback_cache")

    def drop_all_cache(self):
        L.check(L.lib().wholememory_embedding_drop_all_cache(self.c_embedding, _stream_int()), "drop_all_cache")

    def get_embedding_tensor(self):
        if self.embedding_tensor is None:
            c = L.lib().wholememory_embedding_get_embedding_tensor(self.c_embedding)
            self.embedding_tensor = DistributedWholeMemoryTensor(c, self.comm, owner=False)
        return self.embedding_tensor

    def get_optimizer_state_names(self):
        names, out, i = L.lib().wholememory_embedding_get_optimizer_state_names(self.c_embedding), [], 0
        while names and names[i]:
            out.append(names[i].decode())
            i += 1
        return out

    def get_optimizer_state(self, state_name):
        if state_name not in self.optimizer_states:
            c = L.lib().wholememory_embedding_get_optimizer_state(self.c_embedding, state_name.encode())
            if not c:
                raise KeyError(state_name)
            self.optimizer_states[state_name] = DistributedWholeMemoryTensor(c, self.comm, owner=False)
        return self.optimizer_states[state_name]

    def save(self, file_prefix: str):
        self.get_embedding_tensor().to_file_prefix(file_prefix + "_embedding_tensor")
        for state_name in self.get_optimizer_state_names():
            self.get_optimizer_state(state_name).to_file_prefix(file_prefix + "_" + state_name)

    def load(self, file_prefix: str, *, ignore_embedding: bool = False, part_count: Union[int, None] = None):
        if ignore_embedding is False:
            self.get_embedding_tensor().from_file_prefix(file_prefix + "_embedding_tensor", part_count)
        for state_name in self.get_optimizer_state_names():
            self.get_optimizer_state(state_name).from_file_prefix(file_prefix + "_" + state_name, part_count)


def create_embedding(comm: WholeMemoryCommunicator, memory_type: str, memory_location: str, dtype: torch.dtype,
                     sizes: List[int], *, cache_policy=None, embedding_entry_partition: Union[List[int], None] = None,
                     random_init: bool = False, gather_sms: int = -1, round_robin_size: int = 0):
    """embedding.py:410-495.  ``memory_location`` "cuda"; ``cache_policy``: None or a READONLY policy."""
    if cache_policy is not None and cache_policy.access_type != "readonly":
        raise NotImplementedError("only access_type='readonly' cache policies exist on this target: a readwrite device "
                                  "cache fronts a host-resident table, and every table lives in HBM here")
    assert len(sizes) == 2
    if embedding_entry_partition is not None and round_robin_size != 0:
        print("round_robin_size is ignored because embedding_entry_partition is specified")
        round_robin_size = 0
    desc = L.TensorDescription()
    L.lib().wholememory_initialize_tensor_desc(ctypes.byref(desc))
    desc.dim = 2
    desc.sizes[0], desc.sizes[1] = int(sizes[0]), int(sizes[1])
    desc.strides[0], desc.strides[1] = int(sizes[1]), 1
    desc.dtype = torch_dtype_to_wm(dtype)
    part = None
    if embedding_entry_partition is not None:
        part = (ctypes.c_size_t * len(embedding_entry_partition))(*[int(v) for v in embedding_entry_partition])
    c = ctypes.c_void_p()
    L.check(L.lib().wholememory_create_embedding(ctypes.byref(c), ctypes.byref(desc), comm.c_comm,
                                                 memory_type_code(memory_type), memory_location_code(memory_location),
                                                 cache_policy.c_policy if cache_policy is not None else None, part,
                                                 int(gather_sms), int(round_robin_size)),
            "wholememory_create_embedding")
    wm_embedding = WholeMemoryEmbedding(c, comm, cache_policy)
    wm_embedding.adjust_cache = cache_policy is not None  # embedding.py:289: adjust_cache = cache_policy is not None
    if random_init is True:
        local_tensor, _ = wm_embedding.get_embedding_tensor().get_local_tensor()
        if local_tensor.numel():
            torch.nn.init.xavier_uniform_(local_tensor)
    comm.barrier()
    return wm_embedding


def create_embedding_from_filelist(comm: WholeMemoryCommunicator, memory_type: str, memory_location: str,
                                   filelist: Union[List[str], str], dtype: torch.dtype, last_dim_size: int, *,
                                   cache_policy=None, embedding_entry_partition: Union[List[int], None] = None,
                                   gather_sms: int = -1, round_robin_size: int = 0):
    """embedding.py:498-564."""
    import os
    if isinstance(filelist, str):
        filelist = [filelist]
    assert last_dim_size > 0
    file_entry_size = torch.tensor([], dtype=dtype).element_size() * last_dim_size
    total_file_size = 0
    for filename in filelist:
        file_size = os.path.getsize(filename)
        if file_size % file_entry_size != 0:
            raise ValueError("File %s size is %d not mutlple of %d" % (filename, file_size, file_entry_size))
        total_file_size += file_size
    wm_embedding = create_embedding(comm, memory_type, memory_location, dtype,
                                    [total_file_size // file_entry_size, last_dim_size], cache_policy=cache_policy,
                                    embedding_entry_partition=embedding_entry_partition, gather_sms=gather_sms,
                                    round_robin_size=round_robin_size)
    wm_embedding.get_embedding_tensor().from_filelist(filelist, round_robin_size)
    return wm_embedding


def destroy_embedding(wm_embedding: WholeMemoryEmbedding):
    """embedding.py:567-572 (the state / embedding tensor wrappers die with it)."""
    if wm_embedding.c_embedding is not None:
        for t in [wm_embedding.embedding_tensor] + list(wm_embedding.optimizer_states.values()):
            if t is not None:
                t.destroy()
        wm_embedding.embedding_tensor, wm_embedding.optimizer_states = None, dict()
        L.check(L.lib().wholememory_destroy_embedding(wm_embedding.c_embedding), "wholememory_destroy_embedding")
        wm_embedding.c_embedding = None


class WholeMemoryEmbeddingModule(torch.nn.Module):
    """torch.nn.Module wrapper (embedding.py:578-600)."""

    def __init__(self, wm_embedding: WholeMemoryEmbedding):
        super().__init__()
        self.wm_embedding = wm_embedding
        self.embedding_gather_fn = EmbeddingLookupFn.apply

    def forward(self, indice: torch.Tensor, force_dtype: Union[torch.dtype, None] = None):
        return self.embedding_gather_fn(indice, self.wm_embedding.dummy_input, self.wm_embedding, self.training,
                                        force_dtype)


def create_wholememory_optimizer(embeddings: Union[WholeMemoryEmbedding, List[WholeMemoryEmbedding]],
                                 optimizer_type: str, param_dict: dict, *, global_comm=None):
    """embedding.py:608-629.  ``global_comm`` (the barrier after a step) defaults to the first embedding's
    communicator."""
    first = embeddings if isinstance(embeddings, WholeMemoryEmbedding) else embeddings[0]
    wm_optimizer = WholeMemoryOptimizer(global_comm if global_comm is not None else first.comm)
    wm_optimizer.create_optimizer(optimizer_type, param_dict)
    for em in ([embeddings] if isinstance(embeddings, WholeMemoryEmbedding) else embeddings):
        wm_optimizer.add_embedding(em)
    return wm_optimizer


def destroy_wholememory_optimizer(optimizer: WholeMemoryOptimizer):
    """embedding.py:632-638."""
    if optimizer.c_opt is not None:
        L.lib().wholememory_destroy_embedding_optimizer(optimizer.c_opt)
        optimizer.c_opt = None
