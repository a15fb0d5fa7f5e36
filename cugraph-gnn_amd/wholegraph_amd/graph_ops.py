"""Sampled-subgraph ops with the names and behaviour of ``pylibwholegraph.torch.graph_ops``
(/root/reference/python/pylibwholegraph/pylibwholegraph/torch/graph_ops.py:15-95), on the HIP kernels
of ``libwholegraph_amd`` (``graph_append_unique`` / ``csr_add_self_loop``, include/wgamd_ops.h)."""
import torch

from . import _lib as L
from .env import TorchMemoryContext, get_stream, get_wholegraph_env_fns, wrap_torch_tensor


def _require_device_vectors(**tensors):
    for name, t in tensors.items():
        assert t.dim() == 1, f"{name} must be 1-D"
        assert t.is_cuda, f"{name} must live on the GPU"


def append_unique(target_node_tensor: torch.Tensor, neighbor_node_tensor: torch.Tensor,
                  need_neighbor_raw_to_unique: bool = False):
    """Renumbering step of a hop: ``unique = targets ++ (neighbours not among the targets)``.

    The targets keep their positions (ids ``0..T-1``); every other neighbour id appears once, in order
    of first appearance (the reference leaves that order unspecified).  With
    ``need_neighbor_raw_to_unique`` the int32 position of every neighbour inside ``unique`` is returned
    too.  Example (the reference's, graph_ops.py:21-29): targets ``[3, 11, 2, 10]``, neighbours
    ``[4, 5, 2, 11, 6, 9, 10, 5]`` give ``unique = [3, 11, 2, 10, 4, 5, 6, 9]`` and
    ``mapping = [4, 5, 2, 1, 6, 7, 3, 5]``.
    """
    _require_device_vectors(target_node_tensor=target_node_tensor, neighbor_node_tensor=neighbor_node_tensor)
    mapping = (torch.empty(neighbor_node_tensor.shape[0], dtype=torch.int32, device=neighbor_node_tensor.device)
               if need_neighbor_raw_to_unique else None)
    unique_ctx = TorchMemoryContext()           # the op allocates `unique` through the env callbacks
    handles = [wrap_torch_tensor(t) for t in (target_node_tensor, neighbor_node_tensor, mapping)]
    rc = L.lib().graph_append_unique(handles[0].c, handles[1].c, unique_ctx.get_c_context(), handles[2].c,
                                     get_wholegraph_env_fns(), get_stream())
    L.check(rc, "graph_append_unique")
    unique = unique_ctx.get_tensor()
    return (unique, mapping) if need_neighbor_raw_to_unique else unique


def add_csr_self_loop(csr_row_ptr_tensor: torch.Tensor, csr_col_ptr_tensor: torch.Tensor):
    """Row ``i`` of a sampled int32 CSR becomes ``[i] ++ row i`` (GAT-style self loops; graph_ops.py:63-95).
    Existing self loops are not detected.  Returns the new ``(row_ptr, col)``."""
    _require_device_vectors(csr_row_ptr_tensor=csr_row_ptr_tensor, csr_col_ptr_tensor=csr_col_ptr_tensor)
    n_rows = csr_row_ptr_tensor.shape[0] - 1
    new_row_ptr = torch.empty_like(csr_row_ptr_tensor)
    new_col = csr_col_ptr_tensor.new_empty(csr_col_ptr_tensor.shape[0] + n_rows)
    handles = [wrap_torch_tensor(t) for t in (csr_row_ptr_tensor, csr_col_ptr_tensor, new_row_ptr, new_col)]
    L.check(L.lib().csr_add_self_loop(*(h.c for h in handles), get_stream()), "csr_add_self_loop")
    return new_row_ptr, new_col
