"""Sampled-subgraph ops — same names/behaviour as ``pylibwholegraph.torch.graph_ops``
(/root/reference/python/pylibwholegraph/pylibwholegraph/torch/graph_ops.py:15-95)."""
import torch

from . import _lib as L
from .env import TorchMemoryContext, get_stream, get_wholegraph_env_fns, wrap_torch_tensor


def append_unique(target_node_tensor: "torch.Tensor", neighbor_node_tensor: "torch.Tensor",
                  need_neighbor_raw_to_unique: bool = False):
    """Append neighbor_node_tensor to target_node_tensor, keep target_node_tensor unchanged and
    do unique (graph_ops.py:15-60).  e.g. targets [3, 11, 2, 10], neighbours
    [4, 5, 2, 11, 6, 9, 10, 5] -> unique [3, 11, 2, 10, 4, 5, 6, 9] (new nodes in first-appearance
    order — the reference leaves that order unspecified) and mapping [4, 5, 2, 1, 6, 7, 3, 5]."""
    assert target_node_tensor.dim() == 1
    assert neighbor_node_tensor.dim() == 1
    assert target_node_tensor.is_cuda
    assert neighbor_node_tensor.is_cuda
    unique_ctx = TorchMemoryContext()
    mapping = None
    if need_neighbor_raw_to_unique:
        mapping = torch.empty(neighbor_node_tensor.shape[0], device="cuda", dtype=torch.int)
    w_t, w_n, w_m = (wrap_torch_tensor(target_node_tensor), wrap_torch_tensor(neighbor_node_tensor),
                     wrap_torch_tensor(mapping))
    L.check(L.lib().graph_append_unique(w_t.c, w_n.c, unique_ctx.get_c_context(), w_m.c,
                                        get_wholegraph_env_fns(), get_stream()), "graph_append_unique")
    if need_neighbor_raw_to_unique:
        return unique_ctx.get_tensor(), mapping
    return unique_ctx.get_tensor()


def add_csr_self_loop(csr_row_ptr_tensor: "torch.Tensor", csr_col_ptr_tensor: "torch.Tensor"):
    """Add self loop to sampled CSR graph (graph_ops.py:63-95).  Does not check whether the raw
    CSR already holds self loops."""
    assert csr_row_ptr_tensor.dim() == 1
    assert csr_col_ptr_tensor.dim() == 1
    assert csr_row_ptr_tensor.is_cuda
    assert csr_col_ptr_tensor.is_cuda
    out_row = torch.empty((csr_row_ptr_tensor.shape[0],), device="cuda", dtype=csr_row_ptr_tensor.dtype)
    out_col = torch.empty((csr_col_ptr_tensor.shape[0] + csr_row_ptr_tensor.shape[0] - 1,), device="cuda",
                          dtype=csr_col_ptr_tensor.dtype)
    ws = [wrap_torch_tensor(t) for t in (csr_row_ptr_tensor, csr_col_ptr_tensor, out_row, out_col)]
    L.check(L.lib().csr_add_self_loop(ws[0].c, ws[1].c, ws[2].c, ws[3].c, get_stream()), "csr_add_self_loop")
    return out_row, out_col
