"""Neighbour-sampling ops — same names, argument order and return tuples as
``pylibwholegraph.torch.wholegraph_ops``
(/root/reference/python/pylibwholegraph/pylibwholegraph/torch/wholegraph_ops.py:18-175).

The CSR arguments are device ``torch.Tensor``s (or ``WholeMemoryTensor``s wrapping one): on
MI355X the CSR is replicated per GPU (288 GB HBM holds every BASELINE graph; DESIGN.md
§multi-GPU), so there is no mapped/NCCL CSR variant to select.
"""
import random
from typing import Union

import torch

from . import _lib as L
from .env import TorchMemoryContext, get_stream, get_wholegraph_env_fns, wrap_torch_tensor


def _as_device_tensor(t):
    return t.local_tensor if hasattr(t, "local_tensor") else t


def _is_partitioned(t):
    """A table spread over the GPUs of a communicator (C-level handle), as opposed to a tensor this GPU holds whole."""
    return hasattr(t, "c") and getattr(t, "is_distributed", False) is True   # (torch.Tensor.is_distributed is a method)


class _Handle:
    """Stands where ``wrap_torch_tensor`` would: the C tensor of a partitioned table is passed as it is."""

    def __init__(self, t):
        self.c = t.c


def _sample(weighted, row_ptr, col, weight, center_nodes_tensor, max_sample_count, random_seed,
            need_center_local_output, need_edge_output, with_replacement=False):
    partitioned = _is_partitioned(row_ptr) or _is_partitioned(col)
    if partitioned:
        # the CSR itself is partitioned (wholegraph_ops.py:18-83 takes WholeMemory tensors of any memory type): the op
        # fetches row offsets and columns from their owners; collective over the tensors' communicator
        assert not weighted and not with_replacement, "a partitioned CSR is served by the unweighted op only"
    else:
        row_ptr, col = _as_device_tensor(row_ptr), _as_device_tensor(col)
    assert row_ptr.dim() == 1
    assert col.dim() == 1
    assert center_nodes_tensor.dim() == 1
    if weighted:
        weight = _as_device_tensor(weight)
        assert weight.dim() == 1
        assert weight.shape[0] == col.shape[0]
    if random_seed is None:
        random_seed = random.getrandbits(64)
    output_sample_offset_tensor = torch.empty(center_nodes_tensor.shape[0] + 1, device="cuda", dtype=torch.int)
    dest_ctx = TorchMemoryContext()
    lid_ctx = TorchMemoryContext() if need_center_local_output else None
    gid_ctx = TorchMemoryContext() if need_edge_output else None
    w_row = _Handle(row_ptr) if _is_partitioned(row_ptr) else wrap_torch_tensor(_as_device_tensor(row_ptr))
    w_col = _Handle(col) if _is_partitioned(col) else wrap_torch_tensor(_as_device_tensor(col))
    w_seeds, w_off = wrap_torch_tensor(center_nodes_tensor), wrap_torch_tensor(output_sample_offset_tensor)
    lid_c = lid_ctx.get_c_context() if lid_ctx else None
    gid_c = gid_ctx.get_c_context() if gid_ctx else None
    seed = random_seed & 0xFFFFFFFFFFFFFFFF
    if with_replacement:
        assert not weighted, "sampling with replacement is uniform"
        L.check(L.lib().wgamd_csr_uniform_sample_with_replacement(
            w_row.c, w_col.c, w_seeds.c, int(max_sample_count), w_off.c, dest_ctx.get_c_context(),
            lid_c, gid_c, seed, get_wholegraph_env_fns(), get_stream()),
            "wgamd_csr_uniform_sample_with_replacement")
    elif weighted:
        w_weight = wrap_torch_tensor(weight)
        L.check(L.lib().wholegraph_csr_weighted_sample_without_replacement(
            w_row.c, w_col.c, w_weight.c, w_seeds.c, int(max_sample_count), w_off.c, dest_ctx.get_c_context(),
            lid_c, gid_c, seed, get_wholegraph_env_fns(), get_stream()),
            "wholegraph_csr_weighted_sample_without_replacement")
    else:
        L.check(L.lib().wholegraph_csr_unweighted_sample_without_replacement(
            w_row.c, w_col.c, w_seeds.c, int(max_sample_count), w_off.c, dest_ctx.get_c_context(),
            lid_c, gid_c, seed, get_wholegraph_env_fns(), get_stream()),
            "wholegraph_csr_unweighted_sample_without_replacement")
    out = [output_sample_offset_tensor, dest_ctx.get_tensor()]
    if need_center_local_output:
        out.append(lid_ctx.get_tensor())
    if need_edge_output:
        out.append(gid_ctx.get_tensor())
    return tuple(out)


def unweighted_sample_without_replacement(
    wm_csr_row_ptr_tensor,
    wm_csr_col_ptr_tensor,
    center_nodes_tensor: "torch.Tensor",
    max_sample_count: int,
    random_seed: Union[int, None] = None,
    need_center_local_output: bool = False,
    need_edge_output: bool = False,
):
    """Unweighted neighborhood sample in CSR WholeGraph (wholegraph_ops.py:18-83).

    Returns ``(sample_offset, dest[, center_local_id][, edge_gid])``."""
    return _sample(False, wm_csr_row_ptr_tensor, wm_csr_col_ptr_tensor, None, center_nodes_tensor,
                   max_sample_count, random_seed, need_center_local_output, need_edge_output)


def unweighted_sample_with_replacement(wm_csr_row_ptr_tensor, wm_csr_col_ptr_tensor, center_nodes_tensor: "torch.Tensor",
                                       sample_count: int, random_seed: Union[int, None] = None,
                                       need_center_local_output: bool = False, need_edge_output: bool = False):
    """Uniform sampling WITH replacement (cugraph_pyg ``replace=True``; ``wgamd_csr_uniform_sample_with_replacement``):
    every seed with neighbours yields exactly ``sample_count`` picks.  Same return tuple as the op above."""
    return _sample(False, wm_csr_row_ptr_tensor, wm_csr_col_ptr_tensor, None, center_nodes_tensor, sample_count,
                   random_seed, need_center_local_output, need_edge_output, with_replacement=True)


def weighted_sample_without_replacement(
    wm_csr_row_ptr_tensor,
    wm_csr_col_ptr_tensor,
    wm_csr_weight_ptr_tensor,
    center_nodes_tensor: "torch.Tensor",
    max_sample_count: int,
    random_seed: Union[int, None] = None,
    need_center_local_output: bool = False,
    need_edge_output: bool = False,
):
    """Weighted neighborhood sample in CSR WholeGraph (wholegraph_ops.py:86-155)."""
    return _sample(True, wm_csr_row_ptr_tensor, wm_csr_col_ptr_tensor, wm_csr_weight_ptr_tensor,
                   center_nodes_tensor, max_sample_count, random_seed, need_center_local_output,
                   need_edge_output)


def generate_random_positive_int_cpu(random_seed, sub_sequence, output_random_value_count):
    """Host accessor of the op RNG stream (wholegraph_ops.py:158-165)."""
    output = torch.empty((output_random_value_count,), dtype=torch.int)
    w = wrap_torch_tensor(output)
    L.check(L.lib().generate_random_positive_int_cpu(int(random_seed), int(sub_sequence), w.c),
            "generate_random_positive_int_cpu")
    return output


def generate_exponential_distribution_negative_float_cpu(random_seed: int, sub_sequence: int,
                                                         output_random_value_count: int):
    """(wholegraph_ops.py:168-175)"""
    output = torch.empty((output_random_value_count,), dtype=torch.float)
    w = wrap_torch_tensor(output)
    L.check(L.lib().generate_exponential_distribution_negative_float_cpu(int(random_seed), int(sub_sequence), w.c),
            "generate_exponential_distribution_negative_float_cpu")
    return output
