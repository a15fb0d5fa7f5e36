"""``GraphStructure`` — one relation in CSR form + the multi-hop fan-out walk.

Same interface as ``pylibwholegraph.torch.graph_structure.GraphStructure``
(/root/reference/python/pylibwholegraph/pylibwholegraph/torch/graph_structure.py:13-196):
``set_csr_graph``, ``set_node_attribute``/``set_edge_attribute``, the two one-hop samplers and
``multilayer_sample_without_replacement`` with the identical return tuple
``(target_gids, edge_indice, csr_row_ptr, csr_col_ind)``.

Addition (no reference counterpart): ``multilayer_sample_nosync`` runs the same walk with the
no-host-sync hop of ``include/wgamd_ext.h`` — identical results, sizes stay on the device.
"""
import random
from typing import List, Union

import torch

from . import graph_ops, wholegraph_ops
from .fused import NoSyncWalk
from .tensor import WholeMemoryTensor


def _unwrap(t):
    return t.local_tensor if isinstance(t, WholeMemoryTensor) else t


class GraphStructure(object):
    r"""Graph structure storage: the CSR of one relation plus node / edge attributes."""

    def __init__(self):
        super().__init__()
        self.node_count = 0
        self.edge_count = 0
        self.csr_row_ptr = None
        self.csr_col_ind = None
        self.node_attributes = {}
        self.edge_attributes = {}
        self._walks = {}

    def set_csr_graph(self, csr_row_ptr, csr_col_ind):
        """Set the CSR graph structure (row pointer int64, column index int32|int64)."""
        assert csr_row_ptr.dim() == 1
        assert csr_row_ptr.dtype == torch.int64
        assert csr_row_ptr.shape[0] > 1
        self.node_count = csr_row_ptr.shape[0] - 1
        self.edge_count = csr_col_ind.shape[0]
        assert csr_col_ind.dim() == 1
        assert csr_col_ind.dtype == torch.int32 or csr_col_ind.dtype == torch.int64
        self.csr_row_ptr = csr_row_ptr
        self.csr_col_ind = csr_col_ind
        self._walks = {}

    def set_node_attribute(self, attr_name: str, attr_tensor):
        assert attr_name not in self.node_attributes
        assert attr_tensor.shape[0] == self.node_count
        self.node_attributes[attr_name] = attr_tensor

    def set_edge_attribute(self, attr_name: str, attr_tensor):
        assert attr_name not in self.edge_attributes
        assert attr_tensor.shape[0] == self.edge_count
        self.edge_attributes[attr_name] = attr_tensor

    def unweighted_sample_without_replacement_one_hop(
        self,
        center_nodes_tensor: torch.Tensor,
        max_sample_count: int,
        *,
        random_seed: Union[int, None] = None,
        need_center_local_output: bool = False,
        need_edge_output: bool = False,
    ):
        """-> csr_row_ptr, sampled_nodes[, center_node_local_id, edge_index]"""
        return wholegraph_ops.unweighted_sample_without_replacement(
            self.csr_row_ptr, self.csr_col_ind, center_nodes_tensor, max_sample_count, random_seed,
            need_center_local_output, need_edge_output)

    def weighted_sample_without_replacement_one_hop(
        self,
        weight_name: str,
        center_nodes_tensor: torch.Tensor,
        max_sample_count: int,
        *,
        random_seed: Union[int, None] = None,
        need_center_local_output: bool = False,
        need_edge_output: bool = False,
    ):
        assert weight_name in self.edge_attributes
        return wholegraph_ops.weighted_sample_without_replacement(
            self.csr_row_ptr, self.csr_col_ind, self.edge_attributes[weight_name], center_nodes_tensor,
            max_sample_count, random_seed, need_center_local_output, need_edge_output)

    def multilayer_sample_without_replacement(
        self,
        node_ids: torch.Tensor,
        max_neighbors: List[int],
        weight_name: Union[str, None] = None,
        *,
        random_seeds: Union[List[int], None] = None,
    ):
        """Multilayer sample without replacement (graph_structure.py:136-196).

        ``random_seeds`` (extension) pins the per-hop seeds, in execution order (seed hop first);
        the reference draws ``random.getrandbits(64)`` per hop and so is not reproducible.
        :return: target_gids, edge_indice, csr_row_ptr, csr_col_ind
        """
        hops = len(max_neighbors)
        edge_indice = [None] * hops
        csr_row_ptr = [None] * hops
        csr_col_ind = [None] * hops
        target_gids = [None] * (hops + 1)
        target_gids[hops] = node_ids
        for i in range(hops - 1, -1, -1):
            seed = None if random_seeds is None else random_seeds[hops - i - 1]
            if weight_name is None:
                offsets, nbr_gids, src_lids = self.unweighted_sample_without_replacement_one_hop(
                    target_gids[i + 1], max_neighbors[hops - i - 1], random_seed=seed,
                    need_center_local_output=True)
            else:
                offsets, nbr_gids, src_lids = self.weighted_sample_without_replacement_one_hop(
                    weight_name, target_gids[i + 1], max_neighbors[hops - i - 1], random_seed=seed,
                    need_center_local_output=True)
            if nbr_gids.dtype != target_gids[i + 1].dtype:
                nbr_gids = nbr_gids.to(target_gids[i + 1].dtype)
            unique_gids, raw_to_unique = graph_ops.append_unique(target_gids[i + 1], nbr_gids,
                                                                 need_neighbor_raw_to_unique=True)
            csr_row_ptr[i] = offsets
            csr_col_ind[i] = raw_to_unique
            n = nbr_gids.size()[0]
            edge_indice[i] = torch.cat([torch.reshape(raw_to_unique, (1, n)), torch.reshape(src_lids, (1, n))])
            target_gids[i] = unique_gids
        return target_gids, edge_indice, csr_row_ptr, csr_col_ind

    def multilayer_sample_nosync(self, node_ids: torch.Tensor, max_neighbors: List[int],
                                 random_seeds: Union[List[int], None] = None):
        """Same walk, no host synchronisation: returns a ``fused.WalkResult`` whose tensors are
        capacity-sized with device-resident counts (``.finalize()`` trims them to the exact
        tuple of ``multilayer_sample_without_replacement``)."""
        key = (int(node_ids.shape[0]), tuple(max_neighbors), node_ids.dtype)
        walk = self._walks.get(key)
        if walk is None:
            walk = NoSyncWalk(_unwrap(self.csr_row_ptr), _unwrap(self.csr_col_ind), key[0], list(max_neighbors),
                              node_ids.dtype)
            self._walks[key] = walk
        if random_seeds is None:
            random_seeds = [random.getrandbits(64) for _ in max_neighbors]
        return walk.run(node_ids, random_seeds)
