"""``GraphStructure`` — one relation held as a device CSR, plus the multi-hop fan-out walk over it.

Interface parity with ``pylibwholegraph.torch.graph_structure.GraphStructure``
(/root/reference/python/pylibwholegraph/pylibwholegraph/torch/graph_structure.py:13-196): the same
public attributes (``node_count``, ``edge_count``, ``csr_row_ptr``, ``csr_col_ind``,
``node_attributes``, ``edge_attributes``), the same method names and keyword arguments, and
``multilayer_sample_without_replacement`` returns the identical 4-tuple
``(target_gids, edge_indice, csr_row_ptr, csr_col_ind)`` with level ``i`` produced from level
``i + 1`` exactly as the reference does (:156-195).

Extensions with no reference counterpart: ``random_seeds=`` pins the per-hop seeds (the reference
draws ``random.getrandbits(64)`` per hop, so its walks are not reproducible), and
``multilayer_sample_nosync`` runs the same walk through the no-host-sync hop of
``include/wgamd_ext.h`` (sizes stay on the device; identical results).
"""
import random
from typing import List, Optional, Sequence, Union

import torch

from . import graph_ops, wholegraph_ops
from .fused import CapturedWalk, NoSyncWalk
from .tensor import WholeMemoryTensor


def _device_tensor(t):
    return t.local_tensor if isinstance(t, WholeMemoryTensor) else t


class GraphStructure(object):
    def __init__(self):
        super().__init__()
        self.node_count, self.edge_count = 0, 0
        self.csr_row_ptr = self.csr_col_ind = None
        self.node_attributes, self.edge_attributes = {}, {}
        self._walk_cache = {}
        self._captured_ok = True

    # ---- graph + attributes ----------------------------------------------------------------
    def set_csr_graph(self, csr_row_ptr, csr_col_ind):
        """Install the CSR: ``csr_row_ptr`` int64 [V+1], ``csr_col_ind`` int32|int64 [E] (device)."""
        for name, t in (("csr_row_ptr", csr_row_ptr), ("csr_col_ind", csr_col_ind)):
            assert t.dim() == 1, f"{name} must be 1-D"
        assert csr_row_ptr.dtype == torch.int64 and csr_row_ptr.shape[0] > 1
        assert csr_col_ind.dtype in (torch.int32, torch.int64)
        self.csr_row_ptr, self.csr_col_ind = csr_row_ptr, csr_col_ind
        self.node_count, self.edge_count = csr_row_ptr.shape[0] - 1, csr_col_ind.shape[0]
        self._walk_cache.clear()

    def _set_attribute(self, table, count, kind, attr_name, attr_tensor):
        assert attr_name not in table, f"{kind} attribute {attr_name!r} already set"
        assert attr_tensor.shape[0] == count, f"{kind} attribute must have {count} rows"
        table[attr_name] = attr_tensor

    def set_node_attribute(self, attr_name: str, attr_tensor):
        self._set_attribute(self.node_attributes, self.node_count, "node", attr_name, attr_tensor)

    def set_edge_attribute(self, attr_name: str, attr_tensor):
        self._set_attribute(self.edge_attributes, self.edge_count, "edge", attr_name, attr_tensor)

    # ---- one hop ---------------------------------------------------------------------------
    def _one_hop(self, weight_name: Optional[str], centers, fanout, seed, want_lid, want_eid):
        if weight_name is None:
            return wholegraph_ops.unweighted_sample_without_replacement(
                self.csr_row_ptr, self.csr_col_ind, centers, fanout, seed, want_lid, want_eid)
        assert weight_name in self.edge_attributes, f"no edge attribute {weight_name!r}"
        return wholegraph_ops.weighted_sample_without_replacement(
            self.csr_row_ptr, self.csr_col_ind, self.edge_attributes[weight_name], centers, fanout, seed,
            want_lid, want_eid)

    def unweighted_sample_without_replacement_one_hop(self, center_nodes_tensor: torch.Tensor,
                                                      max_sample_count: int, *,
                                                      random_seed: Union[int, None] = None,
                                                      need_center_local_output: bool = False,
                                                      need_edge_output: bool = False):
        """-> (sample_offset, sampled_nodes[, center_local_id][, edge_gid]); uniform, without replacement."""
        return self._one_hop(None, center_nodes_tensor, max_sample_count, random_seed, need_center_local_output,
                             need_edge_output)

    def weighted_sample_without_replacement_one_hop(self, weight_name: str, center_nodes_tensor: torch.Tensor,
                                                    max_sample_count: int, *,
                                                    random_seed: Union[int, None] = None,
                                                    need_center_local_output: bool = False,
                                                    need_edge_output: bool = False):
        """Same, biased by the edge attribute ``weight_name`` (A-Res)."""
        return self._one_hop(weight_name, center_nodes_tensor, max_sample_count, random_seed,
                             need_center_local_output, need_edge_output)

    # ---- multi-hop walk --------------------------------------------------------------------
    def multilayer_sample_without_replacement(self, node_ids: torch.Tensor, max_neighbors: List[int],
                                              weight_name: Union[str, None] = None, *,
                                              random_seeds: Optional[Sequence[int]] = None):
        """Fan-out walk + renumbering.  The k-th executed hop (k = 0 is the seed hop, fan-out
        ``max_neighbors[k]``) fills level ``i = hops - 1 - k``:

        * ``csr_row_ptr[i]`` / ``csr_col_ind[i]``: CSR of the hop — rows = ``target_gids[i+1]``, columns index
          ``target_gids[i]``;
        * ``edge_indice[i]`` = ``[csr_col_ind[i]; center_local_id]`` (2 x E_hop);
        * ``target_gids[i]`` = ``target_gids[i+1]`` followed by the newly discovered vertices.

        :return: target_gids, edge_indice, csr_row_ptr, csr_col_ind
        """
        hops = len(max_neighbors)
        if (weight_name is None and hops > 0 and all(int(m) > 0 for m in max_neighbors) and node_ids.is_cuda
                and node_ids.shape[0] > 0 and self._captured_ok
                and not any(wholegraph_ops._is_partitioned(t) for t in (self.csr_row_ptr, self.csr_col_ind))
                and node_ids.dtype == _device_tensor(self.csr_col_ind).dtype):
            # the walk's ~11 launches per hop replayed from ONE captured HIP graph, one host read-back for all sizes: the
            # same kernels, the same results as the op-by-op loop below (tests/test_gpu_renumber_gather.py), a third of its
            # host time.  Anything the capture cannot serve (biased hops, fan-out -1, a partitioned CSR) takes the loop.
            key = ("captured", int(node_ids.shape[0]), tuple(int(m) for m in max_neighbors), node_ids.dtype)
            try:
                if key not in self._walk_cache:
                    self._walk_cache[key] = CapturedWalk(NoSyncWalk(
                        _device_tensor(self.csr_row_ptr), _device_tensor(self.csr_col_ind), key[1], list(key[2]), node_ids.dtype))
                seeds = random_seeds if random_seeds is not None else [random.getrandbits(64) for _ in max_neighbors]
                res = self._walk_cache[key].run(node_ids.contiguous(), list(seeds))
                tg, ei, rp, ci = res.finalize_single(copy=True)
                tg[hops] = node_ids
                return tg, ei, rp, ci
            except RuntimeError as exc:      # graph capture refused by the runtime: remember, and take the op-by-op loop
                import warnings
                warnings.warn("captured walk unavailable (%s): using the op-by-op walk" % (str(exc).splitlines()[0][:200],))
                self._captured_ok = False
                self._walk_cache.pop(key, None)
        levels = {name: [None] * hops for name in ("edge", "rowptr", "col")}
        target_gids = [None] * hops + [node_ids]
        for k, fanout in enumerate(max_neighbors):
            i = hops - 1 - k
            centers = target_gids[i + 1]
            offsets, nbr_gids, center_lid = self._one_hop(weight_name, centers, fanout,
                                                          None if random_seeds is None else random_seeds[k], True, False)
            if nbr_gids.dtype != centers.dtype:          # int32 CSR columns under int64 seeds
                nbr_gids = nbr_gids.to(centers.dtype)
            target_gids[i], mapping = graph_ops.append_unique(centers, nbr_gids, need_neighbor_raw_to_unique=True)
            levels["rowptr"][i], levels["col"][i] = offsets, mapping
            levels["edge"][i] = torch.stack([mapping, center_lid])
        return target_gids, levels["edge"], levels["rowptr"], levels["col"]

    def multilayer_sample_nosync(self, node_ids: torch.Tensor, max_neighbors: List[int],
                                 random_seeds: Optional[Sequence[int]] = None):
        """The same walk without host synchronisation: returns a ``fused.WalkResult`` (capacity-sized device
        tensors + device-resident counts); ``.finalize()`` trims it to the tuple above."""
        key = (int(node_ids.shape[0]), tuple(max_neighbors), node_ids.dtype)
        if any(wholegraph_ops._is_partitioned(t) for t in (self.csr_row_ptr, self.csr_col_ind)):
            raise NotImplementedError("the no-sync walk reads the CSR with plain loads: it needs a CSR this GPU holds whole "
                                      "(multilayer_sample_without_replacement serves a partitioned one)")
        if key not in self._walk_cache:
            self._walk_cache[key] = NoSyncWalk(_device_tensor(self.csr_row_ptr), _device_tensor(self.csr_col_ind),
                                               key[0], list(max_neighbors), node_ids.dtype)
        seeds = random_seeds if random_seeds is not None else [random.getrandbits(64) for _ in max_neighbors]
        return self._walk_cache[key].run(node_ids, list(seeds))
