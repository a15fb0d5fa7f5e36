"""No-host-sync multi-hop walk (``wgamd_sample_hop_nosync``, include/wgamd_ext.h).

Pre-allocates capacity-sized buffers once per (batch size, fan-out) and replays the same launch
sequence for every mini-batch; sizes live in a small device tensor.  Results are identical to
``GraphStructure.multilayer_sample_without_replacement`` with the same seeds (tests/test_gpu_walk.py).
"""
from dataclasses import dataclass
from typing import List

import torch

from . import _lib as L
from .env import get_stream, torch_dtype_to_wm


@dataclass
class WalkResult:
    """Capacity-sized outputs of one walk.  ``counts[k] = [n_edges_k, n_unique_k]`` for the k-th
    executed hop (seed hop first); everything stays on the device until ``finalize``."""

    hops: int
    seeds: torch.Tensor
    unique: List[torch.Tensor]        # per executed hop: targets ++ new nodes (capacity-sized)
    offsets: List[torch.Tensor]       # int32 [target_cap+1]
    neighbor_lid: List[torch.Tensor]  # int32 [edge_cap]
    center_lid: List[torch.Tensor]    # int32 [edge_cap]
    counts: torch.Tensor              # int32 [hops, 2]
    target_caps: List[int]

    def finalize(self):
        """One D2H of the counts, then trim to the reference tuple
        ``(target_gids, edge_indice, csr_row_ptr, csr_col_ind)`` (graph_structure.py:186-196)."""
        c = self.counts.cpu().tolist()
        hops = self.hops
        target_gids = [None] * (hops + 1)
        edge_indice, csr_row_ptr, csr_col_ind = [None] * hops, [None] * hops, [None] * hops
        target_gids[hops] = self.seeds
        n_targets = self.seeds.shape[0]
        for k in range(hops):
            i = hops - 1 - k
            n_edges, n_unique = c[k]
            csr_row_ptr[i] = self.offsets[k][: n_targets + 1]
            csr_col_ind[i] = self.neighbor_lid[k][:n_edges]
            edge_indice[i] = torch.stack([csr_col_ind[i], self.center_lid[k][:n_edges]])
            target_gids[i] = self.unique[k][:n_unique]
            n_targets = n_unique
        return target_gids, edge_indice, csr_row_ptr, csr_col_ind


class NoSyncWalk:
    def __init__(self, csr_row_ptr: torch.Tensor, csr_col_ind: torch.Tensor, batch_size: int,
                 max_neighbors: List[int], id_dtype=torch.int64):
        assert csr_row_ptr.is_cuda and csr_col_ind.is_cuda
        assert all(m > 0 for m in max_neighbors), "the no-sync walk needs positive fan-outs"
        assert csr_col_ind.dtype == id_dtype, "no-sync walk: seeds and csr_col must share a dtype"
        self.row_ptr, self.col = csr_row_ptr, csr_col_ind
        self.fanouts = list(max_neighbors)
        self.id_dtype = id_dtype
        self.wm_dtype = torch_dtype_to_wm(id_dtype)
        dev = csr_row_ptr.device
        self.target_caps, self.edge_caps = [], []
        t = batch_size
        for m in self.fanouts:
            self.target_caps.append(t)
            self.edge_caps.append(t * m)
            t = t + t * m
        lib = L.lib()
        ws = max(lib.wgamd_sample_hop_workspace_bytes(tc, ec, self.wm_dtype)
                 for tc, ec in zip(self.target_caps, self.edge_caps))
        self.workspace = torch.empty(ws + 256, dtype=torch.uint8, device=dev)
        self.ws_off = (-self.workspace.data_ptr()) % 256
        self.ws_bytes = ws
        self.dev = dev

    def run(self, seeds: torch.Tensor, random_seeds: List[int]) -> WalkResult:
        assert seeds.dtype == self.id_dtype and seeds.shape[0] == self.target_caps[0]
        lib, dev = L.lib(), self.dev
        hops = len(self.fanouts)
        counts = torch.empty((hops, 2), dtype=torch.int32, device=dev)
        n_seeds = torch.full((1,), seeds.shape[0], dtype=torch.int32, device=dev)
        res = WalkResult(hops, seeds, [], [], [], [], counts, self.target_caps)
        targets, n_dev_ptr = seeds, n_seeds.data_ptr()
        stream = get_stream()
        ws_ptr = self.workspace.data_ptr() + self.ws_off
        for k, (m, tc, ec) in enumerate(zip(self.fanouts, self.target_caps, self.edge_caps)):
            offsets = torch.empty(tc + 1, dtype=torch.int32, device=dev)
            nbr_lid = torch.empty(ec, dtype=torch.int32, device=dev)
            ctr_lid = torch.empty(ec, dtype=torch.int32, device=dev)
            unique = torch.empty(tc + ec, dtype=self.id_dtype, device=dev)
            L.check(lib.wgamd_sample_hop_nosync(
                self.row_ptr.data_ptr(), self.col.data_ptr(), self.wm_dtype, targets.data_ptr(), n_dev_ptr, tc, m,
                int(random_seeds[k]) & 0xFFFFFFFFFFFFFFFF, offsets.data_ptr(), nbr_lid.data_ptr(),
                ctr_lid.data_ptr(), None, ec, unique.data_ptr(), counts[k].data_ptr(), ws_ptr, self.ws_bytes,
                stream), "wgamd_sample_hop_nosync")
            res.unique.append(unique)
            res.offsets.append(offsets)
            res.neighbor_lid.append(nbr_lid)
            res.center_lid.append(ctr_lid)
            targets = unique
            n_dev_ptr = counts[k].data_ptr() + 4  # n_unique of this hop = #targets of the next
        res._keepalive = n_seeds
        return res
