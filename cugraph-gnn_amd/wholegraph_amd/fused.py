"""No-host-sync multi-hop walk (``wgamd_sample_hop_nosync`` / ``wgamd_sample_hop_batched_nosync``,
include/wgamd_ext.h), for one mini-batch or a *call group* of G mini-batches per launch sequence.

Pre-allocates capacity-sized buffers once per (G, batch size, fan-out) and replays the same launch
sequence for every call; all sizes live in small device tensors.  Every mini-batch of a call group
gets exactly the result of ``GraphStructure.multilayer_sample_without_replacement`` with its own
seeds (tests/test_gpu_renumber_gather.py, tests/test_gpu_callgroup.py).
"""
from dataclasses import dataclass, field
from typing import List, Sequence, Union

import torch

from . import _lib as L
from .env import get_stream, torch_dtype_to_wm


@dataclass
class WalkResult:
    """Capacity-sized outputs of one walk over a call group of ``n_batches`` mini-batches.

    Hop k (execution order, seed hop first) consumed targets ``target_seg[k]`` / produced
    ``unique[k]`` (per-batch unique lists, concatenated; batch b = rows
    ``[unique_seg[k][b], unique_seg[k][b+1])``).  ``offsets[k]`` is the CSR row_ptr over all hop-k
    targets, ``neighbor_row[k]`` the CSR col as GLOBAL rows of ``unique[k]`` (block-diagonal over the
    batches — exactly what a batched SpMM wants), ``center_row[k]`` the COO row.  ``counts[k] =
    [n_edges, n_unique]``.  Everything stays on the device until ``finalize``."""

    hops: int
    n_batches: int
    batch_size: int
    seeds: torch.Tensor
    unique: List[torch.Tensor] = field(default_factory=list)
    unique_seg: List[torch.Tensor] = field(default_factory=list)   # int32 [G+1] per hop
    target_seg: List[torch.Tensor] = field(default_factory=list)   # int32 [G+1] per hop (input segments)
    target_batch: List[torch.Tensor] = field(default_factory=list)  # int32 [target_cap] batch of every target
    offsets: List[torch.Tensor] = field(default_factory=list)      # int32 [target_cap+1]
    neighbor_row: List[torch.Tensor] = field(default_factory=list)  # int32 [edge_cap]
    center_row: List[torch.Tensor] = field(default_factory=list)    # int32 [edge_cap]
    counts: torch.Tensor = None                                     # int32 [hops, 2]
    target_caps: List[int] = field(default_factory=list)

    # single-batch aliases kept for the G == 1 users
    @property
    def neighbor_lid(self):
        return self.neighbor_row

    @property
    def center_lid(self):
        return self.center_row

    def target_rows_in_unique(self, k: int, n_targets: int) -> torch.Tensor:
        """Row (in ``unique[k]``) of every hop-k target: target i of batch b sits at
        ``i + (unique_seg[k][b] - target_seg[k][b])`` — the ``x[:num_dst]`` slice of the single-batch
        layout becomes this index list in the block-diagonal one."""
        shift = (self.unique_seg[k][:-1] - self.target_seg[k][:-1]).long()
        return torch.arange(n_targets, device=shift.device) + shift[self.target_batch[k][:n_targets].long()]

    def finalize_batches(self):
        """One round of small D2H copies, then per mini-batch the reference tuple
        ``(target_gids, edge_indice, csr_row_ptr, csr_col_ind)`` (graph_structure.py:186-196)."""
        hops, G = self.hops, self.n_batches
        tseg = [t.cpu().tolist() for t in self.target_seg]
        useg = [t.cpu().tolist() for t in self.unique_seg]
        eseg = [self.offsets[k][torch.as_tensor(tseg[k], device=self.offsets[k].device).long()].cpu().tolist()
                for k in range(hops)]
        out = []
        for b in range(G):
            target_gids = [None] * (hops + 1)
            edge_indice, csr_row_ptr, csr_col_ind = [None] * hops, [None] * hops, [None] * hops
            target_gids[hops] = self.seeds[b * self.batch_size:(b + 1) * self.batch_size]
            for k in range(hops):
                i = hops - 1 - k
                t0, t1 = tseg[k][b], tseg[k][b + 1]
                e0, e1 = eseg[k][b], eseg[k][b + 1]
                csr_row_ptr[i] = self.offsets[k][t0:t1 + 1] - e0
                csr_col_ind[i] = self.neighbor_row[k][e0:e1] - useg[k][b]
                edge_indice[i] = torch.stack([csr_col_ind[i], self.center_row[k][e0:e1] - t0])
                target_gids[i] = self.unique[k][useg[k][b]:useg[k][b + 1]]
            out.append((target_gids, edge_indice, csr_row_ptr, csr_col_ind))
        return out

    def finalize(self):
        """Single-batch convenience: the reference tuple of batch 0."""
        assert self.n_batches == 1
        return self.finalize_batches()[0]


class NoSyncWalk:
    def __init__(self, csr_row_ptr: torch.Tensor, csr_col_ind: torch.Tensor, batch_size: int,
                 max_neighbors: List[int], id_dtype=torch.int64, n_batches: int = 1):
        assert csr_row_ptr.is_cuda and csr_col_ind.is_cuda
        assert all(m > 0 for m in max_neighbors), "the no-sync walk needs positive fan-outs"
        assert csr_col_ind.dtype == id_dtype, "no-sync walk: seeds and csr_col must share a dtype"
        self.row_ptr, self.col = csr_row_ptr, csr_col_ind
        self.fanouts = list(max_neighbors)
        self.id_dtype = id_dtype
        self.wm_dtype = torch_dtype_to_wm(id_dtype)
        self.G, self.B = int(n_batches), int(batch_size)
        dev = csr_row_ptr.device
        self.target_caps, self.edge_caps = [], []
        t = self.G * self.B
        for m in self.fanouts:
            self.target_caps.append(t)
            self.edge_caps.append(t * m)
            t = t + t * m
        assert t < (1 << 30), "call group too large: lower n_batches"
        lib = L.lib()
        ws = max(lib.wgamd_sample_hop_workspace_bytes(tc, ec, self.wm_dtype)
                 for tc, ec in zip(self.target_caps, self.edge_caps))
        self.workspace = torch.empty(ws + 256, dtype=torch.uint8, device=dev)
        self.ws_off = (-self.workspace.data_ptr()) % 256
        self.ws_bytes = ws
        self.dev = dev
        # hop-0 segments are known up front: batch b = [b*B, (b+1)*B)
        self.seed_seg = (torch.arange(self.G + 1, dtype=torch.int32, device=dev) * self.B).contiguous()
        self.seed_batch = torch.arange(self.G, dtype=torch.int32, device=dev).repeat_interleave(self.B).contiguous()

    def _seeds_tensor(self, random_seeds) -> torch.Tensor:
        """-> int64 device tensor [hops, G] holding the 64-bit seeds (bit pattern)."""
        if isinstance(random_seeds, torch.Tensor):
            t = random_seeds.to(device=self.dev, dtype=torch.int64)
        else:
            rows = []
            for per_hop in random_seeds:
                vals = [per_hop] * self.G if isinstance(per_hop, int) else list(per_hop)
                rows.append([(int(v) & 0xFFFFFFFFFFFFFFFF) - (1 << 64) if (int(v) & (1 << 63)) else int(v) & 0xFFFFFFFFFFFFFFFF
                             for v in vals])
            t = torch.tensor(rows, dtype=torch.int64, device=self.dev)
        assert t.shape == (len(self.fanouts), self.G), f"random_seeds must be [hops={len(self.fanouts)}, G={self.G}]"
        return t.contiguous()

    def run(self, seeds: torch.Tensor, random_seeds: Union[torch.Tensor, Sequence]) -> WalkResult:
        """``seeds``: [G*B] ids (batch b = seeds[b*B:(b+1)*B]); ``random_seeds``: per hop (execution
        order) either one int (G == 1 / same seed for all batches) or G ints, or an int64 device
        tensor [hops, G]."""
        assert seeds.dtype == self.id_dtype and seeds.shape[0] == self.G * self.B
        lib, dev = L.lib(), self.dev
        hops = len(self.fanouts)
        rs = self._seeds_tensor(random_seeds)
        counts = torch.empty((hops, 2), dtype=torch.int32, device=dev)
        res = WalkResult(hops, self.G, self.B, seeds, counts=counts, target_caps=self.target_caps)
        res._keepalive = [rs]
        targets, t_batch, t_seg = seeds, self.seed_batch, self.seed_seg
        stream = get_stream()
        ws_ptr = self.workspace.data_ptr() + self.ws_off
        for k, (m, tc, ec) in enumerate(zip(self.fanouts, self.target_caps, self.edge_caps)):
            offsets = torch.empty(tc + 1, dtype=torch.int32, device=dev)
            nbr_row = torch.empty(ec, dtype=torch.int32, device=dev)
            ctr_row = torch.empty(ec, dtype=torch.int32, device=dev)
            unique = torch.empty(tc + ec, dtype=self.id_dtype, device=dev)
            u_batch = torch.empty(tc + ec, dtype=torch.int32, device=dev)
            u_seg = torch.empty(self.G + 1, dtype=torch.int32, device=dev)
            L.check(lib.wgamd_sample_hop_batched_nosync(
                self.row_ptr.data_ptr(), self.col.data_ptr(), self.wm_dtype, targets.data_ptr(), t_batch.data_ptr(),
                t_seg.data_ptr(), self.G, tc, m, rs[k].data_ptr(), offsets.data_ptr(), nbr_row.data_ptr(),
                ctr_row.data_ptr(), None, ec, unique.data_ptr(), u_batch.data_ptr(), u_seg.data_ptr(),
                counts[k].data_ptr(), ws_ptr, self.ws_bytes, stream), "wgamd_sample_hop_batched_nosync")
            res.unique.append(unique)
            res.unique_seg.append(u_seg)
            res.target_seg.append(t_seg)
            res.target_batch.append(t_batch)
            res.offsets.append(offsets)
            res.neighbor_row.append(nbr_row)
            res.center_row.append(ctr_row)
            targets, t_batch, t_seg = unique, u_batch, u_seg
        return res


class SingleBatchNoSyncWalk:
    """The G == 1 entry point ``wgamd_sample_hop_nosync`` (scalar seed by value); kept as the plain
    C-ABI form of the walk and exercised by the tests next to the batched one."""

    def __init__(self, csr_row_ptr, csr_col_ind, batch_size, max_neighbors, id_dtype=torch.int64):
        self.inner = NoSyncWalk(csr_row_ptr, csr_col_ind, batch_size, max_neighbors, id_dtype, 1)

    def run(self, seeds, random_seeds: Sequence[int]) -> WalkResult:
        w = self.inner
        lib, dev = L.lib(), w.dev
        hops = len(w.fanouts)
        counts = torch.empty((hops, 2), dtype=torch.int32, device=dev)
        n_seeds = torch.full((1,), seeds.shape[0], dtype=torch.int32, device=dev)
        res = WalkResult(hops, 1, w.B, seeds, counts=counts, target_caps=w.target_caps)
        res._keepalive = [n_seeds]
        targets, n_ptr = seeds, n_seeds.data_ptr()
        zero = torch.zeros(1, dtype=torch.int32, device=dev)
        t_seg = torch.cat([zero, n_seeds])
        for k, (m, tc, ec) in enumerate(zip(w.fanouts, w.target_caps, w.edge_caps)):
            offsets = torch.empty(tc + 1, dtype=torch.int32, device=dev)
            nbr = torch.empty(ec, dtype=torch.int32, device=dev)
            ctr = torch.empty(ec, dtype=torch.int32, device=dev)
            unique = torch.empty(tc + ec, dtype=w.id_dtype, device=dev)
            L.check(lib.wgamd_sample_hop_nosync(
                w.row_ptr.data_ptr(), w.col.data_ptr(), w.wm_dtype, targets.data_ptr(), n_ptr, tc, m,
                int(random_seeds[k]) & 0xFFFFFFFFFFFFFFFF, offsets.data_ptr(), nbr.data_ptr(), ctr.data_ptr(), None, ec,
                unique.data_ptr(), counts[k].data_ptr(), w.workspace.data_ptr() + w.ws_off, w.ws_bytes, get_stream()),
                "wgamd_sample_hop_nosync")
            res.unique.append(unique)
            res.offsets.append(offsets)
            res.neighbor_row.append(nbr)
            res.center_row.append(ctr)
            res.target_seg.append(t_seg)
            u_seg = torch.cat([zero, counts[k, 1:2]])
            res.unique_seg.append(u_seg)
            targets, n_ptr, t_seg = unique, counts[k].data_ptr() + 4, u_seg
        return res
