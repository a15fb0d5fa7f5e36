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


HOP_NO_UNIQUE_PAD = 1   # WGAMD_HOP_NO_UNIQUE_PAD (include/wgamd_ext.h)
HOP_COL_INT32 = 2       # WGAMD_HOP_COL_INT32

_COL32 = {}             # (data_ptr, numel) -> (weakref to the int64 column tensor, its int32 copy)


def compact_columns(col: torch.Tensor, n_vertices: int):
    """The 32-bit twin of an int64 CSR column array whose ids all fit (``n_vertices < 2^31``: every BASELINE graph), made
    once per column tensor and shared by every walk over it; ``None`` when there is nothing to compact.  The int64 API
    stays what callers see (seeds, ``unique``, ``n_id`` are int64): the walk's kernels read the columns — 15 M random picks
    per products call group, each its own 64-byte sector — from half the footprint and carry the sampled neighbours through
    the renumber passes as 32-bit values (``WGAMD_HOP_COL_INT32``).  Costs 4 bytes per edge of extra HBM; the tensor must not
    be modified in place afterwards."""
    import weakref
    if col.dtype != torch.int64 or not (0 < int(n_vertices) < (1 << 31)) or not col.is_cuda:
        return None
    key = (col.data_ptr(), col.numel())
    hit = _COL32.get(key)
    if hit is not None and hit[0]() is col:
        return hit[1]
    for k in [k for k, v in _COL32.items() if v[0]() is None]:
        del _COL32[k]
    c32 = col.to(torch.int32)
    _COL32[key] = (weakref.ref(col), c32)
    return c32


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
        rows = torch.empty((n_targets,), dtype=torch.int64, device=self.unique_seg[k].device)
        L.check(L.lib().wgamd_call_group_target_rows(self.unique_seg[k].data_ptr(), self.target_seg[k].data_ptr(),
                                                     self.target_batch[k].data_ptr(), int(n_targets), rows.data_ptr(),
                                                     get_stream()), "wgamd_call_group_target_rows")
        return rows

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

    def finalize_single(self, copy: bool = False):
        """``finalize()`` for one mini-batch with ONE host read-back (the ``counts`` vector: a single batch's segments are
        ``[0, n]``, so every size follows from it).  ``copy``: return fresh tensors instead of views of the walk's buffers
        (a captured walk overwrites them on its next run)."""
        assert self.n_batches == 1
        hops = self.hops
        sz = self.counts.cpu().tolist()                       # per hop: [edges, unique vertices after the hop]
        keep = (lambda t: t.clone()) if copy else (lambda t: t)
        target_gids = [None] * hops + [self.seeds]
        edge_indice, csr_row_ptr, csr_col_ind = [None] * hops, [None] * hops, [None] * hops
        n_t = int(self.seeds.shape[0])
        for k in range(hops):
            i = hops - 1 - k
            n_e, n_u = sz[k]
            csr_row_ptr[i] = keep(self.offsets[k][:n_t + 1])
            csr_col_ind[i] = keep(self.neighbor_row[k][:n_e])
            edge_indice[i] = torch.stack([csr_col_ind[i], self.center_row[k][:n_e]])
            target_gids[i] = keep(self.unique[k][:n_u])
            n_t = n_u
        return target_gids, edge_indice, csr_row_ptr, csr_col_ind


def _merge_hops_batch_major(fields_per_hop, segs_per_hop, G, dev):
    """Per-hop arrays that are each batch-major (hop h, batch b = [segs[h][b], segs[h][b+1])) -> ONE array per field laid
    out [batch][hop], plus the host offsets of the batches in it.  One scatter for the whole call group
    per hop instead of a ``torch.cat`` per mini-batch and field (the per-batch loop then only takes views)."""
    H = len(fields_per_hop)
    n_fields = len(fields_per_hop[0]) if H else 0
    tot, offs = [0] * G, [0] * (G + 1)
    for s_ in segs_per_hop:
        for b in range(G):
            tot[b] += s_[b + 1] - s_[b]
    for b in range(G):
        offs[b + 1] = offs[b] + tot[b]
    if H == 0:
        return [], offs
    live = [[f[s_[0]:s_[G]] for f in fs] for fs, s_ in zip(fields_per_hop, segs_per_hop)]
    if H == 1:
        return live[0], offs
    # element j of hop h (batch b) lands at  j + shift[h][b],  shift = start of batch b in the merged array + what the
    # earlier hops put there - start of the batch in the hop's own array: one scatter per hop and field, no sort
    shift, before = [], [0] * G
    for s_ in segs_per_hop:
        shift.append([offs[b] + before[b] - s_[b] for b in range(G)])
        before = [before[b] + s_[b + 1] - s_[b] for b in range(G)]
    meta = torch.tensor([[s_[b + 1] - s_[b] for b in range(G)] for s_ in segs_per_hop] + shift, dtype=torch.int64).to(dev)
    batches = torch.arange(G, device=dev)
    merged = [torch.empty(offs[G], dtype=live[0][i].dtype, device=dev) for i in range(n_fields)]
    for h in range(H):
        n_h = segs_per_hop[h][G] - segs_per_hop[h][0]
        if n_h == 0:
            continue
        b_of = torch.repeat_interleave(batches, meta[h], output_size=n_h)
        dest = torch.arange(segs_per_hop[h][0], segs_per_hop[h][G], device=dev) + meta[H + h][b_of]
        for i in range(n_fields):
            merged[i][dest] = live[h][i]
    return merged, offs


class NoSyncWalk:
    def __init__(self, csr_row_ptr: torch.Tensor, csr_col_ind: torch.Tensor, batch_size: int,
                 max_neighbors: List[int], id_dtype=torch.int64, n_batches: int = 1, pad_unique: bool = True,
                 compact_col: bool = True):
        """``pad_unique=False``: the capacity slack of every hop's ``unique`` list is left unwritten instead of padded with
        -1 (``WGAMD_HOP_NO_UNIQUE_PAD``) — for consumers that slice by ``counts`` / ``unique_seg``; the capacity is 3-4x the
        live size, so the padding is most of what the renumber step writes.  ``compact_col``: int64 columns of a graph
        with fewer than 2^31 vertices are read through their 32-bit twin (``compact_columns``); same results."""
        assert csr_row_ptr.is_cuda and csr_col_ind.is_cuda
        assert all(m > 0 for m in max_neighbors), "the no-sync walk needs positive fan-outs"
        self.flags = 0 if pad_unique else HOP_NO_UNIQUE_PAD
        assert csr_col_ind.dtype == id_dtype, "no-sync walk: seeds and csr_col must share a dtype"
        self.row_ptr, self.col = csr_row_ptr, csr_col_ind
        self.n_vertices = int(csr_row_ptr.shape[0]) - 1     # every id is a row of the CSR
        c32 = compact_columns(csr_col_ind, self.n_vertices) if compact_col else None
        if c32 is not None:
            self.col, self.flags = c32, self.flags | HOP_COL_INT32
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
            # (the batch of every unique entry is the next hop's target_batch: the last hop's list has no reader)
            u_batch = torch.empty(tc + ec, dtype=torch.int32, device=dev) if k + 1 < hops else None
            u_seg = torch.empty(self.G + 1, dtype=torch.int32, device=dev)
            L.check(lib.wgamd_sample_hop_batched_nosync_ex(
                self.row_ptr.data_ptr(), self.col.data_ptr(), self.wm_dtype, targets.data_ptr(), t_batch.data_ptr(),
                t_seg.data_ptr(), self.G, tc, m, rs[k].data_ptr(), offsets.data_ptr(), nbr_row.data_ptr(),
                ctr_row.data_ptr(), None, ec, unique.data_ptr(), None if u_batch is None else u_batch.data_ptr(),
                u_seg.data_ptr(),
                counts[k].data_ptr(), ws_ptr, self.ws_bytes, self.n_vertices, self.flags, stream),
                "wgamd_sample_hop_batched_nosync_ex")
            res.unique.append(unique)
            res.unique_seg.append(u_seg)
            res.target_seg.append(t_seg)
            res.target_batch.append(t_batch)
            res.offsets.append(offsets)
            res.neighbor_row.append(nbr_row)
            res.center_row.append(ctr_row)
            targets, t_batch, t_seg = unique, u_batch, u_seg
        return res


class CapturedWalk:
    """A ``NoSyncWalk`` whose launch sequence (~11 kernels per hop) is captured ONCE into a HIP graph
    (``torch.cuda.CUDAGraph``) and replayed: a single mini-batch is a few tens of microseconds of device work behind
    ~0.2 ms of launch calls, so the reference's one-batch calling convention (``GraphStructure.multilayer_sample_without_
    replacement``, graph_structure.py:136-196) is bound by the host — one graph launch instead of 22 kernel launches takes
    that away.  Inputs are copied into static buffers, the result lives in static buffers until the next ``run`` (callers
    copy what they keep).  Uniform sampling over a CSR this GPU holds whole; identical results (the same kernels)."""

    def __init__(self, walk: NoSyncWalk):
        self.walk = walk
        self._graph = self._res = None
        hops = len(walk.fanouts)
        self._seeds = torch.zeros(walk.G * walk.B, dtype=walk.id_dtype, device=walk.dev)
        self._rs = torch.zeros((hops, walk.G), dtype=torch.int64, device=walk.dev)
        self._rs_host = torch.zeros((hops, walk.G), dtype=torch.int64).pin_memory()

    def run(self, seeds: torch.Tensor, random_seeds) -> WalkResult:
        w = self.walk
        if isinstance(random_seeds, torch.Tensor):
            self._rs.copy_(random_seeds.to(torch.int64).view_as(self._rs), non_blocking=True)
        else:
            for k, per_hop in enumerate(random_seeds):
                vals = [per_hop] * w.G if isinstance(per_hop, int) else list(per_hop)
                for b, v in enumerate(vals):
                    v = int(v) & 0xFFFFFFFFFFFFFFFF
                    self._rs_host[k, b] = v - (1 << 64) if v & (1 << 63) else v
            self._rs.copy_(self._rs_host, non_blocking=True)
        self._seeds.copy_(seeds, non_blocking=True)
        if self._graph is None:
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream(device=w.dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):          # an eager pass first (library and allocator warm-up), as capture requires
                w.run(self._seeds, self._rs)
            cur.wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._res = w.run(self._seeds, self._rs)
            self._graph = g
        self._graph.replay()
        return self._res


class SingleBatchNoSyncWalk:
    """The G == 1 entry point ``wgamd_sample_hop_nosync`` (scalar seed by value); kept as the plain
    C-ABI form of the walk and exercised by the tests next to the batched one."""

    def __init__(self, csr_row_ptr, csr_col_ind, batch_size, max_neighbors, id_dtype=torch.int64):
        # (the scalar-seed entry point has no flags argument: it reads the columns in the ids' own width)
        self.inner = NoSyncWalk(csr_row_ptr, csr_col_ind, batch_size, max_neighbors, id_dtype, 1, compact_col=False)

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


# --------------------------------------------------------------------------------------------------
# PyG-style call-group walk (expand only the vertices first seen by the previous hop)
# --------------------------------------------------------------------------------------------------
import ctypes as _ct


class _PygHop(_ct.Structure):
    """ctypes mirror of ``wgamd_pyg_hop_t`` (include/wgamd_ext.h)."""
    _fields_ = [("csr_row_ptr", _ct.c_void_p), ("csr_col", _ct.c_void_p), ("id_dtype", _ct.c_int),
                ("n_batches", _ct.c_int), ("max_sample_count", _ct.c_int), ("random_seeds_dev", _ct.c_void_p),
                ("nodes", _ct.c_void_p), ("node_batch", _ct.c_void_p), ("node_seg", _ct.c_void_p),
                ("node_cap", _ct.c_int64),
                ("frontier", _ct.c_void_p), ("frontier_batch", _ct.c_void_p), ("frontier_seg", _ct.c_void_p),
                ("frontier_local0", _ct.c_void_p), ("frontier_cap", _ct.c_int64),
                ("offsets", _ct.c_void_p), ("neighbor_local", _ct.c_void_p), ("center_local", _ct.c_void_p),
                ("edge_gid", _ct.c_void_p), ("edge_cap", _ct.c_int64),
                ("nodes_out", _ct.c_void_p), ("nodes_out_batch", _ct.c_void_p), ("nodes_out_seg", _ct.c_void_p),
                ("frontier_out", _ct.c_void_p), ("frontier_out_batch", _ct.c_void_p),
                ("frontier_out_seg", _ct.c_void_p), ("frontier_out_local0", _ct.c_void_p),
                ("counts_dev", _ct.c_void_p), ("neighbor_row_scratch", _ct.c_void_p),
                ("center_row_scratch", _ct.c_void_p), ("workspace", _ct.c_void_p), ("workspace_bytes", _ct.c_size_t),
                ("csr_weight", _ct.c_void_p), ("weight_dtype", _ct.c_int), ("max_row_len", _ct.c_int64),
                ("n_vertices", _ct.c_int64), ("flags", _ct.c_uint)]


@dataclass
class PygWalkResult:
    """Capacity-sized outputs of one PyG-style walk over a call group (everything on the device).
    Hop k: ``offsets[k]`` (CSR over that hop's frontier), ``row_local[k]`` / ``col_local[k]`` (per-batch local
    ids of the sampled neighbour / the expanded vertex), ``edge_gid[k]`` (CSR slot), ``frontier_seg[k]``
    (input frontier segments; hop 0: b*B).  ``nodes`` / ``node_seg``: the final per-batch vertex lists."""
    hops: int
    n_batches: int
    batch_size: int
    nodes: torch.Tensor = None
    node_seg: torch.Tensor = None
    offsets: List[torch.Tensor] = field(default_factory=list)
    row_local: List[torch.Tensor] = field(default_factory=list)
    col_local: List[torch.Tensor] = field(default_factory=list)
    edge_gid: List[torch.Tensor] = field(default_factory=list)
    frontier_seg: List[torch.Tensor] = field(default_factory=list)
    frontier_batch: List[torch.Tensor] = field(default_factory=list)    # int32 [frontier_cap] batch of every frontier entry
    frontier_local0: List[torch.Tensor] = field(default_factory=list)   # int32 [G] local id of a batch's first frontier entry
    counts: torch.Tensor = None

    def finalize_batches(self, edge_id: torch.Tensor = None):
        """Per mini-batch ``(node, row, col, edge, num_sampled_nodes, num_sampled_edges)`` — the tuple of
        ``cugraph_pyg_amd.sampler.neighbor_sample``.  ``edge_id`` maps CSR slots to original edge ids."""
        G, hops = self.n_batches, self.hops
        # one D2H copy for every size vector of the call group (each .cpu() is a stream sync)
        pieces = list(self.frontier_seg) + [self.node_seg] + [self.offsets[k][self.frontier_seg[k].long()] for k in range(hops)]
        flat = torch.cat([p_.reshape(-1).to(torch.int64) for p_ in pieces]).cpu().tolist()
        host, at = [], 0
        for p_ in pieces:
            host.append(flat[at:at + p_.numel()])
            at += p_.numel()
        fseg = host[:hops + 1]                                       # hops+1 lists (the last one = final new counts)
        nseg = host[hops + 1]
        eseg = host[hops + 2:]
        # dtype conversion, edge-id lookup and the hop concatenation happen ONCE for the call group
        fields = []
        for k in range(hops):
            lo, hi = eseg[k][0], eseg[k][G]
            gid = self.edge_gid[k]
            fields.append([self.row_local[k][:hi].long(), self.col_local[k][:hi].long(),
                           edge_id[gid[:hi]] if edge_id is not None else gid[:hi]])
        (rows, cols, edges), offs = _merge_hops_batch_major(fields, eseg, G, self.nodes.device)
        sizes_b = [offs[b + 1] - offs[b] for b in range(G)]
        node_v = torch.split(self.nodes[nseg[0]:nseg[G]], [nseg[b + 1] - nseg[b] for b in range(G)])
        row_v, col_v, edge_v = (torch.split(t[:offs[G]], sizes_b) for t in (rows, cols, edges))
        # the call group as a whole, for consumers that fetch per-node / per-edge attributes once per group: every
        # per-batch tensor above is a view into these, batch b = [offsets[b], offsets[b+1])
        self.group_context = dict(nodes=self.nodes[nseg[0]:nseg[G]], node_sizes=[nseg[b + 1] - nseg[b] for b in range(G)],
                                  edges=edges[:offs[G]], edge_sizes=sizes_b)
        # Every mini-batch's ``edge_index`` and its destination-major CSR, made ONCE for the call group (a dozen launches
        # instead of four per mini-batch and layer: the per-batch loop of a PyG script is bound by exactly that host work).
        # The merged edge list of a batch is hop after hop, a hop's edges in the order of its frontier: sorted by destination.
        E, dev = offs[G], self.nodes.device
        if E > 0 and rows.dtype == torch.int64:
            n_b = [nseg[b + 1] - nseg[b] for b in range(G)]
            Nn = nseg[G] - nseg[0]
            meta = torch.tensor([sizes_b, n_b, [v + 1 for v in n_b], [nseg[b] - nseg[0] for b in range(G)], offs[:G]],
                                dtype=torch.int64).to(dev)
            ar = torch.arange(G, device=dev)
            ei_all = torch.stack([rows[:E], cols[:E]])
            b_of_e = torch.repeat_interleave(ar, meta[0], output_size=E)
            keys = cols[:E] + meta[3][b_of_e]                                    # destination as a row of the whole group
            rp_all = torch.searchsorted(keys, torch.arange(Nn + 1, device=dev))
            b_of_p = torch.repeat_interleave(ar, meta[2], output_size=Nn + G)    # per batch n_b + 1 row-pointer entries
            rp_cat = (rp_all[torch.arange(Nn + G, device=dev) - b_of_p] - meta[4][b_of_p]).to(torch.int32)
            col32 = rows[:E].to(torch.int32)
            self.group_context["edge_index"] = torch.split(ei_all, sizes_b, dim=1)
            self.group_context["csr"] = list(zip(torch.split(rp_cat, [v + 1 for v in n_b]), torch.split(col32, sizes_b)))
        out, nn_all, ne_all = [], [], []
        for b in range(G):
            nn = [fseg[0][b + 1] - fseg[0][b]] + [fseg[k + 1][b + 1] - fseg[k + 1][b] for k in range(hops)]
            ne = [eseg[k][b + 1] - eseg[k][b] for k in range(hops)]
            nn_all.append(nn), ne_all.append(ne)
            out.append((node_v[b], row_v[b], col_v[b], edge_v[b], nn, ne))
        # (num_sampled_nodes / num_sampled_edges of every batch as rows of ONE host tensor: torch.tensor() of a short list
        #  costs ~20 us, twice per mini-batch)
        self.group_context["num_sampled_nodes"] = torch.tensor(nn_all).unbind(0)
        self.group_context["num_sampled_edges"] = torch.tensor(ne_all).unbind(0)
        return out


class PygNoSyncWalk:
    """Call-group sampler with PyG hop semantics on ``wgamd_sample_hop_pyg_nosync`` — the MI355X-native
    counterpart of one ``pylibcugraph.homogeneous_uniform_neighbor_sample`` call over many batches
    (SURVEY.md §8 row a14)."""

    def __init__(self, csr_row_ptr, csr_col_ind, batch_size: int, fanout: List[int], n_batches: int = 1,
                 csr_weight: torch.Tensor = None, pad_unique: bool = True, compact_col: bool = True):
        """``csr_weight`` (float32 | float64, one per CSR slot): BIASED sampling — every hop is then what
        ``wholegraph_csr_weighted_sample_without_replacement`` draws (fan-outs <= 256).  ``pad_unique`` / ``compact_col``:
        see ``NoSyncWalk``."""
        self.flags = 0 if pad_unique else HOP_NO_UNIQUE_PAD
        assert csr_row_ptr.is_cuda and csr_col_ind.is_cuda and csr_col_ind.dtype in (torch.int32, torch.int64)
        assert all(0 < f for f in fanout), "the no-sync walk needs positive fan-outs"
        self.weight, self.max_row_len = None, 0
        if csr_weight is not None:
            assert csr_weight.is_cuda and csr_weight.dtype in (torch.float32, torch.float64) and all(f <= 256 for f in fanout)
            assert csr_weight.shape[0] == csr_col_ind.shape[0]
            self.weight = csr_weight.contiguous()
            self.max_row_len = max(int((csr_row_ptr[1:] - csr_row_ptr[:-1]).max()), 1) if csr_row_ptr.shape[0] > 1 else 1
        self.row_ptr, self.col = csr_row_ptr, csr_col_ind
        self.id_dtype, self.wm_dtype = csr_col_ind.dtype, torch_dtype_to_wm(csr_col_ind.dtype)
        c32 = compact_columns(csr_col_ind, int(csr_row_ptr.shape[0]) - 1) if compact_col else None
        if c32 is not None:
            self.col, self.flags = c32, self.flags | HOP_COL_INT32
        self.G, self.B, self.fanout = int(n_batches), int(batch_size), [int(f) for f in fanout]
        dev = csr_row_ptr.device
        self.frontier_caps, self.edge_caps, self.node_caps = [], [], []
        f, n = self.G * self.B, self.G * self.B
        for m in self.fanout:
            self.frontier_caps.append(f)
            self.node_caps.append(n)
            self.edge_caps.append(f * m)
            n, f = n + f * m, f * m
        assert n < (1 << 30), "call group too large: lower n_batches"
        lib = L.lib()
        if self.weight is None:
            ws = max(lib.wgamd_sample_hop_workspace_bytes(max(nc, fc), ec, self.wm_dtype)
                     for nc, fc, ec in zip(self.node_caps, self.frontier_caps, self.edge_caps))
        else:
            ws = max(lib.wgamd_sample_hop_weighted_workspace_bytes(max(nc, fc), ec, self.wm_dtype, self.max_row_len)
                     for nc, fc, ec in zip(self.node_caps, self.frontier_caps, self.edge_caps))
        self.workspace = torch.empty(ws + 256, dtype=torch.uint8, device=dev)
        self.ws_off, self.ws_bytes, self.dev = (-self.workspace.data_ptr()) % 256, ws, dev
        self.seed_seg = (torch.arange(self.G + 1, dtype=torch.int32, device=dev) * self.B).contiguous()
        self.seed_batch = torch.arange(self.G, dtype=torch.int32, device=dev).repeat_interleave(self.B).contiguous()
        self.zeros_g = torch.zeros(self.G, dtype=torch.int32, device=dev)

    def run(self, seeds: torch.Tensor, random_seeds, seed_seg: torch.Tensor = None, seed_batch: torch.Tensor = None) -> PygWalkResult:
        """``seeds`` [G*B] (batch b = seeds[b*B:(b+1)*B]); ``random_seeds`` int64 tensor / nested list [hops, G].
        Ragged seed lists (link prediction: the de-duplicated endpoints of a batch's seed edges): ``seeds`` holds the lists
        back to back, padded to the capacity G*B, ``seed_seg`` int32 [G+1] their offsets and ``seed_batch`` int32 [G*B]
        the batch of every live entry — the kernels only read below ``seed_seg[G]``."""
        assert seeds.dtype == self.id_dtype and seeds.shape[0] == self.G * self.B
        assert (seed_seg is None) == (seed_batch is None)
        lib, dev, G = L.lib(), self.dev, self.G
        hops = len(self.fanout)
        if isinstance(random_seeds, torch.Tensor):
            rs = random_seeds.to(device=dev, dtype=torch.int64).contiguous()
        else:
            rs = torch.tensor([[v - (1 << 64) if v & (1 << 63) else v for v in (int(x) & 0xFFFFFFFFFFFFFFFF for x in row)]
                               for row in random_seeds], dtype=torch.int64, device=dev)
        assert rs.shape == (hops, G)
        res = PygWalkResult(hops, G, self.B, counts=torch.empty((hops, 2), dtype=torch.int32, device=dev))
        res._keepalive = [rs]
        if seed_seg is None:
            seed_seg, seed_batch = self.seed_seg, self.seed_batch
        else:
            assert seed_seg.dtype == torch.int32 and seed_seg.shape[0] == G + 1 and seed_seg.is_contiguous()
            assert seed_batch.dtype == torch.int32 and seed_batch.shape[0] == G * self.B and seed_batch.is_contiguous()
        nodes, n_batch, n_seg = seeds, seed_batch, seed_seg
        front, f_batch, f_seg, f_local0 = seeds, seed_batch, seed_seg, self.zeros_g
        i32 = dict(dtype=torch.int32, device=dev)
        for k, (m, fc, nc, ec) in enumerate(zip(self.fanout, self.frontier_caps, self.node_caps, self.edge_caps)):
            offsets = torch.empty(fc + 1, **i32)
            row_l, col_l = torch.empty(ec, **i32), torch.empty(ec, **i32)
            scratch_r, scratch_c = torch.empty(ec, **i32), torch.empty(ec, **i32)
            gid = torch.empty(ec, dtype=torch.int64, device=dev)
            nodes_out = torch.empty(nc + ec, dtype=self.id_dtype, device=dev)
            nodes_out_batch, nodes_out_seg = torch.empty(nc + ec, **i32), torch.empty(G + 1, **i32)
            f_out = torch.empty(ec, dtype=self.id_dtype, device=dev)
            f_out_batch, f_out_seg, f_out_l0 = torch.empty(ec, **i32), torch.empty(G + 1, **i32), torch.empty(G, **i32)
            p = _PygHop(self.row_ptr.data_ptr(), self.col.data_ptr(), self.wm_dtype, G, m, rs[k].data_ptr(),
                        nodes.data_ptr(), n_batch.data_ptr(), n_seg.data_ptr(), nc,
                        front.data_ptr(), f_batch.data_ptr(), f_seg.data_ptr(), f_local0.data_ptr(), fc,
                        offsets.data_ptr(), row_l.data_ptr(), col_l.data_ptr(), gid.data_ptr(), ec,
                        nodes_out.data_ptr(), nodes_out_batch.data_ptr(), nodes_out_seg.data_ptr(),
                        f_out.data_ptr(), f_out_batch.data_ptr(), f_out_seg.data_ptr(), f_out_l0.data_ptr(),
                        res.counts[k].data_ptr(), scratch_r.data_ptr(), scratch_c.data_ptr(),
                        self.workspace.data_ptr() + self.ws_off, self.ws_bytes,
                        None if self.weight is None else self.weight.data_ptr(),
                        0 if self.weight is None else torch_dtype_to_wm(self.weight.dtype), self.max_row_len,
                        int(self.row_ptr.shape[0]) - 1, self.flags)
            L.check(lib.wgamd_sample_hop_pyg_nosync(_ct.byref(p), get_stream()), "wgamd_sample_hop_pyg_nosync")
            res.offsets.append(offsets)
            res.row_local.append(row_l)
            res.col_local.append(col_l)
            res.edge_gid.append(gid)
            res.frontier_seg.append(f_seg)
            res.frontier_batch.append(f_batch)
            res.frontier_local0.append(f_local0)
            res._keepalive += [scratch_r, scratch_c, f_batch, n_batch, front, f_local0]
            nodes, n_batch, n_seg = nodes_out, nodes_out_batch, nodes_out_seg
            front, f_batch, f_seg, f_local0 = f_out, f_out_batch, f_out_seg, f_out_l0
        res.frontier_seg.append(f_seg)      # new vertices of the last hop (for num_sampled_nodes)
        res.nodes, res.node_seg = nodes, n_seg
        return res


class HeteroPygWalk:
    """Call-group sampler for HETEROGENEOUS graphs on ``wgamd_sample_hop_pyg_nosync`` (one call per hop and edge type
    for all G mini-batches): the frontier of the edge type's destination node type is sampled on that type's CSR and
    the neighbours are renumbered against the per-batch vertex lists of the source node type.  Produces, per
    mini-batch, exactly what ``cugraph_pyg_amd.sampler.hetero_neighbor_sample`` returns for that batch alone (same
    per-batch seeds; tests/test_gpu_pyg_loader.py) — the counterpart of one
    ``pylibcugraph.heterogeneous_uniform_neighbor_sample`` call over many batches.
    No host synchronisation inside ``run``; sizes are read once in ``finalize_batches``."""

    def __init__(self, graphs, batch_size: int, fanout, n_batches: int, biased: bool = False, num_nodes=None,
                 pad_unique: bool = True):
        """``num_nodes``: {node type: vertex count} — with it the renumber table of every hop packs (batch, id, position)
        into one word (ids of a type are then known to be below its count).  ``pad_unique``: see ``NoSyncWalk``."""
        self.flags = 0 if pad_unique else HOP_NO_UNIQUE_PAD
        self.biased = bool(biased)
        self.num_nodes = dict(num_nodes) if num_nodes else {}
        self.etypes = sorted(graphs.keys())
        self.graphs = graphs
        self.G, self.B = int(n_batches), int(batch_size)
        self.fanout = {et: [int(f) for f in fanout.get(et, [])] for et in self.etypes}
        self.hops = len(next(iter(fanout.values())))
        for et in self.etypes:
            if not self.fanout[et]:
                self.fanout[et] = [0] * self.hops
            assert len(self.fanout[et]) == self.hops and all(f >= 0 for f in self.fanout[et])
            assert graphs[et].col.dtype == torch.int64 and graphs[et].row_ptr.is_cuda
        self.ntypes = sorted({t for et in self.etypes for t in (et[0], et[2])})
        self.dev = graphs[self.etypes[0]].row_ptr.device
        self.wm_dtype = torch_dtype_to_wm(torch.int64)
        self._ws = None
        dev = self.dev
        self.seed_seg = (torch.arange(self.G + 1, dtype=torch.int32, device=dev) * self.B).contiguous()
        self.seed_batch = torch.arange(self.G, dtype=torch.int32, device=dev).repeat_interleave(self.B).contiguous()
        self.zeros_g = torch.zeros(self.G, dtype=torch.int32, device=dev)
        self.zeros_g1 = torch.zeros(self.G + 1, dtype=torch.int32, device=dev)
        # 32-bit twins of the column arrays (type-local ids of the SOURCE type) where that type's vertex count is known
        self.col32 = {et: compact_columns(graphs[et].col, int(self.num_nodes.get(et[0], 0))) for et in self.etypes}
        self.max_row_len = {}
        if self.biased:   # per edge type: weights + maximum degree (sizes the key slabs of the biased hop)
            for et in self.etypes:
                g = graphs[et]
                assert g.weight is not None and g.weight.dtype in (torch.float32, torch.float64)
                assert all(f <= 256 for f in self.fanout[et])
                self.max_row_len[et] = max(int((g.row_ptr[1:] - g.row_ptr[:-1]).max()), 1) if g.row_ptr.shape[0] > 1 else 1

    def _workspace(self, node_cap, edge_cap, max_row_len=0):
        if max_row_len > 0:
            need = L.lib().wgamd_sample_hop_weighted_workspace_bytes(node_cap, edge_cap, self.wm_dtype, max_row_len)
        else:
            need = L.lib().wgamd_sample_hop_workspace_bytes(node_cap, edge_cap, self.wm_dtype)
        if self._ws is None or self._ws.numel() < need + 256:
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.dev)
        off = (-self._ws.data_ptr()) % 256
        return self._ws.data_ptr() + off, self._ws.numel() - off

    def _frontier(self, st, cap):
        """Batch-major list of the vertices a type gained since ``st['begin']`` (per-batch local index), capacity
        ``cap``: (ids, batch, seg, local0).  Entries past the live total are padding the kernels never read."""
        G, dev = self.G, self.dev
        if G <= 4095 and st["nodes"].shape[0] >= 1 and st["seg"].dtype == torch.int32 and st["begin"].dtype == torch.int32:
            # one kernel (wgamd_frontier_list) instead of the thirteen framework ops below
            nodes, seg, begin = st["nodes"].contiguous(), st["seg"].contiguous(), st["begin"].contiguous()
            ids = torch.empty(cap, dtype=torch.int64, device=dev)
            b = torch.empty(cap, dtype=torch.int32, device=dev)
            f_seg = torch.empty(G + 1, dtype=torch.int32, device=dev)
            L.check(L.lib().wgamd_frontier_list(nodes.data_ptr(), int(nodes.shape[0]), seg.data_ptr(), begin.data_ptr(), G, int(cap),
                                                ids.data_ptr(), b.data_ptr(), f_seg.data_ptr(), get_stream()), "wgamd_frontier_list")
            return ids, b, f_seg, begin
        size_b = st["seg"][1:] - st["seg"][:-1]
        cnt = (size_b - st["begin"]).to(torch.int32)
        f_seg = torch.zeros(G + 1, dtype=torch.int32, device=dev)
        f_seg[1:] = torch.cumsum(cnt, 0)
        p = torch.arange(cap, dtype=torch.int32, device=dev)
        b = torch.searchsorted(f_seg[1:].contiguous(), p, right=True).clamp_(max=G - 1)
        src = st["seg"][:-1][b].long() + st["begin"][b].long() + (p - f_seg[:-1][b]).long()
        ids = st["nodes"][src.clamp_(0, st["nodes"].shape[0] - 1)]
        return ids.contiguous(), b.to(torch.int32).contiguous(), f_seg, st["begin"].contiguous()

    def run(self, seed_type: str, seeds: torch.Tensor, random_seeds: torch.Tensor, seed_lists=None):
        """``seeds`` [G*B] type-local ids of ``seed_type``; ``random_seeds`` int64 [hops * n_etypes, G]: row
        ``h * n_etypes + t`` holds the per-batch seeds of hop h / edge type t (sorted order).
        ``seed_lists`` (link prediction: both endpoint types of the seed edges start the walk, with ragged per-batch lists):
        {node type: (ids padded to a capacity, int32 offsets [G+1], int32 batch of every live entry)} replaces
        ``seed_type`` / ``seeds``."""
        lib, dev, G = L.lib(), self.dev, self.G
        assert seed_lists is not None or (seeds.dtype == torch.int64 and seeds.shape[0] == G * self.B and seeds.is_cuda)
        rs = random_seeds.to(device=dev, dtype=torch.int64).contiguous()
        assert rs.shape == (self.hops * len(self.etypes), G)
        i32 = dict(dtype=torch.int32, device=dev)
        state = {}
        for t in self.ntypes:   # per node type: batch-major vertex lists, capacity, start of the current frontier
            state[t] = dict(nodes=torch.zeros(1, dtype=torch.int64, device=dev), batch=torch.zeros(1, **i32),
                            seg=self.zeros_g1, cap=0, begin=self.zeros_g, gained_cap=0)
        if seed_lists is None:
            state[seed_type] = dict(nodes=seeds, batch=self.seed_batch, seg=self.seed_seg, cap=G * self.B,
                                    begin=self.zeros_g, gained_cap=G * self.B)
        else:
            for t, (ids, seg, batch) in seed_lists.items():
                assert ids.dtype == torch.int64 and seg.dtype == torch.int32 and batch.dtype == torch.int32
                assert seg.shape[0] == G + 1 and batch.shape[0] == ids.shape[0]
                state[t] = dict(nodes=ids.contiguous(), batch=batch.contiguous(), seg=seg.contiguous(), cap=int(ids.shape[0]),
                                begin=self.zeros_g, gained_cap=int(ids.shape[0]))
        rec = dict(calls=[], sizes=[{t: (state[t]["seg"][1:] - state[t]["seg"][:-1]) for t in self.ntypes}])
        keep = [rs]
        for h in range(self.hops):
            fronts = {t: (self._frontier(state[t], state[t]["gained_cap"]) if state[t]["gained_cap"] > 0 else None)
                      for t in self.ntypes}
            f_caps = {t: state[t]["gained_cap"] for t in self.ntypes}
            for t in self.ntypes:   # what this hop adds becomes the next frontier
                st = state[t]
                st["begin"] = (st["seg"][1:] - st["seg"][:-1]).to(torch.int32).contiguous()
                st["gained_cap"] = 0
            for ti, et in enumerate(self.etypes):
                src_t, _, dst_t = et
                m, fr = self.fanout[et][h], fronts[dst_t]
                if m == 0 or fr is None:
                    rec["calls"].append(None)
                    continue
                g = self.graphs[et]
                f_ids, f_batch, f_seg, f_l0 = fr
                fc = f_caps[dst_t]
                ec = fc * m
                st = state[src_t]
                nc = max(st["cap"], 1)
                offsets = torch.empty(fc + 1, **i32)
                row_l, col_l = torch.empty(ec, **i32), torch.empty(ec, **i32)
                scratch_r, scratch_c = torch.empty(ec, **i32), torch.empty(ec, **i32)
                gid = torch.empty(ec, dtype=torch.int64, device=dev)
                nodes_out = torch.empty(nc + ec, dtype=torch.int64, device=dev)
                nodes_out_batch, nodes_out_seg = torch.empty(nc + ec, **i32), torch.empty(G + 1, **i32)
                f_out = torch.empty(ec, dtype=torch.int64, device=dev)
                f_out_batch, f_out_seg, f_out_l0 = torch.empty(ec, **i32), torch.empty(G + 1, **i32), torch.empty(G, **i32)
                counts = torch.empty(2, **i32)
                mrl = self.max_row_len.get(et, 0)
                ws_ptr, ws_bytes = self._workspace(max(nc, fc), ec, mrl)
                c32 = self.col32[et]
                p = _PygHop(g.row_ptr.data_ptr(), (g.col if c32 is None else c32).data_ptr(), self.wm_dtype, G, m,
                            rs[h * len(self.etypes) + ti].data_ptr(),
                            st["nodes"].data_ptr(), st["batch"].data_ptr(), st["seg"].data_ptr(), nc,
                            f_ids.data_ptr(), f_batch.data_ptr(), f_seg.data_ptr(), f_l0.data_ptr(), fc,
                            offsets.data_ptr(), row_l.data_ptr(), col_l.data_ptr(), gid.data_ptr(), ec,
                            nodes_out.data_ptr(), nodes_out_batch.data_ptr(), nodes_out_seg.data_ptr(),
                            f_out.data_ptr(), f_out_batch.data_ptr(), f_out_seg.data_ptr(), f_out_l0.data_ptr(),
                            counts.data_ptr(), scratch_r.data_ptr(), scratch_c.data_ptr(), ws_ptr, ws_bytes,
                            g.weight.data_ptr() if self.biased else None,
                            torch_dtype_to_wm(g.weight.dtype) if self.biased else 0, mrl,
                            int(self.num_nodes.get(src_t, 0)),   # the renumbered ids are type-local ids of the SOURCE type
                            self.flags | (0 if c32 is None else HOP_COL_INT32))
                L.check(lib.wgamd_sample_hop_pyg_nosync(_ct.byref(p), get_stream()), "wgamd_sample_hop_pyg_nosync")
                keep += [scratch_r, scratch_c, f_out, f_out_batch, f_out_seg, f_out_l0, counts, f_ids, f_batch, f_l0,
                         st["nodes"], st["batch"], st["seg"]]
                # (f_batch / f_local0 / frontier_cap: what a call-group consumer needs to place the hop's rows in the node list
                #  of the destination type without going through per-batch views — bench_mag.py)
                # (counts = {sampled edges, vertices of the source type after the call}: device-resident; a call-group consumer
                #  reads them in its one size read-back instead of indexing `offsets` once more)
                rec["calls"].append(dict(et=et, offsets=offsets, row=row_l, col=col_l, gid=gid, f_seg=f_seg, f_batch=f_batch,
                                         f_local0=f_l0, frontier_cap=fc, hop=h, counts=counts))
                st["nodes"], st["batch"], st["seg"] = nodes_out, nodes_out_batch, nodes_out_seg
                st["cap"] = nc + ec
                st["gained_cap"] += ec
            rec["sizes"].append({t: (state[t]["seg"][1:] - state[t]["seg"][:-1]) for t in self.ntypes})
        rec["state"], rec["keep"] = state, keep
        return rec

    def finalize_batches(self, rec):
        """Per mini-batch the tuple of ``hetero_neighbor_sample``: (node{type}, row{etype}, col{etype}, edge{etype},
        num_sampled_nodes{type}, num_sampled_edges{etype})."""
        G, dev = self.G, self.dev
        state = rec["state"]
        # every small size vector of the call group goes to the host in ONE copy (each .cpu() is a stream sync)
        pieces = [state[t]["seg"] for t in self.ntypes]
        pieces += [s_[t] for s_ in rec["sizes"] for t in self.ntypes]
        live_calls = [c for c in rec["calls"] if c is not None]
        pieces += [c["offsets"][c["f_seg"].long()] for c in live_calls]
        flat = torch.cat([p_.reshape(-1).to(torch.int64) for p_ in pieces]).cpu().tolist()
        host, at = [], 0
        for p_ in pieces:
            host.append(flat[at:at + p_.numel()])
            at += p_.numel()
        it = iter(host)
        nseg = {t: next(it) for t in self.ntypes}
        sizes = [{t: next(it) for t in self.ntypes} for _ in rec["sizes"]]
        calls = [None if c is None else dict(c, eseg=next(it)) for c in rec["calls"]]
        empty = torch.zeros(0, dtype=torch.int64, device=dev)
        n_et = len(self.etypes)
        # per edge type: dtype conversion, edge-id lookup and the hop concatenation ONCE for the call group
        merged, offs, num_edges_b = {}, {}, {}
        for ti, et in enumerate(self.etypes):
            fields, segs = [], []
            for h in range(self.hops):
                c = calls[h * n_et + ti]
                if c is None:
                    continue
                hi = c["eseg"][G]
                fields.append([c["row"][:hi].long(), c["col"][:hi].long(), self.graphs[et].edge_id[c["gid"][:hi]]])
                segs.append(c["eseg"])
            m, offs[et] = _merge_hops_batch_major(fields, segs, G, dev)
            merged[et] = m if m else [empty, empty, empty]
            num_edges_b[et] = [[(calls[h * n_et + ti]["eseg"][b + 1] - calls[h * n_et + ti]["eseg"][b])
                                if calls[h * n_et + ti] is not None else 0 for h in range(self.hops)] for b in range(G)]
        # all per-batch views of a field in one torch.split call (a Python slice per batch, type and field costs more
        # than the kernels of the whole walk)
        def views(t, seg):
            return torch.split(t[seg[0]:seg[G]], [seg[b + 1] - seg[b] for b in range(G)])
        node_v = {t: (views(state[t]["nodes"], nseg[t]) if state[t]["cap"] > 0 else (empty,) * G) for t in self.ntypes}
        edge_v = {et: [views(merged[et][i], offs[et]) for i in range(3)] for et in self.etypes}
        # the call group as a whole (see PygWalkResult.finalize_batches): per type the concatenated ids + per-batch sizes
        rec["group_context"] = dict(
            nodes={t: ((state[t]["nodes"][nseg[t][0]:nseg[t][G]] if state[t]["cap"] > 0 else empty),
                       [nseg[t][b + 1] - nseg[t][b] for b in range(G)] if state[t]["cap"] > 0 else [0] * G)
                   for t in self.ntypes},
            edges={et: (merged[et][2][:offs[et][G]], [offs[et][b + 1] - offs[et][b] for b in range(G)]) for et in self.etypes})
        out = []
        for b in range(G):
            num_nodes = {t: [sizes[0][t][b]] + [sizes[h + 1][t][b] - sizes[h][t][b] for h in range(self.hops)]
                         for t in self.ntypes}
            out.append(({t: node_v[t][b] for t in self.ntypes}, {et: edge_v[et][0][b] for et in self.etypes},
                        {et: edge_v[et][1][b] for et in self.etypes}, {et: edge_v[et][2][b] for et in self.etypes},
                        num_nodes, {et: num_edges_b[et][b] for et in self.etypes}))
        return out
