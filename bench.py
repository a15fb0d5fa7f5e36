#!/usr/bin/env python
"""bench.py — sampled-edges/s of the mini-batch hot path on the ogbn-products-like workload.

One "step" = `batches_per_step` (default 8 x 191) mini-batches of 1024 seeds through the whole hot path with everything
resident in HBM, processed as `--groups-per-step` (8) CALL GROUPS of `--call-group` mini-batches (default: the loaders' own
memory-sized call group, 191 on a 288 GB MI355X) — the launch shape is FIXED and does not depend on --steps:   2-hop uniform fan-out walk [25,10] (sample + renumber, no host sync)  ->
feature gather (fp32, F=100: every DISTINCT row of the call group's node list once — the list is de-duplicated on the walk
stream right behind the walk, layer 1 reads the gathered rows through the inverse index; `--fetch rows` = x = feat[n_id] row for
row, the form rounds 1-5 measured, reported as variants.materialised_full)  ->  2-layer GraphSAGE forward (mean aggregation +
lin_l/lin_r in HIP).
`value` = sampled edges of all ranks / max-over-ranks wall time of exactly K steps (the driver's `--steps 20 --warmup 5`
= 160 timed call groups after 40 untimed ones, a steady-state software-pipelined region of ~0.7 s).

N > 1 (one process per GPU, seeds sharded, CSR replicated): the HEADLINE is the north-star multi-GPU path — the feature table
range-partitioned over the ranks (per = ceil(V/W)) and fetched by the RCCL all-to-all-v pipeline of wholememory_gather
(csrc/wg_comm.hip; reference gather_op_impl_nccl.cu:23-171).  The peer-mapped fetch (HIP IPC loads over xGMI) and the
collective-free replicated table are measured in the same run and reported under `placements`.  Before anything is timed a
pre-flight (`selftest`) sends a known-answer table through the same exchange on every rank and compares bit for bit.

Contract: python bench.py --gpus N --steps K --warmup W   (N>1: launched by torch.distributed.run)
Prints ONE JSON line on rank 0 with `roofline` (dominant HIP kernel, measured live with HIP
events on the launch stream) and `cpu_baseline` (the C oracle, OpenMP over host cores, bounded sample).
Synthetic data: RMAT (a=.57,b=.19,c=.19) symmetrised + dedup, V=2,449,029, ~123.7 M directed
edges; features U(-1,1) fp32 [V,100]; random-init weights.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "cugraph-gnn_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # cpu_baseline: no spinning OpenMP workers (read at libgomp load)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

V_PRODUCTS = 2_449_029
E_UNDIRECTED = 61_859_140
FEAT_DIM = 100
# name -> (V, undirected RMAT edges, feature dim, classes, fan-out)   [SURVEY.md §8 dataset sizes / BASELINE configs]
WORKLOADS = {"products": (V_PRODUCTS, E_UNDIRECTED, 100, 47, [25, 10]),
             "papers100m": (111_059_956, 807_842_936, 128, 172, [25, 10]),
             "rmat26": (1 << 26, 1 << 29, 256, 64, [15, 10, 5])}
HIDDEN = 256
SPMM1, SPMM2 = "spmm1(mean)+self", "spmm2(mean)+self"   # stage labels (layer-1 F = feature dim, layer-j F = HIDDEN)


def spmm_label(j):
    return "spmm%d(mean)+self" % (j + 1)

CLASSES = 47
BATCH = 1024
FANOUT = [25, 10]
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (≈6.3 TB/s achievable)
MFMA_F32_PEAK_TFPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA (v_mfma_f32_16x16x4_f32), dense
MFMA_BF16_PEAK_TFPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA, dense


def rmat_csr(n_nodes, n_undirected, seed, device, a=0.57, b=0.19, c=0.19):
    """RMAT edges at scale ceil(log2 V), folded into [0,V), randomly relabelled, symmetrised,
    deduplicated, returned as CSR (row_ptr int64, col int64) on `device`."""
    g = torch.Generator(device=device).manual_seed(seed)
    scale = int(np.ceil(np.log2(n_nodes)))
    src = torch.zeros(n_undirected, dtype=torch.int64, device=device)
    dst = torch.zeros(n_undirected, dtype=torch.int64, device=device)
    for _ in range(scale):
        u = torch.rand(n_undirected, generator=g, device=device)
        sbit = (u >= a + b).to(torch.int64)
        dbit = (((u >= a) & (u < a + b)) | (u >= a + b + c)).to(torch.int64)
        src = (src << 1) | sbit
        dst = (dst << 1) | dbit
    perm = torch.randperm(1 << scale, generator=g, device=device)
    src, dst = perm[src] % n_nodes, perm[dst] % n_nodes
    del perm
    keep = src != dst
    src, dst = src[keep], dst[keep]
    del keep
    keys = torch.cat([src * n_nodes + dst, dst * n_nodes + src])
    del src, dst
    keys = torch.unique(keys)  # sorted => CSR order, duplicates dropped
    rows = torch.div(keys, n_nodes, rounding_mode="floor")
    col = keys - rows * n_nodes
    del keys
    deg = torch.bincount(rows, minlength=n_nodes)
    del rows
    row_ptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=device)
    row_ptr[1:] = torch.cumsum(deg, 0)
    return row_ptr, col


class SagePipeline:
    """The measured hot path (one instance per rank).  Mini-batches are processed in CALL GROUPS of G
    (cugraph_pyg's local_seeds_per_call idea): one launch sequence samples + renumbers G mini-batches
    (each with its own seed, each renumbered on its own), then ONE gather / SpMM / GEMM runs over the
    block-diagonal concatenation.  Groups are software-pipelined: the walk of group g+1 is enqueued
    before the host reads the (tiny, pinned) size vector of group g, so the GPU queue never drains."""

    def __init__(self, row_ptr, col, feat_table, device, G, overlap_walk=True, walk_priority=0, walk_stream=None, dedup=True):
        from wholegraph_amd import fused, nn
        self.nn = nn
        self.device = device
        self.G = G
        self.distinct_rows = []
        # dedup (round 6, the headline's feature fetch at every N): the call group's node list names a table row once per
        # mini-batch that sampled it (10.9 M rows per products group over 1.19 M distinct ones).  The list is de-duplicated on
        # the walk stream right behind the walk (wgamd_unique_bounded_live: mark / scan over V / compact / look up, no host
        # sync — the count travels with the walk's sizes), every DISTINCT row is gathered once (locally, or through the RCCL
        # exchange when the table is partitioned) and layer 1 reads the gathered rows through the inverse index (its src_ids).
        # dedup=False: x = feat[n_id] row for row, as rounds 1-5 measured it (variants.materialised_full).
        self.dedup = bool(dedup)
        self.mode = "fused"            # the mode of the pass being enqueued (run_groups / probe_stages set it)
        self.n_vertices = int(row_ptr.shape[0]) - 1
        self.walk = fused.NoSyncWalk(row_ptr, col, BATCH, FANOUT, col.dtype, G, pad_unique=False)   # ids take the CSR's column dtype
        self.feat = feat_table  # WholeMemoryTensor
        g = torch.Generator(device=device).manual_seed(1)
        L = len(FANOUT)
        dims = [FEAT_DIM] + [HIDDEN] * (L - 1) + [CLASSES]
        self.dims = dims
        self.convs = [nn.SAGEConv(dims[j], dims[j + 1]).to(device) for j in range(L)]
        for c in self.convs:
            for p in c.parameters():
                p.data = (torch.rand(p.shape, generator=g, device=device) - 0.5) * 0.1
                p.requires_grad_(False)
        # [W_l | W_r]^T so that lin_l(agg) + lin_r(x_self) is one GEMM over the [agg | x_self] rows
        self.w_t = [torch.cat([c.lin_l.weight, c.lin_r.weight], dim=1).t().contiguous() for c in self.convs]
        self.bias = [c.lin_l.bias for c in self.convs]
        self.fused_relu = hasattr(torch, "_addmm_activation")
        # the walk's kernels are short and feed the NEXT group: on a high-priority stream they take the wave slots that free
        # up between the long streaming kernels of the forward pass instead of queueing behind them
        self.walk_stream = walk_stream if walk_stream is not None else (
            torch.cuda.Stream(device=device, priority=walk_priority) if overlap_walk else None)
        self.host_wait_s = 0.0
        self.rs_base = None
        self.distributed = self.feat.is_distributed
        path = self.feat.fetch_path() if hasattr(self.feat, "fetch_path") else "all-to-all"
        self.fetch_tag = "peer-mapped" if "peer-mapped" in path else "all-to-all"
        self._bufs = {}
        # Schedule switch (WGAMD_BENCH_GATHER_AHEAD=1, off by default): the feature gather of group g+1 is enqueued on the WALK
        # stream right behind the walk of g+1, so that it runs next to the layers of group g on the main stream instead of in
        # front of the layers of g+1 (two x buffers; same kernels, same results).  Measured (one box, two runs each): 3.85 /
        # 3.87 G edges/s without, 3.72 / 3.71 with — two HBM-bound kernels side by side take longer than one after the other.
        self.gather_ahead = (os.environ.get("WGAMD_BENCH_GATHER_AHEAD", "0") == "1" and self.walk_stream is not None
                             and not self.distributed)
        self._x_flip = 0

    def prefetch(self, pending):
        """gather_ahead: x = feat[n_id] of an already sampled group, on the walk stream (the host waits for the group's sizes
        first — the GPU has the layers of the previous group queued meanwhile)."""
        from wholegraph_amd.tensor import local_gather
        res, sizes_h, ev = pending
        t_wait = time.perf_counter()
        ev.synchronize()
        self.host_wait_s += time.perf_counter() - t_wait
        sz = sizes_h.tolist()
        u_last = sz[len(sz) - 1][1]
        name = "x%d" % self._x_flip
        self._x_flip ^= 1
        buf = self.rows_buffer(name, u_last, FEAT_DIM)
        self._bufs[name].record_stream(self.walk_stream)
        with torch.cuda.stream(self.walk_stream):
            x = local_gather(self.feat.local_tensor, res.unique[len(sz) - 1][:u_last], buf)
            evx = torch.cuda.Event()
            evx.record(self.walk_stream)
        res.x_prefetched = (x, evx)

    def rows_buffer(self, name, n_rows, n_cols):
        """[n_rows, n_cols] view of a per-purpose buffer that only ever grows (by 12 % steps): the sizes of a call group
        differ by a per cent or so from group to group, and a fresh multi-GB block from the caching allocator (hipMalloc)
        in the middle of the timed region costs tens of milliseconds.  Safe to reuse: every use is on the main stream."""
        buf = self._bufs.get(name)
        if buf is None or buf.shape[0] < n_rows or buf.shape[1] != n_cols:
            buf = torch.empty((int(n_rows * 1.12) + 1024, n_cols), dtype=torch.float32, device=self.device)
            self._bufs[name] = buf
        return buf[:n_rows]

    def dense(self, a, w_t, bias, relu):
        if relu and self.fused_relu:
            try:
                return torch._addmm_activation(bias, a, w_t, use_gelu=False)   # bias + ReLU in the GEMM epilogue
            except RuntimeError:
                self.fused_relu = False
        out = torch.addmm(bias, a, w_t)
        return out.relu_() if relu else out

    def sample(self, seeds, group_id):
        """Enqueue the walk of one call group + the async D2H of its sizes."""
        hops = len(FANOUT)
        if self.rs_base is None:    # seed of (hop k, batch b) = 62 + hops*global_batch + k
            self.rs_base = (torch.arange(self.G, device=self.device, dtype=torch.int64).view(1, -1) * hops
                            + torch.arange(hops, device=self.device, dtype=torch.int64).view(-1, 1) + 62)
        rs = self.rs_base + group_id * self.G * hops    # one launch per call group

        def distinct_rows(res):
            if not self.dedup or self.mode.endswith("_fetch"):     # (the fetch-in-the-layer variants read the table through n_id)
                return
            from wholegraph_amd.tensor import unique_bounded_nosync
            distinct, inverse, info = unique_bounded_nosync(res.unique[hops - 1], res.counts[hops - 1][1:2], self.n_vertices)
            info_h = torch.empty(2, dtype=torch.int32, pin_memory=True)
            info_h.copy_(info, non_blocking=True)
            res.fetch = (distinct, inverse, info_h)
        if self.walk_stream is None:
            res = self.walk.run(seeds, rs)
            distinct_rows(res)
            sizes_h = torch.empty((hops, 2), dtype=torch.int32, pin_memory=True)
            sizes_h.copy_(res.counts, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            return res, sizes_h, ev
        # The walk (integer, random-access bound) of group g+1 runs on its own HIP stream next to the feature
        # fetch / aggregation / GEMM of group g: different bottlenecks (random sectors + atomics vs streaming HBM +
        # MFMA), so they overlap instead of queueing behind each other.
        self.walk_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.walk_stream):
            res = self.walk.run(seeds, rs)
            distinct_rows(res)
            sizes_h = torch.empty((hops, 2), dtype=torch.int32, pin_memory=True)
            sizes_h.copy_(res.counts, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.walk_stream)
        return res, sizes_h, ev

    def forward(self, res, sizes_h, ev, timers=None, mode="split"):
        """Feature fetch + L-layer SAGE forward of one call group with exact (host-known) sizes."""
        nn = self.nn
        t_wait = time.perf_counter()
        ev.synchronize()
        self.host_wait_s += time.perf_counter() - t_wait     # host blocked on the walk of this group (--host-profile)
        sz = sizes_h.tolist()                   # per hop (seed hop first): [edges, unique nodes after the hop]
        L = len(sz)
        n_edges, n_uniq = [v[0] for v in sz], [v[1] for v in sz]
        if self.walk_stream is not None:
            # the walk's outputs were allocated on the walk stream and are consumed here on the main one
            main = torch.cuda.current_stream()
            for lst in (res.unique, res.unique_seg, res.target_seg, res.target_batch, res.offsets, res.neighbor_row,
                        res.center_row):
                for t in lst:
                    t.record_stream(main)
            for t in (getattr(res, "fetch", None) or ())[:2]:
                t.record_stream(main)
        t0 = self.G * BATCH

        def stage(name, fn):
            if timers is None:
                return fn()
            # a short device-side spin first: the host prepares and enqueues the launch while it runs, so the
            # start event is reached with the kernel already queued behind it and the event pair brackets the
            # kernel itself, not the host's launch latency (rocprofv3's per-kernel average is the cross-check)
            torch.cuda._sleep(300_000)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = fn()
            e.record()
            timers.append((name, s, e))
            return out

        # mode: "split" = gather | aggregate kernel | library GEMM;  "fused" = gather | ONE kernel per layer (aggregate in
        # LDS + fp32-MFMA transform) where the shape allows;  "split_fetch" / "fused_fetch" = the same two with the feature
        # fetch folded into layer 1 (x = feat[n_id] never materialised)
        fused_fetch = mode.endswith("_fetch")
        fused_layer = mode.startswith("fused")
        u_last = n_uniq[L - 1]
        n_id = res.unique[L - 1][:u_last]
        lazy = None
        ids1 = None          # layer 1 reads its input THROUGH this index (dedup: the inverse of the distinct-row list)
        if self.dedup and not fused_fetch and getattr(res, "fetch", None) is not None:
            distinct, inverse, info_h = res.fetch
            n_d, bad = (int(v) for v in info_h.tolist())
            assert not bad, "a sampled vertex id is not a row of the feature table"
            if timers is not None:
                self.distinct_rows.append(n_d)
            ids1 = inverse[:u_last]
            if self.distributed:      # every distinct row through the exchange ONCE; nothing is expanded afterwards
                x = stage("gather(" + self.fetch_tag + ")", lambda: self.feat.gather(distinct[:n_d], dedup=False))
            else:
                from wholegraph_amd.tensor import local_gather
                x = stage("gather", lambda: local_gather(self.feat.local_tensor, distinct[:n_d],
                                                         self.rows_buffer("x", n_d, FEAT_DIM)))
        elif fused_fetch and self.distributed:
            # a PEER-MAPPED partitioned table: layer 1 reads every rank's partition itself through byte offsets over this
            # process's mapping (wgamd_mapped_row_offsets): remote rows cross xGMI inside the layer kernel, no gathered copy
            lazy = stage("row_offsets(peer-mapped)", lambda: nn.mapped_lazy_rows(self.feat, n_id))
            x = None
        elif fused_fetch:
            x = None          # never materialised: layer 1 reads the feature table through n_id
        elif self.distributed:
            if timers is not None:   # stage probe only (untimed here): what the de-duplicated fetch puts on the wire
                self.distinct_rows.append(int(torch.unique(n_id).numel()))
            x = stage("gather(" + self.fetch_tag + ")", lambda: self.feat.gather(n_id))      # (de-duplicates on the wire, expands)
        elif timers is None and getattr(res, "x_prefetched", None) is not None:
            x, evx = res.x_prefetched     # gathered on the walk stream while the previous group's layers ran here
            torch.cuda.current_stream().wait_event(evx)
        else:
            from wholegraph_amd.tensor import local_gather
            x = stage("gather", lambda: local_gather(self.feat.local_tensor, n_id,
                                                     self.rows_buffer("x", u_last, FEAT_DIM)))
        # layer j consumes hop k = L-1-j (deepest first): one kernel builds [mean_j x_j | x_i] for the hop's targets,
        # one GEMM applies [W_l | W_r] with bias (+ ReLU between layers)
        h = x
        for j in range(L):
            k = L - 1 - j
            if k >= 1:
                n_dst = n_uniq[k - 1]
                rows = res.target_rows_in_unique(k, n_dst)   # "x[:num_dst]" of the block-diagonal layout
            else:
                n_dst = t0   # the seeds = the first BATCH rows of every batch's hop-0 unique list
                rows = res.target_rows_in_unique(0, n_dst)
            ptr, nbr = res.offsets[k][:n_dst + 1], res.neighbor_row[k][:n_edges[k]]
            if fused_layer and nn.sage_layer_fused_preferred(self.dims[j], self.dims[j + 1]):
                fetch = j == 0 and fused_fetch
                table = (lazy.table if lazy is not None else self.feat.local_tensor) if fetch else h
                through = (lazy.ids if lazy is not None else n_id) if fetch else (ids1 if j == 0 else None)
                h = stage(("fetch+" if fetch else "") + "sage_layer%d(fused)" % (j + 1), lambda: nn.sage_layer_fused_forward(
                    ptr, nbr, table, rows, self.w_t[j], self.bias[j], relu=j < L - 1,
                    mean=True, src_ids=through,
                    # (row stride = the width the kernel runs at: a 47-class head is computed as 64 zero-padded columns)
                    out=self.rows_buffer("h%d" % j, n_dst, -(-self.dims[j + 1] // 64) * 64)[:, :self.dims[j + 1]]))
                continue
            if j == 0 and fused_fetch:
                cat = stage("fetch+" + spmm_label(0), lambda: nn.sage_aggregate_fetch_forward(
                    ptr, nbr, self.feat.local_tensor, n_id, rows, True))
            elif j == 0 and ids1 is not None:
                cat = stage(spmm_label(0), lambda: nn.sage_aggregate_fetch_forward(ptr, nbr, h, ids1, rows, True))
            else:
                cat = stage(spmm_label(j), lambda: nn.sage_aggregate_forward(ptr, nbr, h, rows, True))
            h = stage("dense%d" % (j + 1), lambda: self.dense(cat, self.w_t[j], self.bias[j], relu=j < L - 1))
        return h, tuple(v for pair in sz for v in pair)


def loader_api_variants(row_ptr, col, table, convs, seeds, n_groups, G, which=("loader_api", "train_step", "train_step_per_batch", "forward_per_batch", "loader_api_materialised", "loader_api_per_batch", "gat")):
    """The same workload through the DROP-IN API: GraphStore + FeatureStore -> cugraph_pyg_amd NeighborLoader ->
    wholegraph_amd.nn.SAGEConv x L forward (the surface of python/cugraph-pyg/cugraph_pyg/loader/node_loader.py:16-178 and
    sampler/sampler.py:51-165).  `loader_api`: the epoch iterated in call groups (loader.call_groups(): one block-diagonal
    graph per `local_seeds_per_call` seeds, lazy x — the table is read through n_id inside the first layer's kernel);
    `loader_api_materialised`: the same with x = feat[n_id] gathered; `loader_api_per_batch`: the classic loop, one Data per
    1024 seeds, x / edge_index into SAGEConv (every layer over all sampled edges, as plain PyG code does).  PyG sampling
    semantics (a hop expands only the vertices the previous hop discovered), so a mini-batch has somewhat fewer edges than the
    WholeGraph-style walk of the headline; values are its own sampled edges per second."""
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    dev, V = row_ptr.device, int(row_ptr.shape[0]) - 1
    L = len(FANOUT)
    gs, fs = GraphStore(), FeatureStore()
    dst = torch.repeat_interleave(torch.arange(V, device=dev), row_ptr[1:] - row_ptr[:-1])
    gs[("n", "e", "n"), "coo", False, (V, V)] = torch.stack([col.to(torch.int64), dst])   # PyG: row 0 = source (the neighbour)
    del dst
    fs["n", "x", None] = table
    out = {}

    def group_pass(lazy, n_warm=5):
        loader = NeighborLoader((fs, gs), FANOUT, input_nodes=seeds[:(n_groups + n_warm) * G * BATCH], batch_size=BATCH,
                                shuffle=False, random_state=62)
        edges, t0, n = 0, None, 0
        with torch.no_grad():
            for grp in loader.call_groups():
                if n == n_warm:
                    torch.cuda.synchronize()
                    t0, edges = time.perf_counter(), 0
                h = grp.x if lazy else grp.node_attr("x", lazy=False)
                for j, c in enumerate(convs):
                    h = c(h, grp.layer_graph(j), act="relu" if j < L - 1 else None)
                edges += grp.num_edges
                n += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return edges / dt, dt / max(n - n_warm, 1) * 1e3, edges / max(n - n_warm, 1) / G

    if "loader_api" in which:
        v, ms, epb = group_pass(True)
        out["loader_api"] = {"value": v, "ms_per_call_group": ms, "edges_per_batch": epb,
                             "note": "GraphStore + FeatureStore -> NeighborLoader.call_groups() (%d mini-batches per group, the loader's "
                                     "default) -> nn.SAGEConv x %d forward; x lazy (table read through n_id in the layer-1 kernel)" % (G, L)}

    def train_pass(n_warm=3):
        """The same loop TRAINING: forward (x lazy) -> cross-entropy on synthetic labels -> backward -> SGD step, every call
        group — what the reference's examples do per mini-batch (python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:
        119-125 + examples/*: loss.backward(); optimizer.step()).  Forward and backward run on the one-kernel layer
        (wholegraph_amd.nn._SageLayer: wgamd_sage_layer_fused_bf16x3_train / wgamd_sage_wgrad_bf16x3 / the layer kernel over the
        transposed hop for the hidden state's gradient)."""
        from wholegraph_amd import nn as wnn
        model = torch.nn.ModuleList([wnn.SAGEConv(c.in_channels[0], c.out_channels) for c in convs]).to(dev)
        with torch.no_grad():
            for m, c in zip(model, convs):
                for pm, pc in zip(m.parameters(), c.parameters()):
                    pm.copy_(pc)
        opt = torch.optim.SGD(model.parameters(), lr=0.01)
        labels = torch.randint(0, CLASSES, (V,), generator=torch.Generator(device=dev).manual_seed(5), device=dev)
        loader = NeighborLoader((fs, gs), FANOUT, input_nodes=seeds[:(n_groups + n_warm) * G * BATCH], batch_size=BATCH,
                                shuffle=False, random_state=62)
        edges, t0, n, losses, wgrad_calls = 0, None, 0, [], []
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        fwd_ms = bwd_ms = 0.0
        for grp in loader.call_groups():
            if n == n_warm:
                torch.cuda.synchronize()
                t0, edges = time.perf_counter(), 0
            probe = n == n_warm + n_groups - 1      # the last group: forward / backward+step split by events
            if probe:
                # ... and every weight-gradient launch of its backward pass between its own HIP events (the roofline of the
                # backward's dominant kernel: algorithmic bytes = rows x (2F x 4 + 8 + N x 4 x (2 with the ReLU mask)))
                real_wgrad = wnn.sage_wgrad

                def timed_wgrad(agg, x, self_rows, grad_out, *a, **kw):
                    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s_.record()
                    real_wgrad(agg, x, self_rows, grad_out, *a, **kw)
                    e_.record()
                    wgrad_calls.append((int(agg.shape[0]), int(agg.shape[1]), int(grad_out.shape[1]),
                                        (len(a) > 3 and a[3] is not None) or kw.get("act_out") is not None, s_, e_))
                wnn.sage_wgrad = timed_wgrad
                ev[0].record()
            h = grp.x
            for j, c in enumerate(model):
                h = c(h, grp.layer_graph(j), act="relu" if j < L - 1 else None)
            loss = wnn.cross_entropy(h, labels[grp.batch])     # (wgamd_softmax_xent: one launch forward, one backward)
            if probe:
                ev[1].record()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            if probe:
                ev[2].record()
                wnn.sage_wgrad = real_wgrad
            if n in (0, n_warm + n_groups - 1):
                losses.append(loss.detach())
            edges += grp.num_edges
            n += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        fwd_ms, bwd_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
        roof = None
        if wgrad_calls:
            rows, F_, N_, masked, s_, e_ = max(wgrad_calls, key=lambda c: c[0] * c[2])
            ms_ = s_.elapsed_time(e_)
            by = rows * (2 * F_ * 4 + 8 + N_ * 4 * (2 if masked else 1))
            roof = {"bound": "hbm", "kernel": "sage_wgrad_kernel", "rows": rows, "F": F_, "N": N_, "relu_mask": masked,
                    "avg_launch_ms": round(ms_, 4), "algorithmic_bytes_per_launch": int(by), "achieved": round(by / ms_ / 1e6, 1),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(by / ms_ / 1e6 / HBM_PEAK_GBPS, 4),
                    "mfma_TFps": round(6 * 2.0 * rows * 2 * F_ * N_ / ms_ / 1e9, 1),
                    "mfma_frac": round(6 * 2.0 * rows * 2 * F_ * N_ / ms_ / 1e9 / MFMA_BF16_PEAK_TFPS, 4),
                    "timing": "HIP events around the launch + its partial-sum reduction, largest hop of the last timed call group"}
            hit = load_pmc("sage_wgrad_kernel", want_void=False, workload="train")
            if hit:
                roof["traffic"] = hit["bytes"]
                roof["traffic_source"] = hit["source"] + " kernel " + hit["kernel"]
        return edges / dt, dt / max(n - n_warm, 1) * 1e3, fwd_ms, bwd_ms, [round(float(v), 4) for v in losses], roof

    try:
        if "train_step" not in which:
            raise KeyError
        v, ms, fwd_ms, bwd_ms, losses, wroof = train_pass()
        out["train_step"] = {"value": v, "ms_per_call_group": ms, "forward_loss_ms": round(fwd_ms, 3),
                             "backward_step_ms": round(bwd_ms, 3), "loss_first_last": losses, "wgrad_roofline": wroof,
                             "note": "NeighborLoader.call_groups() -> nn.SAGEConv x %d (x lazy) -> cross-entropy -> backward -> SGD step "
                                     "per call group of %d mini-batches; value = sampled edges per second of the whole training "
                                     "loop (walk overlapped on its own stream as in loader_api)" % (L, G)}
    except KeyError:
        pass
    except Exception as exc:   # noqa: BLE001
        out["train_step"] = {"value": None, "error": repr(exc)[:300]}
    def per_batch_pass(train, n_warm=1, n_timed=4):
        """The reference's optimizer semantics: SAMPLE per call group, STEP per mini-batch of BATCH seeds
        (pylibwholegraph/torch/gnn_model.py:119-125: loss.backward(); optimizer.step() inside the batch loop).
        cugraph_pyg_amd.loader.PerBatchStep: the group's walk and lazy x are made once, every mini-batch is staged into
        fixed-size buffers by one launch and the whole step (trimmed forward over the mini-batch's own slice of the layer
        graphs, cross-entropy, backward, SGD) is one HIP-graph replay.  `train=False`: the forward alone, per mini-batch."""
        from wholegraph_amd import nn as wnn
        from cugraph_pyg_amd.loader import PerBatchStep
        model = torch.nn.ModuleList([wnn.SAGEConv(c.in_channels[0], c.out_channels) for c in convs]).to(dev)
        with torch.no_grad():
            for m, c in zip(model, convs):
                for pm, pc in zip(m.parameters(), c.parameters()):
                    pm.copy_(pc)
        opt = torch.optim.SGD(model.parameters(), lr=0.01)
        labels = torch.randint(0, CLASSES, (V,), generator=torch.Generator(device=dev).manual_seed(5), device=dev)

        def step(batch):
            if train:
                opt.zero_grad(set_to_none=True)
            h = batch.x
            for j, c in enumerate(model):
                h = c(h, batch.layer_graph(j), act="relu" if j < L - 1 else None)
            if not train:
                return h[:batch.batch_size].sum()
            # mean loss over the live seeds of the row_cap[0] output rows: no slice (its backward is a zero-fill + a copy per step)
            loss = wnn.cross_entropy(h, labels.index_select(0, batch.n_id[:h.shape[0]]), batch.seed_mask)
            loss.backward()
            opt.step()
            return loss
        loader = NeighborLoader((fs, gs), FANOUT, input_nodes=seeds[:(n_timed + n_warm) * G * BATCH], batch_size=BATCH,
                                shuffle=False, random_state=62)
        stepper = PerBatchStep(step, table=table, optimizer=opt if train else None)
        edges, t0, n, steps, first, last = 0, None, 0, 0, None, None
        with torch.set_grad_enabled(train):
            for grp in loader.call_groups():
                if n == n_warm:
                    torch.cuda.synchronize()
                    t0, edges, steps = time.perf_counter(), 0, 0
                for b in range(grp.n_batches):
                    last = stepper(grp, b)
                    if first is None:
                        first = float(last.detach())
                edges += grp.num_edges
                steps += grp.n_batches
                n += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"value": edges / dt, "ms_per_call_group": dt / max(n - n_warm, 1) * 1e3, "ms_per_mini_batch": dt / max(steps, 1) * 1e3,
                "optimizer_steps_timed": steps if train else 0, "graph_captures": stepper.captures,
                "loss_first_last": [round(first, 4), round(float(last.detach()), 4)] if train else None,
                "buffers": {"rows": stepper.batch.row_cap, "edges": stepper.batch.edge_cap, "nodes": stepper.batch.node_cap}}

    for name, train in (("train_step_per_batch", True), ("forward_per_batch", False)):
        if name not in which:
            continue
        try:
            out[name] = per_batch_pass(train)
            out[name]["note"] = (
                "REFERENCE OPTIMIZER SEMANTICS: NeighborLoader.call_groups() samples %d mini-batches per call, then ONE SGD step per "
                "mini-batch of %d seeds (%d steps per call group): wgamd_call_group_stage_batch + one HIP-graph replay of forward "
                "(trimmed, x lazy) -> cross-entropy -> backward -> SGD per mini-batch (cugraph_pyg_amd.loader.PerBatchStep)"
                % (G, BATCH, G) if train else
                "the forward alone per mini-batch of %d seeds through the same staged buffers + HIP graph (what a per-batch "
                "inference loop gets without consuming call groups whole)" % BATCH)
        except Exception as exc:   # noqa: BLE001
            out[name] = {"value": None, "error": repr(exc)[:400]}
    if "loader_api_materialised" in which:
        v, ms, epb = group_pass(False)
        out["loader_api_materialised"] = {"value": v, "ms_per_call_group": ms, "edges_per_batch": epb,
                                          "note": "the same loop with x = feat[n_id] gathered once per call group (wholememory_gather)"}
    if "gat" in which and table.shape[1] % 4 == 0 and table.shape[1] <= 256 and L == 2:
        # the GAT half of north_star on the SAME graph, features and loader: 2 x nn.GATConv (edge softmax + attention-weighted
        # sum, 4 heads x 64 then 1 x classes; self loops) over the call groups' trimmed layer graphs, aggregate-first, x lazy
        from wholegraph_amd import nn as wnn
        gat = torch.nn.ModuleList([wnn.GATConv(int(table.shape[1]), 64, heads=4), wnn.GATConv(256, CLASSES, heads=1)]).to(dev)
        labels = torch.randint(0, CLASSES, (V,), generator=torch.Generator(device=dev).manual_seed(6), device=dev)
        opt = torch.optim.SGD(gat.parameters(), lr=0.01)
        for train in (False, True):
            n_g, n_warm = min(n_groups, 6), 6
            ids = seeds[:(n_g + n_warm) * G * BATCH]
            loader = NeighborLoader((fs, gs), FANOUT, input_nodes=ids, batch_size=BATCH, shuffle=False, random_state=62)
            edges, t0, n, loss = 0, None, 0, None
            with torch.set_grad_enabled(train):
                for grp in loader.call_groups():
                    if n == n_warm:
                        torch.cuda.synchronize()
                        t0, edges = time.perf_counter(), 0
                    h = grp.x
                    for j, c in enumerate(gat):
                        h = c(h, grp.layer_graph(j), act="relu" if j == 0 else None)
                    if train:
                        loss = wnn.cross_entropy(h, labels[ids[n * G * BATCH:n * G * BATCH + h.shape[0]]])
                        opt.zero_grad(set_to_none=True)
                        loss.backward()
                        opt.step()
                    edges += grp.num_edges
                    n += 1
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out["gat_train_step" if train else "gat_loader_api"] = {
                "value": edges / dt, "ms_per_call_group": dt / max(n - n_warm, 1) * 1e3,
                "note": "NeighborLoader.call_groups() -> nn.GATConv(%d, 64, heads=4) -> ReLU -> nn.GATConv(256, %d, heads=1), self loops, "
                        "aggregate-first over the trimmed layer graphs, x lazy%s" % (
                            int(table.shape[1]), CLASSES, " -> cross-entropy -> backward -> SGD step per call group" if train else " (forward)")}
    if "loader_api_per_batch" not in which:
        return out
    n_b, n_warm = 96, 16
    loader = NeighborLoader((fs, gs), FANOUT, input_nodes=seeds[:(n_b + n_warm) * BATCH], batch_size=BATCH, shuffle=False,
                            random_state=62)
    edges, n, t0 = 0, 0, None
    with torch.no_grad():
        for batch in loader:
            if n == n_warm:
                torch.cuda.synchronize()
                t0, edges = time.perf_counter(), 0
            h = batch.x
            for j, c in enumerate(convs):
                h = c(h, batch.edge_index, act="relu" if j < L - 1 else None)
            _ = h[:batch.batch_size]
            edges += int(batch.edge_index.shape[1])
            n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["loader_api_per_batch"] = {"value": edges / dt, "ms_per_batch": dt / max(n - n_warm, 1) * 1e3,
                                   "note": "for batch in NeighborLoader: SAGEConv(batch.x, batch.edge_index) x %d, one Data per %d "
                                           "seeds (%d timed mini-batches): bound by per-batch host work, not by the device" % (L, BATCH, n - n_warm)}
    return out


def masked_stream(device, first_cu, n_cus):
    """A HIP stream whose kernels may only occupy CUs [first_cu, first_cu + n_cus) of the CU-mask enumeration
    (hipExtStreamCreateWithCUMask), wrapped for torch.  The runtime spreads the bits of the mask over the XCDs, so a
    contiguous bit range is an equal share of every XCD."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    words = (ctypes.c_uint32 * 16)()
    for cu in range(first_cu, first_cu + n_cus):
        words[cu // 32] |= 1 << (cu % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 16, words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    return torch.cuda.ExternalStream(st.value, device=device)


def usable_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup quota (a container that
    reports 128 cores but is throttled to a few makes spinning OpenMP threads stall for tens of ms)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(row_ptr_h, col_h, feat_h, seeds_h, weights, budget_s=15.0):
    L = len(FANOUT)
    """The C oracle (OpenMP over seeds) + torch-CPU dense layers on the same workload, bounded."""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # must be set before libgomp is loaded by the oracle
    import oracle
    oracle.build()
    threads = usable_cpus()
    oracle.set_num_threads(threads)
    torch.set_num_threads(max(1, threads))
    t0 = time.perf_counter()
    edges, batches = 0, 0
    for b in range(len(seeds_h)):
        tg, ei, rp, ci = oracle.multilayer_sample(row_ptr_h, col_h, seeds_h[b], FANOUT, [62 + L * b + k for k in range(L)])
        h = oracle.gather_rows(feat_h, tg[0])
        for j, (wl, bl, wr) in enumerate(weights):
            agg = oracle.spmm_csr(rp[j], ci[j], h, mean=True)
            out = torch.from_numpy(agg) @ wl.T + bl + torch.from_numpy(h[: len(tg[j + 1])]) @ wr.T
            h = (torch.relu(out) if j < L - 1 else out).numpy()
        edges += int(sum(c.size for c in ci))
        batches += 1
        if time.perf_counter() - t0 > budget_s and batches >= 3:
            break
    dt = time.perf_counter() - t0
    return {"value": edges / dt, "unit": "sampled-edges/s", "cores": threads, "kind": "port",
            "sample": f"{batches} mini-batches of {BATCH} seeds, fan-out {FANOUT}, same graph/features "
                      f"(C oracle with OpenMP + torch CPU linear), {dt:.1f} s"}


PROFILE_ROUNDS = ("r06", "r05", "r04", "r03", "r02", "r01")


def load_pmc(kernel_prefix, want_void=True, workload="products", prefer=None):
    """HBM bytes per launch of a kernel from the committed PMC passes (profiles/rNN/pmc_traffic.json — pmc_traffic_<workload>.json
    for the other BASELINE configurations —, newest round first: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of THIS
    command's launch shape, FETCH_SIZE x 2 per MI355X_MICROARCH.md §HBM; tools/pmc_summary.py).  Counters cannot be read from
    inside the timed process."""
    fname = "pmc_traffic.json" if workload == "products" else "pmc_traffic_%s.json" % workload
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", rnd, fname)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            pmc = json.load(f)
        hit = [(k, v) for k, v in pmc["kernels"].items() if k.startswith(kernel_prefix)
               and (not want_void or "<void" in k or "<long" not in k)]
        if prefer is not None:      # the instantiation whose id type (first template argument) is `prefer`, where the pass has it
            hit = [kv for kv in hit if kv[0].startswith(kernel_prefix + "<" + prefer + ",")] or hit
        per_shape = [kv for kv in hit if "#large" in kv[0]]
        if per_shape:     # one kernel, two launch shapes per call group: `#large` is the layer-1 launch
            hit = per_shape
        if hit:
            # several instantiations of one kernel (a 3-layer model: one per layer shape): the dominant stage is the one that
            # moves the most bytes
            k, v = max(hit, key=lambda kv: kv[1]["traffic_bytes"])
            return {"kernel": k, "bytes": v["traffic_bytes"], "launches": v["launches"], "max_bytes": v.get("max_traffic_bytes"),
                    "source": "profiles/%s/%s (%s)" % (rnd, fname, ", ".join(pmc["source"]))}
    return None


def load_profiled_avg(kernel, workload="products", prefer=None):
    """Average launch duration (ns) of the dominant kernel in the COMMITTED rocprofv3 --kernel-trace --stats summary of this
    command (profiles/rNN/rNN_kernel_stats.csv — <workload>_kernel_stats.csv for the other configurations —, newest round first).  A kernel that serves two layers appears as two template
    instantiations; the dominant stage is the longer one.  Lets the line carry `frac_profiled` next to the live HIP-event
    `frac`, so the two cannot drift apart unnoticed."""
    import csv
    for rnd in PROFILE_ROUNDS:
        base = os.path.join(ROOT, "profiles", rnd, (rnd if workload == "products" else workload) + "_kernel_stats")
        # the dominant-launch-shape statistics of the same trace where they exist (tools/trace_large_launches.py: the plain
        # average of a kernel mixes the call-group launches with the 1024-seed launches of the per-batch variant)
        path = base + "_large.csv" if os.path.exists(base + "_large.csv") else base + ".csv"
        if not os.path.exists(path):
            continue
        with open(path, newline="") as f:
            rows = [r for r in csv.DictReader(f) if kernel + "<" in r["Name"] or kernel + "(" in r["Name"]]
        plain = [r for r in rows if kernel + "<void" in r["Name"]]   # not the fetch-folded variant (ids type != void)
        pref = [r for r in rows if prefer is not None and kernel + "<" + prefer + "," in r["Name"].replace(" ", "")]
        rows = pref or plain or rows
        if rows:
            r = max(rows, key=lambda r: float(r["AverageNs"]))
            return {"avg_ns": float(r["AverageNs"]), "calls": int(r["Calls"]), "min_ns": float(r["MinNs"]),
                    "source": "profiles/%s/%s" % (rnd, os.path.basename(path))}
    return None


def relaunch_if_needed(args):
    """`--gpus N` is the number of ranks of this run.  Launched bare (`python bench.py --gpus N`, no WORLD_SIZE in the
    environment) with N > 1, this process re-executes the same command line under `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N` (the driver's own launch shape) and exits with the launcher's code; launched by a
    launcher whose WORLD_SIZE disagrees with `--gpus`, it fails loudly instead of printing a line with the wrong n_gpus."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        if args.gpus <= 1:
            return
        import socket
        import subprocess
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stdout.flush()
        sys.exit(subprocess.call(cmd, env=env))
    if int(env_world) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={env_world}; refusing to print a line "
                 f"whose n_gpus is not the number of ranks that ran (launch with --nproc-per-node {args.gpus}, or pass "
                 f"--gpus {env_world})")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="timed steps; one step = --groups-per-step call groups")
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps before the timed region")
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["mag"], default="products",
                    help="products = BASELINE configs[1] (the metric's config); papers100m = configs[2] scale on ONE GPU "
                         "(the north-star 10x-vs-CPU statement); rmat26 = configs[3]; mag = configs[4], the heterogeneous "
                         "2-hop walk + HeteroConv(GATConv) pipeline of bench_mag.py")
    ap.add_argument("--nodes", type=int, default=None)
    ap.add_argument("--edges", type=int, default=None, help="undirected RMAT edges before symmetrising")
    ap.add_argument("--call-group", type=int, default=0,
                    help="mini-batches per launch sequence (fixed launch shape); 0 = what the loaders use by default: the "
                         "memory-sized call group of cugraph_pyg_amd.sampler.default_local_seeds_per_call (2 %% of the device "
                         "memory: 191 mini-batches for fan-out [25, 10] with int64 ids on a 288 GB MI355X)")
    ap.add_argument("--groups-per-step", type=int, default=8, help="call groups per step (batches_per_step = G x this)")
    ap.add_argument("--feature-placement", choices=["auto", "replicated", "partitioned", "both"], default="auto",
                    help="N>1: 'partitioned' range-partitions the table and fetches remote rows over xGMI (RCCL all-to-all-v = "
                         "the HEADLINE, peer-mapped loads next to it); 'replicated' keeps the whole table on every GPU (no "
                         "data-path collective); auto/both = all of them for tables <= 36 GB (the replicated one is measured "
                         "first because it cannot hang, and reported under `placements`), partitioned only above")
    ap.add_argument("--id-dtype", choices=["auto", "int32", "int64"], default="auto",
                    help="dtype of csr_col / seeds / node ids: auto = int64 (the cugraph_pyg convention, data/graph_store.py:298-299 "
                         "— the interface north_star names); int32 (the WholeGraph test default, "
                         "cpp/tests/wholegraph_ops/wholegraph_csr_unweighted_sample_without_replacement_tests.cu:101) is timed as "
                         "the 'ids_int32' variant when V < 2^31")
    ap.add_argument("--partitioned-fetch", choices=["auto", "mapped", "alltoall"], default="auto",
                    help="how a partitioned table serves remote rows: alltoall = RCCL all-to-all-v exchange only; mapped = "
                         "peer-mapped partitions only (HIP IPC, loads over xGMI in one kernel; single node); auto = both "
                         "(headline: all-to-all)")
    ap.add_argument("--selftest", action="store_true",
                    help="run the known-answer pre-flight of the feature exchange even on a single-rank communicator (it always "
                         "runs for partitioned placements at N > 1)")
    ap.add_argument("--host-profile", action="store_true",
                    help="print (stderr) where the HOST spends a call group: enqueueing the walk, enqueueing the forward, blocked")
    ap.add_argument("--force-partitioned", action="store_true",
                    help="test aid: take the N>1 code path (RCCL all-to-all feature store) with a single rank")
    ap.add_argument("--no-overlap", action="store_true", help="run the walk on the main stream (no second HIP stream)")
    ap.add_argument("--walk-priority", type=int, default=0, help="HIP stream priority of the walk stream (-1 = high)")
    ap.add_argument("--walk-cus", type=int, default=0,
                    help="> 0: SPATIAL partition of the chip — the walk stream may only occupy this many CUs and the forward "
                         "pass (gather, SAGE layers) only the others (hipExtStreamCreateWithCUMask), so the walk's chain of "
                         "small dependent kernels is always resident instead of waiting for the persistent layer kernel")
    ap.add_argument("--layer-kernel", choices=["auto", "fused", "split"], default="auto",
                    help="auto/fused: every SAGE layer whose shape allows it runs as ONE kernel (neighbour rows -> LDS "
                         "operand tile -> MFMA); split: aggregation kernel + library GEMM per layer.  The other one is "
                         "timed as a variant")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for the "
                                                            "one-GPU rehearsal of the N>1 control flow in the tests)")
    ap.add_argument("--share-gpu", action="store_true", help="test aid: every rank uses cuda:0")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra timed passes of the other code paths")
    ap.add_argument("--fetch", choices=["distinct", "rows"], default="distinct",
                    help="feature fetch of the headline: 'distinct' (default) gathers every distinct row of a call group once and "
                         "lets layer 1 read through the inverse index; 'rows' gathers x = feat[n_id] row for row (rounds 1-5)")
    ap.add_argument("--extra-placement-timeout", type=int, default=180,
                    help="seconds the also-measured feature placement may take before the headline line is printed without it")
    ap.add_argument("--mag-rels", choices=["all", "r5"], default="all",
                    help="--workload mag: all 8 directed edge types (4 + reverses, ~42 M edges: BASELINE configs[4]) or the six "
                         "that rounds 3-5 ran (no rev_cites / rev_affiliated_with, 35.8 M edges) for round-to-round comparison")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    args = ap.parse_args()
    relaunch_if_needed(args)
    if args.workload == "mag":
        import bench_mag
        return bench_mag.main(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_partitioned:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or args.force_partitioned:
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world,
                                **({"device_id": device} if args.dist_backend == "nccl" else {}))

    from wholegraph_amd import WholeMemoryTensor, equal_entry_partition
    from wholegraph_amd import nn as nn_mod

    # ---- synthetic workload (replicated CSR, replicated and/or range-partitioned features) ----
    global FEAT_DIM, CLASSES, FANOUT
    wv, we, FEAT_DIM, CLASSES, FANOUT = WORKLOADS[args.workload]
    L = len(FANOUT)
    args.nodes = args.nodes or wv
    args.edges = args.edges or we
    row_ptr, col = rmat_csr(args.nodes, args.edges, seed=0, device=device)
    V, E = args.nodes, int(col.shape[0])
    id_dtype = torch.int32 if (args.id_dtype == "int32" and V < (1 << 31)) else torch.int64
    col_alt = None                      # the other id width, timed as a variant (single-GPU products runs only)
    if id_dtype == torch.int64 and V < (1 << 31) and not (args.no_variants or world > 1 or args.workload != "products"):
        col_alt = col.to(torch.int32)
    col = col.to(id_dtype)
    table_bytes = V * FEAT_DIM * 4
    # Placements of the feature table, in MEASUREMENT order (what cannot hang first) — the headline is picked afterwards
    # by HEAD_PREF: the range-partitioned table fetched by the RCCL all-to-all-v pipeline (north_star), then the peer-mapped
    # fetch, then the collective-free replicated table.
    #   "partitioned"        = WHOLEMEMORY_MT_DISTRIBUTED handle, bucketing + two all-to-all-v over RCCL
    #   "partitioned_mapped" = WHOLEMEMORY_MT_CHUNKED handle, peer partitions mapped through HIP IPC, one gather kernel
    #   "replicated"         = the whole table on every GPU
    HEAD_PREF = ["partitioned", "partitioned_mapped", "replicated"]
    if args.force_partitioned:
        placements = ["partitioned"]
    elif world == 1 or args.feature_placement == "replicated":
        placements = ["replicated"]
    else:
        part = {"auto": ["partitioned", "partitioned_mapped"], "alltoall": ["partitioned"],
                "mapped": ["partitioned_mapped"]}[args.partitioned_fetch]
        if args.dist_backend != "nccl":
            part = ["partitioned"]       # the torch.distributed rehearsal path has one exchange implementation
        if args.feature_placement == "partitioned" or (args.feature_placement == "auto" and table_bytes > (36 << 30)):
            placements = part
        else:
            placements = ["replicated"] + part
    wm_comm = None

    def library_comm():
        """ONE RCCL communicator of libwholegraph_amd for every table of this run (created on first use, collectively)."""
        nonlocal wm_comm
        if wm_comm is None:
            import wholegraph_amd as wg
            wm_comm = wg.create_group_communicator()
        return wm_comm

    def make_partitioned(placement, rows, dim, fill):
        """A [rows, dim] fp32 table range-partitioned over the ranks; `fill(local, first_row)` writes this rank's rows."""
        if args.dist_backend == "nccl":
            # the table is a handle of the library: bucketing, the id / row all-to-all-v (RCCL send/recv groups) or the
            # peer-mapped loads, and the row kernels all run inside wholememory_gather (csrc/wg_comm.hip)
            import wholegraph_amd as wg
            comm = library_comm()
            mtype = "chunked" if placement == "partitioned_mapped" else "distributed"
            if mtype == "chunked" and not (world > 1 and comm.support_type_location("chunked", "cuda")):
                raise RuntimeError("peer-mapped memory type not available on this communicator (ranks do not share a node)")
            t = wg.create_wholememory_tensor(comm, mtype, "cuda", [rows, dim], torch.float32, [dim, 1])
            local, first = t.get_local_tensor()
            fill(local, first)
            return t
        # torch.distributed pipeline (wholegraph_amd/dist.py); the gloo tests put two ranks on one GPU this way
        offs = equal_entry_partition(rows, world)
        local = torch.empty((offs[rank + 1] - offs[rank], dim), dtype=torch.float32, device=device)
        fill(local, offs[rank])
        return WholeMemoryTensor(local, global_rows=rows, partition_offsets=offs)

    def make_table(placement):
        gfeat = torch.Generator(device=device)
        if placement == "replicated":
            # single GPU, or a table that is small next to 288 GB of HBM3E: every GPU keeps the whole table and the
            # walk + fetch need no collective at all (seeds are the independent units)
            gfeat.manual_seed(100)
            return WholeMemoryTensor((torch.rand((V, FEAT_DIM), generator=gfeat, device=device) * 2 - 1))
        gfeat.manual_seed(100 + rank)
        return make_partitioned(placement, V, FEAT_DIM,
                                lambda local, first: local.copy_(torch.rand(tuple(local.shape), generator=gfeat, device=device) * 2 - 1))

    def selftest(placement):
        """Pre-flight of the exchange, before anything is timed: a KNOWN-ANSWER table (row i, column j holds
        (3 i + j) mod 2^20, exact in fp32 — the reference pytest's `table[i,j] = i + j` idea,
        tests/wholegraph_torch/ops/test_wholegraph_gather_scatter.py:12-27) partitioned exactly like the feature table goes
        through the SAME gather path on every rank with 100,001 random ids (+ negatives, + both ends of every partition) and
        must come back bit for bit; ids < 0 must leave their output rows untouched.  Returns a dict for the JSON line;
        raises on mismatch (every rank learns the verdict through the all-reduce in the caller)."""
        rows = 65_537 * world + 3
        cols = torch.arange(FEAT_DIM, device=device, dtype=torch.int64).view(1, -1)

        def kat(ids):
            return ((ids.view(-1, 1) * 3 + cols) & 0xFFFFF).to(torch.float32)

        t = make_partitioned(placement, rows, FEAT_DIM, lambda local, first: local.copy_(
            kat(torch.arange(first, first + local.shape[0], device=device, dtype=torch.int64))))
        g = torch.Generator(device=device).manual_seed(4242 + rank)
        ids = torch.randint(0, rows, (100_001,), generator=g, device=device)
        edges = torch.tensor(equal_entry_partition(rows, world), device=device)
        ids[:2 * world] = torch.cat([edges[:-1], edges[1:] - 1]).clamp_(0, rows - 1)   # first / last row of every rank
        ids[5000:5007] = -1
        ids = ids.to(id_dtype)
        got = t.gather(ids)
        torch.cuda.synchronize()
        if os.environ.get("WGAMD_BENCH_TEST_CORRUPT"):
            got[777, 3] += 1.0        # test hook: the pre-flight must notice ONE wrong element
        want = kat(ids.to(torch.int64))
        live = (ids >= 0)
        ok = bool(torch.equal(got[live], want[live]))
        # the timed groups take the DE-DUPLICATED form of the same fetch (every distinct row through the exchange once, expanded
        # locally: gather(dedup=...), wholegraph_amd/tensor.py) — the known answer must come back through it as well
        got_d = t.gather(ids, dedup=True)
        torch.cuda.synchronize()
        ok_d = bool(torch.equal(got_d[live], want[live]))
        ok = ok and ok_d
        info = {"rows": rows, "ids_per_rank": int(ids.numel()), "bit_exact": ok, "dedup_bit_exact": ok_d, "path": feature_fetch_path(t)}
        if hasattr(t, "comm") and hasattr(t.comm, "rccl_info"):
            info["rccl_ranks"], info["rccl_version"] = t.comm.rccl_info()
            if args.dist_backend == "nccl" and info["rccl_ranks"] != args.gpus:
                raise RuntimeError("selftest: the RCCL communicator holds %r ranks but --gpus is %d" % (info["rccl_ranks"], args.gpus))
        if hasattr(t, "destroy"):
            t.destroy()        # collective for a peer-mapped handle (every rank is here)
        if not ok:
            bad = int((got[live] != want[live]).any(dim=1).sum())
            raise RuntimeError("selftest: %d of %d gathered rows differ from the known answer (%s)" % (bad, int(live.sum()), info["path"]))
        return info

    # ---- step geometry: FIXED launch shape (G mini-batches per call group), independent of --steps ----
    auto_group = args.call_group <= 0
    if auto_group:
        from cugraph_pyg_amd.sampler.sampler import default_local_seeds_per_call
        args.call_group = max(1, default_local_seeds_per_call(FANOUT, BATCH, 8 if id_dtype == torch.int64 else 4) // BATCH)
    G, gps = args.call_group, args.groups_per_step
    std_call_group = G if auto_group else -1     # the committed profiles / PMC passes are of the default launch shape
    groups = args.steps * gps
    # at least 3 untimed call groups: sizes differ from group to group, the caching allocator and the two-stream pipeline
    # are only in steady state after a few of them (--warmup is honoured as a minimum)
    warm_groups = max(args.warmup * gps, 3)
    total_groups = groups + warm_groups
    gseed = torch.Generator(device=device).manual_seed(7 + rank)  # every rank its own seed shard
    distinct = min(total_groups, 48)           # seed sets are reused round-robin beyond this (RNG seeds still differ)
    need = distinct * G * BATCH
    reps = (need + V - 1) // V
    order = torch.cat([torch.randperm(V, generator=gseed, device=device) for _ in range(reps)])
    batches = order[:need].view(distinct, G * BATCH).to(id_dtype).contiguous()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    host_prof = {"sample_enqueue": 0.0, "forward_enqueue": 0.0, "wait_for_walk": 0.0, "groups": 0}
    last_per_rank = []

    def run_groups(pipe, first, last, timers=None, sizes=None, mode="split"):
        """software pipeline: walk(g+1) is enqueued before forward(g) waits for the sizes of g"""
        pipe.mode = mode
        pending = pipe.sample(batches[first % distinct], first)
        for g in range(first, last):
            t0 = time.perf_counter()
            nxt = pipe.sample(batches[(g + 1) % distinct], g + 1) if g + 1 < last else None
            t1 = time.perf_counter()
            w0 = pipe.host_wait_s
            _, sz = pipe.forward(*pending, timers=timers, mode=mode)
            host_prof["sample_enqueue"] += t1 - t0
            host_prof["forward_enqueue"] += time.perf_counter() - t1 - (pipe.host_wait_s - w0)
            host_prof["wait_for_walk"] += pipe.host_wait_s - w0
            host_prof["groups"] += 1
            if sizes is not None:
                sizes.append(sz)
            if nxt is not None and timers is None and pipe.gather_ahead and not mode.endswith("_fetch"):
                pipe.prefetch(nxt)
            pending = nxt

    def measure(pipe, mode):
        if fwd_masked is not None:
            with torch.cuda.stream(fwd_masked):
                return measure_(pipe, mode)
        return measure_(pipe, mode)

    def measure_(pipe, mode):
        """warm-up steps, then EXACTLY args.steps steps between barriers; (max seconds over ranks, total edges)"""
        run_groups(pipe, 0, warm_groups, mode=mode)
        barrier()
        t_start = time.perf_counter()
        szs = []
        for k in host_prof:
            host_prof[k] = 0
        run_groups(pipe, warm_groups, total_groups, sizes=szs, mode=mode)
        barrier()
        secs = time.perf_counter() - t_start
        if args.host_profile and rank == 0:
            n_g = max(host_prof["groups"], 1)
            print("[host-profile %s] per call group: total %.3f ms | host: walk enqueue %.3f, forward enqueue %.3f, blocked on "
                  "the walk %.3f" % (mode, secs / n_g * 1e3, host_prof["sample_enqueue"] / n_g * 1e3,
                                     host_prof["forward_enqueue"] / n_g * 1e3, host_prof["wait_for_walk"] / n_g * 1e3),
                  file=sys.stderr, flush=True)
        st = torch.tensor([secs, float(sum(sum(v[0::2]) for v in szs))], dtype=torch.float64, device=device)
        if world > 1:
            every = [torch.zeros_like(st) for _ in range(world)]
            dist.all_gather(every, st)
            per_rank = [(float(t[0]), float(t[1])) for t in every]
        else:
            per_rank = [(float(st[0]), float(st[1]))]
        last_per_rank[:] = per_rank      # (seconds, edges) of every rank for the pass just measured
        return max(t for t, _ in per_rank), sum(e for _, e in per_rank)

    # ---- per-stage HIP-event timing pass (same pipeline, same stream; outside the timed region) --
    def probe_stages(pipe, mode, n_groups):
        if fwd_masked is not None:
            with torch.cuda.stream(fwd_masked):
                return probe_stages_(pipe, mode, n_groups)
        return probe_stages_(pipe, mode, n_groups)

    def probe_stages_(pipe, mode, n_groups):
        acc, sizes_p = {}, []
        pipe.mode = mode
        for g in range(warm_groups, warm_groups + n_groups):
            timers = []
            w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ws = pipe.walk_stream if pipe.walk_stream is not None else torch.cuda.current_stream()
            torch.cuda.synchronize()
            w0.record(ws)
            pend = pipe.sample(batches[g % distinct], g)
            w1.record(ws)
            _, sz = pipe.forward(*pend, timers=timers, mode=mode)
            torch.cuda.synchronize()
            timers.append(("walk(sample+renumber x%d)" % L, w0, w1))
            for name, a, b in timers:
                acc[name] = acc.get(name, 0.0) + a.elapsed_time(b)
            sizes_p.append(sz)
        return {k: v / n_groups for k, v in acc.items()}, sizes_p      # per call group

    results = {}
    placement_errors = {}
    selftests = {}
    walk_masked = fwd_masked = None
    if args.walk_cus > 0:
        n_cu = torch.cuda.get_device_properties(device).multi_processor_count
        walk_masked = masked_stream(device, n_cu - args.walk_cus, args.walk_cus)
        fwd_masked = masked_stream(device, 0, n_cu - args.walk_cus)
    def run_placement(placement, feat):
        """Measure one feature placement: headline pass, variants (first placement only), stage timings."""
        nonlocal batches
        partitioned = placement != "replicated"
        pipe = SagePipeline(row_ptr, col, feat, device, G, overlap_walk=not args.no_overlap, walk_priority=args.walk_priority,
                            walk_stream=walk_masked, dedup=args.fetch == "distinct")
        # headline: explicit feature gather (the reference's flow), then every SAGE layer whose shape allows it as ONE
        # kernel; --layer-kernel split keeps the aggregation kernel + library GEMM pair for every layer
        fusable = nn_mod.sage_layer_fused_preferred(pipe.dims[0], pipe.dims[1])
        head_mode = "fused" if (args.layer_kernel != "split" and fusable) else "split"
        dt, edges_total = measure(pipe, head_mode)
        per_rank = list(last_per_rank)
        # variants on the same groups: the other layer kernel, and the feature fetch folded into layer 1
        variants = {}
        others = (["split"] if head_mode == "fused" else (["fused"] if fusable else [])) + (
            [] if partitioned else ["split_fetch"] + (["fused_fetch"] if fusable else []))
        notes = {"split": "aggregate kernel -> [agg|x_self] in HBM -> hipBLASLt GEMM (two kernels per layer)",
                 "fused": "every SAGE layer the shape allows as ONE kernel (wgamd_sage_layer_fused_f32: neighbour rows -> "
                          "LDS operand tile -> MFMA), explicit feature gather kept",
                 "split_fetch": "feature fetch folded into the layer-1 aggregation kernel (wgamd_sage_aggregate_fetch_f32), "
                                "then GEMM",
                 "fused_fetch": "feature fetch + aggregation + MFMA transform of layer 1 in ONE kernel "
                                "(wgamd_sage_layer_fused_f32 reading the feature table through n_id): x = feat[n_id] never "
                                "exists"}
        for m in ([] if (args.no_variants or world > 1) else others):
            vs, ve = measure(pipe, m)
            variants[m] = {"value": ve / vs, "ms_per_step": vs / args.steps * 1e3, "note": notes[m]}
        if pipe.dedup and not (args.no_variants or world > 1):
            pipe_full = SagePipeline(row_ptr, col, feat, device, G, overlap_walk=not args.no_overlap, walk_priority=args.walk_priority,
                                     walk_stream=walk_masked, dedup=False)
            pipe_full.convs, pipe_full.w_t, pipe_full.bias = pipe.convs, pipe.w_t, pipe.bias
            vs, ve = measure(pipe_full, head_mode)
            variants["materialised_full"] = {"value": ve / vs, "ms_per_step": vs / args.steps * 1e3,
                                             "note": "x = feat[n_id] gathered row for row (10.9 M rows per products call group), layer 1 "
                                                     "reads x directly: the headline pipeline of rounds 1-5"}
            del pipe_full
        lazy_pass = None
        if placement == "partitioned_mapped" and fusable and not args.no_variants:
            # the same groups with the fetch folded into layer 1 OVER THE MAPPING (no gather, remote rows read by the layer
            # kernel across xGMI): every rank runs it (measure() is collective), reported next to the placement's gather value
            vs, ve = measure(pipe, "fused_fetch")
            lazy_pass = {"value": ve / vs, "ms_per_step": vs / args.steps * 1e3,
                         "note": "layer 1 reads the peer-mapped partitions itself (wgamd_mapped_row_offsets + "
                                 "WGAMD_IDS_BYTE_OFFSETS): x = feat[n_id] never exists"}
        if col_alt is not None and world == 1:
            # the same pipeline with int32 ids (csr_col, seeds, node lists): the WholeGraph test default
            pipe32 = SagePipeline(row_ptr, col_alt, feat, device, G, overlap_walk=not args.no_overlap, walk_priority=args.walk_priority,
                                  dedup=args.fetch == "distinct")
            b64 = batches
            batches = b64.to(torch.int32)
            vs, ve = measure(pipe32, head_mode)
            variants["ids_int32"] = {"value": ve / vs, "ms_per_step": vs / args.steps * 1e3,
                                     "note": "csr_col / seeds / node ids as int32 (V < 2^31: the WholeGraph test default; "
                                             "headline: int64, the cugraph_pyg convention)"}
            batches = b64
            del pipe32
        if world == 1 and not partitioned and not args.no_variants:
            # the drop-in loader API on the same graph, table, weights and seed stream
            try:
                variants.update(loader_api_variants(row_ptr, col, feat.local_tensor, pipe.convs, order, min(groups, 24), G))
            except Exception as exc:   # a variant must never cost the headline line
                variants["loader_api"] = {"value": None, "error": repr(exc)[:300]}
        stage_n = max(10, min(groups, 20))
        stage_ms, psizes = probe_stages(pipe, head_mode, stage_n)
        # the BASELINE metric also names the stand-alone SAGEConv SpMM: time it (aggregate kernel + GEMM) when the
        # headline path runs the layer as one kernel
        split_ms = probe_stages(pipe, "split", 10)[0] if head_mode != "split" else stage_ms
        results[placement] = dict(dt=dt, edges=edges_total, variants=variants, stage_ms=stage_ms, psizes=psizes,
                                  split_ms=split_ms, head_mode=head_mode, stage_n=stage_n, pipe=pipe, feat=feat,
                                  per_rank=per_rank, lazy_pass=lazy_pass)

    def emit_line():
        """Rank 0 prints the ONE JSON line from whatever placements were measured; the headline is the first of HEAD_PREF
        that went through (N > 1: the RCCL all-to-all-v exchange).  Nothing measured = an error line, exit code 1."""
        measured = [p_ for p_ in HEAD_PREF if p_ in results]
        if not measured:
            if rank == 0:
                print(json.dumps({"metric": "sampled-edges/sec", "value": None, "unit": "sampled-edges/s", "n_gpus": world,
                                  "error": "no feature placement could be measured", "placement_errors": placement_errors,
                                  "selftest": selftests}), flush=True)
            return False
        head_name = measured[0]
        head = results[head_name]
        pipe, feat = head["pipe"], head["feat"]
        partitioned = head_name != "replicated"
        dt, edges_total, variants, stage_ms, psizes, split_ms, head_mode, stage_n = (
            head[k] for k in ("dt", "edges", "variants", "stage_ms", "psizes", "split_ms", "head_mode", "stage_n"))
        fused = variants.get("fused_fetch") or variants.get("split_fetch")
        hop_e = [sum(s[2 * k] for s in psizes) / stage_n for k in range(L)]       # edges per call group, seed hop first
        hop_u = [sum(s[2 * k + 1] for s in psizes) / stage_n for k in range(L)]   # unique nodes after each hop
        n_src = hop_u[L - 1]

        if rank == 0:
            # algorithmic bytes per launch; b = id bytes, fp32 features
            F = FEAT_DIM
            idb = 4 if id_dtype == torch.int32 else 8
            n_fetch = (sum(pipe.distinct_rows) / len(pipe.distinct_rows)) if (pipe.dedup and pipe.distinct_rows) else n_src
            kernels = {"gather": ("row_copy_kernel", n_fetch * ((8 if pipe.dedup else idb) + 2 * 4 * F))}
            layer1_ids = "int" if pipe.dedup else None    # layer 1 reads the gathered rows through the int32 inverse index
            spmm_root = {}
            for j in range(L):
                k = L - 1 - j
                fj = pipe.dims[j]
                n_dst = hop_u[k - 1] if k >= 1 else G * BATCH
                # STRICT SURVEY §8(d): E(4F+4) + N_dst(4F+8).  The kernel also copies the root row next to the aggregate
                # (one more row read and written per destination): that term is reported under its own key, never in spmm_GBps
                kernels[spmm_label(j)] = ("spmm_csr_kernel", hop_e[k] * (4 * fj + 4) + n_dst * (4 * fj + 8))
                spmm_root[spmm_label(j)] = n_dst * (8 * fj + 8)
            for j in range(L):   # one-kernel layers: same reads minus the [agg|x_self] round trip, plus the output write
                k = L - 1 - j
                fj, nj = pipe.dims[j], pipe.dims[j + 1]
                n_dst = hop_u[k - 1] if k >= 1 else G * BATCH
                # (the layer runs at its output width padded to 64 / 128 / 256: 47 classes -> 64 columns)
                kname = "sage_layer_mfma_kernel" if (nn_mod.sage_layer_fused_precision() == "bf16x3" and nn_mod.L.lib(
                ).wgamd_sage_layer_bf16x3_supported(fj, nn_mod._padded_width(nj))) else "sage_layer_fused_kernel"
                kernels["sage_layer%d(fused)" % (j + 1)] = (kname, hop_e[k] * (4 * fj + 4) + n_dst * (4 * fj + 16) + n_dst * 4 * nj)
            std_shape = G == std_call_group and args.nodes == wv and args.edges == we   # the shape the committed profiles are of

            def roof_entry(st):
                """The roofline of one stage's kernel: algorithmic bytes over the live HIP-event time (`frac`), over the committed
                rocprofv3 average (`frac_profiled`), the committed PMC traffic, and for a one-kernel layer its matrix side."""
                ach = kernels[st][1] / (stage_ms[st] * 1e-3) / 1e9
                r = {"bound": "hbm", "kernel": kernels[st][0], "stage": st, "achieved": round(ach, 1),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                     "traffic": None, "algorithmic_bytes_per_launch": int(kernels[st][1]),
                     "avg_launch_ms": round(stage_ms[st], 5),
                     "timing": "HIP events around the launch on the launch stream, one launch per call group of "
                               f"{G} mini-batches, averaged over {stage_n} call groups"}
                pref = layer1_ids if st == "sage_layer1(fused)" else ("long" if (st == "gather" and pipe.dedup) else None)
                if st.startswith("sage_layer") and st != "sage_layer1(fused)":
                    # a hidden layer reads its input directly (ids type void); rows wider than 128 floats take 64-lane groups
                    pref = "void,64" if pipe.dims[int(st[len("sage_layer")]) - 1] > 128 else "void"
                if std_shape:
                    prof = load_profiled_avg(r["kernel"], args.workload, prefer=pref)
                    if prof:    # the same algorithmic bytes over the committed profile's average launch duration
                        r["frac_profiled"] = round(kernels[st][1] / (prof["avg_ns"] * 1e-9) / 1e9 / HBM_PEAK_GBPS, 4)
                        r["profiled_avg_launch_ms"] = round(prof["avg_ns"] * 1e-6, 5)
                        r["profiled_source"] = "%s (%d launches, min %.1f us)" % (prof["source"], prof["calls"], prof["min_ns"] * 1e-3)
                    hit = load_pmc(r["kernel"], workload=args.workload, prefer=pref)
                    if hit:
                        r["traffic"] = hit["bytes"]
                        r["traffic_over_algorithmic"] = round(hit["bytes"] / kernels[st][1], 3)
                        r["traffic_source"] = hit["source"] + " kernel " + hit["kernel"]
                if st.startswith("sage_layer"):
                    # one-kernel layer: HBM-side and MFMA-side work of the same launch
                    jj = int(st[len("sage_layer")]) - 1
                    kk = L - 1 - jj
                    n_dst_ = hop_u[kk - 1] if kk >= 1 else G * BATCH
                    flops = 2.0 * n_dst_ * 2 * pipe.dims[jj] * pipe.dims[jj + 1]
                    r["hbm_frac"] = r["frac"]
                    if r["kernel"] != "sage_layer_mfma_kernel":
                        tfs = flops / (stage_ms[st] * 1e-3) / 1e12
                        r["mfma_TFps"], r["mfma_frac"] = round(tfs, 1), round(tfs / MFMA_F32_PEAK_TFPS, 4)
                        r["mfma_dtype"] = "f32 (v_mfma_f32_16x16x4_f32)"
                        if r["mfma_frac"] > r["frac"]:
                            r.update(bound="mfma", achieved=round(tfs, 1), peak=MFMA_F32_PEAK_TFPS, unit="TFLOP/s", frac=r["mfma_frac"])
                    else:
                        # 3-way bf16 split of both operands, 6 bf16 MFMA products per fp32 product (fp32 accumulate): the matrix
                        # work is 6 x flops on the bf16 pipe (2.5 PFLOP/s dense) -> far from binding, the launch is HBM-bound
                        tfs = 6.0 * flops / (stage_ms[st] * 1e-3) / 1e12
                        r["mfma_TFps"], r["mfma_frac"] = round(tfs, 1), round(tfs / MFMA_BF16_PEAK_TFPS, 4)
                        r["mfma_dtype"] = "bf16x3 split (6 v_mfma_f32_32x32x16_bf16 per fp32 product, fp32 accumulate)"
                return r

            # dominant kernel = the longest stage; the runner-up rides along under `also` (the feature gather and the
            # one-kernel layer 1 are within a per cent of each other on the products workload: which one leads depends on the box)
            ranked = sorted((k for k in kernels if k in stage_ms), key=lambda k: -stage_ms[k])
            dom = ranked[0] if ranked else None
            roofline = roof_entry(dom) if dom is not None else None
            if roofline is not None and len(ranked) > 1:
                roofline["also"] = roof_entry(ranked[1])
            spmm_gbps = spmm_root_gbps = spmm_pmc = None
            if SPMM1 in split_ms:
                spmm_gbps = kernels[SPMM1][1] / (split_ms[SPMM1] * 1e-3) / 1e9
                spmm_root_gbps = (kernels[SPMM1][1] + spmm_root[SPMM1]) / (split_ms[SPMM1] * 1e-3) / 1e9
                hit = load_pmc("spmm_csr_kernel", workload=args.workload) if std_shape else None
                if hit:     # real HBM utilisation of the launch: (2 x FETCH_SIZE + WRITE_SIZE) / time / 8 TB/s
                    spmm_pmc = {"hbm_util": round(hit["bytes"] / (split_ms[SPMM1] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                "traffic_bytes_per_launch": hit["bytes"], "source": hit["source"] + " kernel " + hit["kernel"]}
            cpu = None
            if not args.no_cpu_baseline and world == 1:   # the CPU baseline is timed at N=1 only (rank 0 owns the host cores)
                nb = min(2048, order.numel() // BATCH)   # time-bounded inside cpu_baseline (--cpu-budget seconds)
                cb = order[: nb * BATCH].view(nb, BATCH).to(id_dtype).cpu().numpy()  # same seed stream, one mini-batch at a time
                if V * FEAT_DIM * 4 > (8 << 30):
                    # papers100M-scale table: a lazily-zeroed host array of the same shape (only the gathered
                    # rows' pages are ever touched; values do not matter for the timing)
                    feat_h = np.zeros((V, FEAT_DIM), dtype=np.float32)
                elif partitioned:
                    feat_h = np.random.default_rng(0).random((V, FEAT_DIM), dtype=np.float32) * 2 - 1
                else:
                    feat_h = feat.local_tensor.cpu().numpy()
                weights = [(c.lin_l.weight.cpu(), c.lin_l.bias.cpu(), c.lin_r.weight.cpu()) for c in pipe.convs]
                cpu = cpu_baseline(row_ptr.cpu().numpy(), col.cpu().numpy(), feat_h, cb, weights, args.cpu_budget)
            wl_name = "RMAT-26" if args.workload == "rmat26" else "ogbn-" + args.workload
            prec = nn_mod.sage_layer_fused_precision() if hasattr(nn_mod, "sage_layer_fused_precision") else "f32"
            out = {
                "metric": "sampled-edges/sec (sample+renumber+feature-gather+SAGEConv fwd), %s-like fan-out %s"
                          % (wl_name, FANOUT),
                "value": edges_total / dt,
                "unit": "sampled-edges/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": dt / args.steps * 1e3,
                "higher_is_better": True,
                "scaling": "weak",
                "vs_baseline": None,
                "dtype": ("int32" if id_dtype == torch.int32 else "int64") + " ids (int32 timed as a variant) + f32 features" + ("" if head_mode != "fused" or prec == "f32" else
                                                       " (SAGE lin_l/lin_r product: bf16x3-split MFMA, f32 accumulate)"),
                "data": "synthetic",
                "config": {"workload": wl_name + "-like RMAT: V=%d, E=%d directed (CSR row_ptr i64 / col %s replicated per GPU), "
                                       "feat fp32 [V,%d]%s, batch %d/GPU, fan-out %s, %d-layer SAGEConv(mean) %s fwd, "
                                       "step = %d call groups of %d mini-batches"
                                       % (V, E, "i32" if id_dtype == torch.int32 else "i64", FEAT_DIM,
                                          " range-partitioned + xGMI feature fetch" if partitioned else
                                          ("" if world == 1 else " replicated per GPU"),
                                          BATCH, FANOUT, L, "-".join(str(d) for d in pipe.dims), gps, G),
                           "parallelism": ("dp%d seeds + feature all-to-all (%s)" % (world, feature_fetch_path(feat))) if partitioned
                           else "dp%d (seeds sharded, no data-path collective)" % world},
                "call_group": G,
                "batches_per_step": G * gps,
                "timed_region_ms": round(dt * 1e3, 2),
                "timed_call_groups": groups,
                "ms_per_batch": dt / (groups * G) * 1e3,
                "edges_per_batch": dict([("hop%d" % (k + 1), hop_e[k] / G) for k in range(L)] + [("unique_nodes", n_src / G)]),
                "stage_ms_per_call_group": {k: round(v, 5) for k, v in stage_ms.items()},
                "split_stage_ms_per_call_group": None if split_ms is stage_ms else {k: round(v, 5) for k, v in split_ms.items()},
                "spmm_GBps": None if spmm_gbps is None else round(spmm_gbps, 1),
                "spmm_frac_of_hbm_peak": None if spmm_gbps is None else round(spmm_gbps / HBM_PEAK_GBPS, 4),
                "spmm_accounting": "SURVEY.md §8(d) strict: E(4F+4) + N_dst(4F+8) bytes / HIP-event time of the layer-1 "
                                   "aggregation launch (effective bandwidth: part of x is served by L2/MALL)",
                "spmm_with_root_copy_GBps": None if spmm_root_gbps is None else round(spmm_root_gbps, 1),
                "spmm_hbm_util_pmc": spmm_pmc,
                "layer_kernel": head_mode,
                "feature_fetch": ("distinct rows of the call group gathered ONCE (list de-duplicated on the walk stream: "
                                  "wgamd_unique_bounded_live), layer 1 reads them through the inverse index; %d listed rows -> %d "
                                  "gathered per call group" % (int(n_src), int(n_fetch))) if pipe.dedup else
                                 "x = feat[n_id] row for row",
                # the same pipeline with EVERY listed row gathered (what the reference's loader fetches per mini-batch; the headline of
                # rounds 1-5), measured in this run — next to `value` at the top level so that neither figure is read without the other
                "value_row_for_row_fetch": (variants.get("materialised_full") or {}).get("value") if pipe.dedup else None,
                "fused_fetch_variant": fused,
                "variants": variants,
                "roofline": roofline,
                "cpu_baseline": cpu,
            }
            if cpu is not None:
                out["gpu_over_cpu"] = round(out["value"] / cpu["value"], 2)
            if placement_errors:
                out["placement_errors"] = placement_errors
            if world > 1 or partitioned:
                # every measured placement side by side; for the partitioned ones the per-GPU xGMI bytes of the feature fetch
                # (SURVEY §8(d) all-to-all bytes: n_remote (b + 4F)) and the fraction of the (world-1) x 153 GB/s links they
                # sustain during the gather stage
                link_peak = max(world - 1, 1) * 153.0

                def report(name):
                    pr = results[name]
                    rep = {"value": pr["edges"] / pr["dt"], "ms_per_step": pr["dt"] / args.steps * 1e3,
                           "per_rank_value": [round(e_ / t_, 1) for t_, e_ in pr["per_rank"]]}
                    if pr.get("lazy_pass"):
                        rep["fetch_in_layer"] = pr["lazy_pass"]
                    if name != "replicated":
                        p_src = sum(s_[2 * L - 1] for s_ in pr["psizes"]) / pr["stage_n"]
                        # the partitioned fetch sends every DISTINCT row of the call group once (the list is de-duplicated
                        # behind the walk; layer 1 reads the fetched rows through the inverse index); ids travel as int64
                        from wholegraph_amd.tensor import dedup_pays
                        dd = pr["pipe"].distinct_rows
                        deduped = bool(dd) and (pr["pipe"].dedup or dedup_pays(int(p_src), V, world))
                        wire_rows = sum(dd) / len(dd) if deduped else p_src
                        a2a = wire_rows * (world - 1) / max(world, 1) * (8 + 4 * F)
                        rep.update(requested_rows_per_call_group=int(p_src), deduplicated=deduped,
                                   wire_rows_per_call_group=int(wire_rows))
                        gname = next((k for k in pr["stage_ms"] if k.startswith("gather")), None)
                        gms = pr["stage_ms"].get(gname) if gname else None
                        rep.update(gather_stage=gname, gather_ms_per_call_group=None if gms is None else round(gms, 4),
                                   feature_fetch=feature_fetch_path(pr["feat"]), all_to_all_bytes_per_gpu=int(a2a),
                                   xgmi_frac=None if (gms is None or world == 1) else round(a2a / (gms * 1e-3) / 1e9 / link_peak, 4))
                    return rep
                out["headline_placement"] = head_name
                out["placements"] = {name: report(name) for name in HEAD_PREF if name in results}
                out["per_rank_value"] = out["placements"][head_name]["per_rank_value"]
                out["xgmi_peak_GBps"] = link_peak
                if partitioned:
                    out["all_to_all_bytes_per_gpu"] = out["placements"][head_name]["all_to_all_bytes_per_gpu"]
                    out["xgmi_frac"] = out["placements"][head_name]["xgmi_frac"]
                st = selftests.get(head_name) or {}
                out["rccl_ranks"] = st.get("rccl_ranks")
                out["rccl_version"] = st.get("rccl_version")
            if selftests:
                out["selftest"] = selftests
            # RCCL writes a version banner through C stdio; push it out first so the JSON is the LAST line
            import ctypes
            ctypes.CDLL(None).fflush(None)
            print(json.dumps(out), flush=True)
        return True

    def arm_watchdog(seconds, what):
        """A placement with collectives in it runs under a watchdog: if it has not finished in time (a rank threw inside a
        collective and left the others waiting), the line is printed from what IS measured — the placements are measured
        safest first — and the process ends: a scaling run never loses its line to a path that hangs."""
        import threading

        def fire():
            placement_errors[what] = "timed out after %d s (watchdog)" % seconds
            results.pop(what, None)
            ok = False
            try:
                ok = emit_line()
            finally:
                os._exit(0 if ok else 1)
        t = threading.Timer(seconds, fire)
        t.daemon = True
        t.start()
        return t

    import signal
    for placement in list(placements):
        protected = world > 1 or placement != "replicated"
        if not protected:
            run_placement(placement, make_table(placement))
            continue
        # Every rank agrees on whether the table could be built and the pre-flight passed before anyone enters a collective
        # of the measurement; the measurement runs under the watchdog; every rank agrees again on whether it went through.
        dog = arm_watchdog(args.extra_placement_timeout, placement)

        # ... and if another rank DIES in it (a fault is not an exception), the launcher sends SIGTERM to the rest: rank 0
        # answers with the line it already has instead of going down with it
        def on_term(signum, _frame, what=placement):
            placement_errors[what] = "a rank died during this placement (signal %d)" % signum
            results.pop(what, None)
            ok = False
            try:
                ok = emit_line()
            finally:
                os._exit(0 if ok else 1)
        old_term = signal.signal(signal.SIGTERM, on_term) if world > 1 else None
        hooks = placement != placements[0]       # test hooks act on the first placement after the first
        feat, ok = None, 1
        try:
            if placement != "replicated" and (world > 1 or args.selftest):
                selftests[placement] = selftest(placement)
            feat = make_table(placement)
        except Exception as e:       # noqa: BLE001
            placement_errors[placement] = repr(e)
            if placement in selftests or placement == "replicated":
                pass
            else:
                selftests[placement] = {"bit_exact": False, "error": repr(e)}
            ok = 0
        def agree(ok_here, what=placement):
            """min over the ranks of "this placement went through here".  If the agreement ITSELF fails a peer is gone (a dead
            rank raises nothing over there, but its closed connections do here — and an uncaught exception would end this rank
            before the launcher's SIGTERM reaches the handler above): answer with the line already measured, as on_term does."""
            flag_ = torch.tensor([ok_here], device=device)
            if world > 1:
                try:
                    dist.all_reduce(flag_, op=dist.ReduceOp.MIN)
                except Exception as e:   # noqa: BLE001
                    placement_errors[what] = "a rank died during this placement (%s)" % (repr(e)[:200],)
                    results.pop(what, None)
                    done = False
                    try:
                        done = emit_line()
                    finally:
                        os._exit(0 if done else 1)
            return int(flag_)

        flag = agree(ok)
        if flag == 1:
            try:
                if hooks and os.environ.get("WGAMD_BENCH_TEST_STALL") and rank == world - 1:
                    time.sleep(10 ** 6)      # test hook: one rank never reaches the collectives of this pass
                if hooks and os.environ.get("WGAMD_BENCH_TEST_DIE") and rank == world - 1:
                    os.kill(os.getpid(), signal.SIGKILL)   # test hook: one rank dies in this pass
                run_placement(placement, feat)
            except Exception as e:   # noqa: BLE001
                placement_errors[placement] = repr(e)
                ok = 0
            flag = agree(ok)
        dog.cancel()
        if world > 1:
            signal.signal(signal.SIGTERM, old_term if old_term is not None else signal.SIG_DFL)
        if flag == 0:
            placement_errors.setdefault(placement, "another rank could not build / verify / measure this placement")
            results.pop(placement, None)
    ok = emit_line()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(1)


def feature_fetch_path(feat):
    """How a partitioned table's remote rows travel (reported, not chosen here)."""
    try:
        return feat.fetch_path()
    except AttributeError:
        return "all-to-all-v (RCCL send/recv groups)"


if __name__ == "__main__":
    main()
