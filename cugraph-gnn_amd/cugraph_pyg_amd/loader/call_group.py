"""Call-group iteration of a ``NeighborLoader`` epoch: G mini-batches at a time as ONE block-diagonal graph.

``cugraph_pyg`` already samples ``local_seeds_per_call`` seeds per library call and only then cuts the result into
mini-batches (/root/reference/python/cugraph-pyg/cugraph_pyg/sampler/distributed_sampler.py:279-343,391-410;
sampler/sampler.py:51-165 hands them out one ``Data`` at a time).  On an MI355X a mini-batch of 1024 seeds is ~25 us of
device work — less than the Python it takes to build its ``Data`` — so a loop that wants the device's speed consumes the call
group as a whole, the way PyG batches small graphs (``torch_geometric.data.Batch``: node lists back to back, edge indices
shifted): ``loader.call_groups()`` yields ``CallGroup`` objects whose ``x`` / ``edge_index`` / ``n_id`` cover all G
mini-batches, ``batch_ptr`` says which rows belong to which mini-batch, and ``layer_graph(j)`` is the TRIMMED graph of layer j
(what ``torch_geometric.utils.trim_to_layer`` computes from ``num_sampled_nodes`` / ``num_sampled_edges``) in the CSR form
``wholegraph_amd.nn.SAGEConv`` takes.  Every mini-batch inside is bit for bit what ``for batch in loader`` yields
(``to_data_list()``; tests/test_gpu_call_group_loader.py).

``x`` is LAZY when the feature table lives whole on this device: a ``wholegraph_amd.nn.LazyRows`` (table + ``n_id``) that
``nn.SAGEConv`` reads through directly, so the gathered ``[N, F]`` copy — the largest tensor of a mini-batch — never exists;
any other use gathers it once.
"""
from typing import List, Optional

import torch

from wholegraph_amd import _lib as L
from wholegraph_amd.env import get_stream
from wholegraph_amd.nn import HeteroLayerGraph, HopGraph, LayerGraph, LazyRows, RelationHop, mapped_lazy_rows


def _peer_mapped_f32(wm) -> bool:
    """A handle-backed float32 [rows, F] table whose partitions are all addressable from this GPU (CHUNKED / CONTINUOUS over
    more than one rank of a node)."""
    path = getattr(wm, "fetch_path", None)
    try:
        return (path is not None and "peer-mapped" in path() and wm.dtype == torch.float32 and wm.dim() == 2
                and wm.comm.get_size() > 1)
    except Exception:      # noqa: BLE001  (a tensor class without these queries)
        return False


class CallGroup:
    """G consecutive mini-batches of an epoch (``n_batches``), homogeneous graph, PyG sampling semantics.

    Node rows are batch-major: mini-batch b owns rows ``[node_ptr[b], node_ptr[b+1])`` of ``n_id`` / ``x`` and its seeds are
    the first rows of that stretch.  The model's output for the seeds is batch-major too: mini-batch b's seeds are rows
    ``[batch_ptr[b], batch_ptr[b+1])`` of the last layer's output."""

    def __init__(self, walk_result, feature_store, graph, first_batch: int, input_id: torch.Tensor, sizes_h: torch.Tensor,
                 event, walk_stream):
        self._res, self._fs, self._graph = walk_result, feature_store, graph
        self.first_batch, self.n_batches, self.hops = first_batch, walk_result.n_batches, walk_result.hops
        self.input_id = input_id
        self._sizes_h, self._event, self._walk_stream = sizes_h, event, walk_stream
        self._ready = False
        self._layers = {}
        self._edge_index = self._e_id = None

    # ---- sizes: one pinned read-back per call group ----------------------------------------------------------------
    def _wait(self):
        if self._ready:
            return
        self._event.synchronize()
        v = self._sizes_h.tolist()
        H = self.hops
        self.num_nodes = v[0]
        self._n_front = v[1:2 + H]            # live entries of frontier k (k = 0: the seeds); [H]: new vertices of the last hop
        self._n_edges = v[2 + H:2 + 2 * H]
        self.num_edges = sum(self._n_edges)
        self.num_seeds = self._n_front[0]
        if self._walk_stream is not None:     # allocated on the walk stream, consumed on the caller's
            main = torch.cuda.current_stream()
            r = self._res
            for t in [r.nodes, r.node_seg] + r.offsets + r.row_local + r.col_local + r.edge_gid + r.frontier_seg + \
                    r.frontier_batch + r.frontier_local0:
                t.record_stream(main)
        self._ready = True

    # ---- nodes -----------------------------------------------------------------------------------------------------
    @property
    def n_id(self) -> torch.Tensor:
        self._wait()
        return self._res.nodes[:self.num_nodes]

    @property
    def node_ptr(self) -> torch.Tensor:
        """int32 [G + 1]: rows of mini-batch b in ``n_id`` / ``x``."""
        return self._res.node_seg

    @property
    def batch_ptr(self) -> torch.Tensor:
        """int32 [G + 1]: rows of mini-batch b's seeds in the model output (and in ``input_id``)."""
        return self._res.frontier_seg[0]

    @property
    def batch(self) -> torch.Tensor:
        """The seeds of all mini-batches (global ids), batch-major — ``Data.batch`` of every mini-batch back to back."""
        self._wait()
        seg = self._res.node_seg[:-1].long()
        cnt = (self._res.frontier_seg[0][1:] - self._res.frontier_seg[0][:-1]).long()
        rows = torch.repeat_interleave(seg - self._res.frontier_seg[0][:-1].long(), cnt, output_size=self.num_seeds) + \
            torch.arange(self.num_seeds, device=seg.device)
        return self._res.nodes[rows]

    def node_attr(self, name: str, group_name=None, lazy: bool = True):
        """A stored node attribute for all rows of the call group: ``LazyRows`` (table + ``n_id``, nothing gathered) when
        the table is a float32 matrix held whole on this device and ``lazy``; the gathered rows otherwise (one fetch per
        call group — a collective when the FeatureStore is partitioned)."""
        from ..sampler.sampler import _fetch_rows_agreed
        self._wait()
        if group_name is None:
            names = sorted({a.group_name for a in self._fs.get_all_tensor_attrs()
                            if a.attr_name == name and not isinstance(a.group_name, tuple)})
            if len(names) != 1:
                raise KeyError(f"node attribute {name!r}: found in groups {names}")
            group_name = names[0]
        t = self._fs[group_name, name, None]
        wm = getattr(t, "_tensor", None)
        table = getattr(wm, "local_tensor", None)
        if (lazy and table is not None and not getattr(wm, "is_distributed", True) and table.is_cuda and table.dim() == 2
                and table.dtype == torch.float32 and table.stride(1) == 1):
            return LazyRows(table, self.n_id)
        if lazy and _peer_mapped_f32(wm):
            return mapped_lazy_rows(wm, self.n_id)      # partitions on several GPUs, every one mapped here: read in the layer
        return _fetch_rows_agreed(t, self.n_id)

    @property
    def x(self):
        return self.node_attr("x")

    @property
    def y(self):
        return self.node_attr("y", lazy=False)

    # ---- edges -----------------------------------------------------------------------------------------------------
    def layer_graph(self, layer: int) -> LayerGraph:
        """The hops layer ``layer`` (0 = the one that reads ``x``) of an H-layer model runs over, trimmed: layer j
        computes rows only for the vertices the seeds can still see through the layers after it — the vertices discovered
        by hops < H - j — from the edges of hops <= H - 1 - j (``trim_to_layer``).  Its output rows: the seeds of all
        mini-batches first (batch-major), then the vertices discovered by hop 0, hop 1, ... (each batch-major); the LAST
        layer's output is exactly the seeds' rows.  ``col`` / ``self_rows`` of a hop index the layer's input: ``x`` for layer
        0, the previous layer's output otherwise."""
        self._wait()
        H, G, res = self.hops, self.n_batches, self._res
        if not 0 <= layer < H:
            raise IndexError(f"layer {layer} of a {H}-hop call group")
        if layer in self._layers:
            return self._layers[layer]
        dev = res.nodes.device
        if layer == 0:
            zeros = torch.zeros(G + 1, dtype=torch.int32, device=dev)
            seg_tab = torch.stack([zeros, res.node_seg.to(torch.int32)]).contiguous()
            seg_base = torch.zeros(1, dtype=torch.int64, device=dev)
            n_seg = 1
        else:
            ran = list(range(H - layer + 1))              # the hops the previous layer ran = the segments of its output
            rows, base, at = [], [], 0
            for k in ran:
                l0 = torch.zeros(G + 1, dtype=torch.int32, device=dev)
                l0[:G] = res.frontier_local0[k]
                rows += [l0, res.frontier_seg[k].to(torch.int32)]
                base.append(at)
                at += self._n_front[k]
            seg_tab = torch.stack(rows).contiguous()
            seg_base = torch.tensor(base, dtype=torch.int64, device=dev)
            n_seg = len(ran)
        hops = []
        for k in range(H - layer):
            n_f, n_e = self._n_front[k], self._n_edges[k]
            self_rows = torch.empty(n_f, dtype=torch.int64, device=dev)
            col = torch.empty(max(n_e, 1), dtype=torch.int32, device=dev)
            L.check(L.lib().wgamd_call_group_layer_cols(
                res.offsets[k].data_ptr(), res.frontier_batch[k].data_ptr(), res.frontier_seg[k].data_ptr(),
                res.frontier_local0[k].data_ptr(), res.row_local[k].data_ptr(), n_f, G, n_seg, seg_tab.data_ptr(),
                seg_base.data_ptr(), self_rows.data_ptr(), col.data_ptr(), get_stream()), "wgamd_call_group_layer_cols")
            hops.append(HopGraph(res.offsets[k][:n_f + 1], col[:n_e], self_rows))
        lg = LayerGraph(hops)
        lg._keep = (seg_tab, seg_base)
        self._layers[layer] = lg
        return lg

    def _coo(self):
        """All sampled edges, hop-major, as rows of ``n_id``: (source row, destination row, CSR slot)."""
        self._wait()
        if self._edge_index is None:
            res, G = self._res, self.n_batches
            src, dst, gid = [], [], []
            node0 = res.node_seg[:-1].long()
            for k in range(self.hops):
                n_f, n_e = self._n_front[k], self._n_edges[k]
                deg = (res.offsets[k][1:n_f + 1] - res.offsets[k][:n_f]).long()
                b_of_e = torch.repeat_interleave(res.frontier_batch[k][:n_f].long(), deg, output_size=n_e)
                src.append(res.row_local[k][:n_e].long() + node0[b_of_e])
                dst.append(res.col_local[k][:n_e].long() + node0[b_of_e])
                gid.append(res.edge_gid[k][:n_e])
            self._edge_index = torch.stack([torch.cat(src), torch.cat(dst)])
            g = torch.cat(gid)
            self._e_id = self._graph.edge_id[g] if self._graph.edge_id is not None else g
        return self._edge_index, self._e_id

    @property
    def edge_index(self) -> torch.Tensor:
        """int64 [2, E]: PyG convention (row 0 = source = the sampled neighbour, row 1 = destination), rows of ``n_id``, all
        mini-batches, hop-major — the block-diagonal union of every mini-batch's ``edge_index``."""
        return self._coo()[0]

    @property
    def e_id(self) -> torch.Tensor:
        return self._coo()[1]

    @property
    def num_sampled_edges(self) -> List[int]:
        """Edges per hop, summed over the mini-batches."""
        self._wait()
        return list(self._n_edges)

    @property
    def num_sampled_nodes(self) -> List[int]:
        """Vertices per hop (seeds first), summed over the mini-batches."""
        self._wait()
        return list(self._n_front)

    # ---- the mini-batches one at a time --------------------------------------------------------------------------------
    def to_data_list(self):
        """The G mini-batches as the ``Data`` objects ``for batch in loader`` yields (same tensors, bit for bit)."""
        from ..sampler.sampler import filter_store_from_group, group_attribute_views
        self._wait()
        outs = self._res.finalize_batches(self._graph.edge_id)
        views = group_attribute_views(self._fs, self._res.group_context)
        datas, seed_ptr = [], self._res.frontier_seg[0].tolist()
        for j, (node, row, col, edge, nn, ne) in enumerate(outs):
            d = filter_store_from_group(self._fs, views, j, node, row, col, edge)
            d.n_id, d.e_id = node, edge.to(torch.long)
            d.batch = node[:nn[0]]
            d.num_sampled_nodes, d.num_sampled_edges = torch.tensor(nn), torch.tensor(ne)
            d.input_id = self.input_id[seed_ptr[j]:seed_ptr[j + 1]]
            d.batch_size = d.input_id.size(0)
            datas.append(d)
        return datas


class HeteroCallGroup:
    """G consecutive mini-batches of a HETEROGENEOUS loader epoch as one block-diagonal graph (BASELINE configs[4]; the
    reference's surface: ``NeighborLoader`` over a heterogeneous ``GraphStore``, loader/neighbor_loader.py:173-201,
    sampler/sampler.py:231-502, call shape examples/mag_lp_mnmg.py:141).  Per node type the vertices of all mini-batches are
    batch-major (``n_id[t]``, offsets ``node_ptr[t]``), a batch's vertices in discovery order — so "the vertices the seeds can
    still see through k more layers" is a per-batch PREFIX of every list, which is what the trimmed ``layer_graph(j)``
    renumbers against.  Every mini-batch inside equals what ``for batch in loader`` yields (same walk, same seeds)."""

    def __init__(self, rec, walk, feature_store, seed_type, first_batch: int, input_id, sizes_h, event, walk_stream, cseg):
        self._rec, self._walk, self._fs, self.seed_type = rec, walk, feature_store, seed_type
        self.first_batch, self.n_batches, self.hops = first_batch, walk.G, walk.hops
        self.input_id, self._sizes_h, self._event, self._walk_stream, self._cseg = input_id, sizes_h, event, walk_stream, cseg
        self.node_types, self.edge_types = list(walk.ntypes), list(walk.etypes)
        self._ready = False
        self._rows, self._layers, self._cseg32 = {}, {}, {}

    def _wait(self):
        if self._ready:
            return
        self._event.synchronize()
        it = iter(self._sizes_h.tolist())
        self.num_nodes = {t: next(it) for t in self.node_types}
        # vertices per type after k hops (k = 0: the seeds), summed over the mini-batches
        self._n_level = [{t: next(it) for t in self.node_types} for _ in range(self.hops)] + [dict(self.num_nodes)]
        self._live = [None if c is None else (next(it), next(it)) for c in self._rec["calls"]]   # (frontier entries, edges)
        self.num_edges = sum(lv[1] for lv in self._live if lv is not None)
        self.num_seeds = self._n_level[0][self.seed_type]
        if self._walk_stream is not None:      # allocated on the walk stream, consumed on the caller's
            main = torch.cuda.current_stream()
            for t in self.node_types:
                for k in ("nodes", "seg"):
                    self._rec["state"][t][k].record_stream(main)
            for c in self._rec["calls"]:
                if c is not None:
                    for k in ("offsets", "row", "f_batch", "f_seg", "f_local0"):
                        c[k].record_stream(main)
            for lvl in self._cseg:
                for v in lvl.values():
                    v.record_stream(main)
            # every other walk-stream tensor the consumer reads (batch_ptr = calls_seed_seg, the per-call edge counts)
            if torch.is_tensor(self._rec.get("calls_seed_seg")):
                self._rec["calls_seed_seg"].record_stream(main)
            for c in self._rec["calls"]:
                if c is not None and torch.is_tensor(c.get("counts")):
                    c["counts"].record_stream(main)
        self._ready = True

    # ---- nodes -----------------------------------------------------------------------------------------------------
    @property
    def n_id(self):
        """{node type: global (type-local) ids of the vertices of all mini-batches, batch-major}."""
        self._wait()
        return {t: self._rec["state"][t]["nodes"][:self.num_nodes[t]] for t in self.node_types}

    @property
    def node_ptr(self):
        return {t: self._rec["state"][t]["seg"] for t in self.node_types}

    def node_attr(self, name: str, lazy: bool = True):
        """{node type: attribute rows of its vertices} — ``LazyRows`` (table + ids, nothing gathered: the first layer's
        gather makes its attention logits in the same pass) where the table lives whole on this device, gathered otherwise."""
        from ..sampler.sampler import _fetch_rows_agreed
        self._wait()
        have = {a.group_name for a in self._fs.get_all_tensor_attrs() if a.attr_name == name and not isinstance(a.group_name, tuple)}
        out, ids = {}, self.n_id
        for t in self.node_types:
            if t not in have:
                continue
            ten = self._fs[t, name, None]
            wm = getattr(ten, "_tensor", None)
            table = getattr(wm, "local_tensor", None)
            if (lazy and table is not None and not getattr(wm, "is_distributed", True) and table.is_cuda and table.dim() == 2
                    and table.dtype == torch.float32 and table.stride(1) == 1):
                out[t] = LazyRows(table, ids[t])
            else:
                out[t] = _fetch_rows_agreed(ten, ids[t])
        return out

    @property
    def x_dict(self):
        return self.node_attr("x")

    @property
    def batch_ptr(self) -> torch.Tensor:
        """int32 [G + 1]: rows of mini-batch b's seeds in the last layer's output of the seed type."""
        return self._rec["calls_seed_seg"]

    @property
    def num_sampled_nodes(self):
        """{node type: vertices per hop (seeds first), summed over the mini-batches}."""
        self._wait()
        return {t: [self._n_level[0][t]] + [self._n_level[k + 1][t] - self._n_level[k][t] for k in range(self.hops)]
                for t in self.node_types}

    @property
    def num_sampled_edges(self):
        """{edge type: edges per hop, summed over the mini-batches}."""
        self._wait()
        n_et = len(self.edge_types)
        return {et: [(self._live[h * n_et + ti] or (0, 0))[1] for h in range(self.hops)] for ti, et in enumerate(self.edge_types)}

    # ---- edges -----------------------------------------------------------------------------------------------------
    def _seg(self, level: int, t: str, as32: bool):
        """Per-batch offsets of the numbering "vertices of type t discovered by the first ``level`` hops" (level == hops: all)."""
        if level == self.hops:
            seg = self._rec["state"][t]["seg"]
            return seg if as32 else seg.long()
        if not as32:
            return self._cseg[level][t]
        key = (level, t)
        if key not in self._cseg32:
            self._cseg32[key] = self._cseg[level][t].to(torch.int32)
        return self._cseg32[key]

    def _numbered(self, ci: int, level_a: int, level_b: int, want_col_b: bool):
        """Rows of call ``ci``'s frontier entries / edge sources in the numberings of two levels — ONE launch
        (``wgamd_call_group_hop_rows_batched``) for both, cached: the output numbering of layer j is the input numbering of layer j + 1."""
        c, (n_f, n_e) = self._rec["calls"][ci], self._live[ci]
        got = self._rows.setdefault(ci, {})
        if level_a in got and (level_b in got or level_b == 0) and (not want_col_b or got.get(level_b, (None, None))[1] is not None):
            return got
        dev = c["offsets"].device
        src_t, _, dst_t = c["et"]
        dst_a = torch.empty(n_f, dtype=torch.int64, device=dev)
        col_a = torch.empty(max(n_e, 1), dtype=torch.int32, device=dev)
        need_b = level_b != 0 or want_col_b
        dst_b = torch.empty(n_f, dtype=torch.int64, device=dev) if need_b else None
        col_b = torch.empty(max(n_e, 1), dtype=torch.int32, device=dev) if want_col_b else None
        state = self._rec["state"]
        L.check(L.lib().wgamd_call_group_hop_rows_batched(
            c["offsets"].data_ptr(), c["f_seg"].data_ptr(), c["f_local0"].data_ptr(), c["row"].data_ptr(),
            n_f, self.n_batches, self._seg(level_a, dst_t, True).data_ptr(), self._seg(level_b, dst_t, False).data_ptr() if need_b else None,
            self._seg(level_a, src_t, True).data_ptr(), self._seg(level_b, src_t, False).data_ptr() if want_col_b else None,
            dst_a.data_ptr(), dst_b.data_ptr() if need_b else None, col_a.data_ptr(), col_b.data_ptr() if want_col_b else None,
            get_stream()), "wgamd_call_group_hop_rows_batched")
        got[level_a] = (dst_a, col_a[:n_e])
        if need_b:
            got[level_b] = (dst_b, col_b[:n_e] if want_col_b else None)
        return got

    def layer_graph(self, layer: int) -> HeteroLayerGraph:
        """The relation hops layer ``layer`` (0 = the one that reads ``x_dict``) of an H-layer model runs over, trimmed
        (``torch_geometric.utils.trim_to_layer`` per node / edge type): it computes rows for the vertices discovered by the
        first ``H - 1 - layer`` hops from the edges of hops ``<= H - 1 - layer``; the LAST layer's output of the seed type is
        exactly the seeds' rows, batch-major."""
        self._wait()
        H = self.hops
        if not 0 <= layer < H:
            raise IndexError(f"layer {layer} of a {H}-hop call group")
        if layer in self._layers:
            return self._layers[layer]
        lvl_in, lvl_out = H - layer, H - 1 - layer
        rels = []
        for ci, (c, lv) in enumerate(zip(self._rec["calls"], self._live)):
            if c is None or lv[0] == 0 or c["hop"] > lvl_out:
                continue            # (a hop with frontier entries but no sampled edge stays: its rows still get act(bias))
            n_f, n_e = lv
            got = self._numbered(ci, lvl_in, lvl_out, want_col_b=c["hop"] < lvl_out)
            dst_in, col_in = got[lvl_in]
            rels.append(RelationHop(c["et"], c["hop"], c["offsets"][:n_f + 1], col_in, dst_in,
                                    None if lvl_out == 0 else got[lvl_out][0], n_e, self._walk.fanout[c["et"]][c["hop"]]))
        lg = HeteroLayerGraph(rels, {t: self._n_level[lvl_out][t] for t in self.node_types}, self.node_types)
        self._layers[layer] = lg
        return lg


class CallGroupIterator:
    """Software-pipelined: the walk of call group g + 1 is enqueued (on its own HIP stream) before group g is handed out, so
    the device never waits for the host's read-back of group g's sizes."""

    def __init__(self, data, core_sampler, input_data, random_state: int, batch_size: int, overlap: bool = True):
        from ..sampler.sampler import HeteroNeighborSampler
        self._fs, self._gs = data
        self._smp, self._B, self._rs = core_sampler, int(batch_size), int(random_state)
        self._hetero = isinstance(core_sampler, HeteroNeighborSampler)
        if input_data.time is not None or not core_sampler.call_groups_ok() or not input_data.node.is_cuda:
            raise NotImplementedError("call_groups(): uniform or positively-weighted sampling with positive fan-outs, no "
                                      "replacement, not disjoint / temporal, seeds on the device")
        self._seed_type = input_data.input_type
        self._seeds = input_data.node.to(torch.int64 if self._hetero else core_sampler.graph.col.dtype)
        self._input_id = input_data.input_id
        n, B = int(self._seeds.shape[0]), self._B
        G = max(1, core_sampler.seeds_per_call(B) // B)
        n_full = n // B
        # (first batch, batches, seeds of the ragged last one or None)
        self._plan = [(b, min(G, n_full - b), None) for b in range(0, n_full, G)]
        if n % B:
            self._plan.append((n_full, 1, n - n_full * B))
        self._stream = torch.cuda.Stream(device=self._seeds.device) if overlap else None
        self._at, self._pending = 0, None

    def __len__(self):
        return len(self._plan)

    def __iter__(self):
        return self

    def _launch_hetero(self, i) -> Optional[HeteroCallGroup]:
        from ..sampler.sampler import _as_i64, hop_seed
        if i >= len(self._plan):
            return None
        b0, g, ragged = self._plan[i]
        smp, B, dev = self._smp, self._B, self._seeds.device
        H, n_et = len(next(iter(smp.fanout.values()))), len(smp.graphs)
        rs = torch.tensor([[_as_i64(hop_seed(self._rs + b0 + j, k)) for j in range(g)] for k in range(H * n_et)], dtype=torch.int64)
        walk = smp._call_group_walk(B if ragged is None else ragged, g)

        def enqueue():
            if ragged is None:
                rec = walk.run(self._seed_type, self._seeds[b0 * B:(b0 + g) * B].contiguous(), rs)
                n_seeds = g * B
            else:   # the last, short mini-batch: a one-batch group over its own seed list
                ids = self._seeds[b0 * B:b0 * B + ragged].contiguous()
                rec = walk.run(None, None, rs, seed_lists={self._seed_type: (
                    ids, torch.tensor([0, ragged], dtype=torch.int32, device=dev), torch.zeros(ragged, dtype=torch.int32, device=dev))})
                n_seeds = ragged
            state, G_ = rec["state"], walk.G
            T = len(walk.ntypes)
            pieces = [state[t]["seg"][G_:G_ + 1] for t in walk.ntypes]
            # vertices per type after k hops as batch-major compact numberings — every (level, type) in ONE cumsum (the walk
            # of a heterogeneous call group is a chain of ~170 small launches; glue that can be one launch is one launch)
            sizes = torch.stack([rec["sizes"][k][t] for k in range(H) for t in walk.ntypes]).to(torch.int64)      # [H T, G]
            cs_all = torch.zeros((H * T, G_ + 1), dtype=torch.int64, device=dev)
            cs_all[:, 1:] = torch.cumsum(sizes, 1)
            cseg = [{t: cs_all[k * T + i] for i, t in enumerate(walk.ntypes)} for k in range(H)]
            totals = cs_all[:, G_].to(torch.int32)
            pieces.append(totals)
            for c in rec["calls"]:
                if c is not None:
                    pieces += [c["f_seg"][G_:G_ + 1], c["counts"][0:1]]       # (frontier entries, sampled edges): views
            rec["calls_seed_seg"] = cseg[0][self._seed_type].to(torch.int32)
            sizes_d = torch.cat([p.to(torch.int32).reshape(-1) for p in pieces])
            sizes_h = torch.empty(sizes_d.shape, dtype=torch.int32, pin_memory=True)
            sizes_h.copy_(sizes_d, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            return rec, sizes_h, ev, n_seeds, cseg

        if self._stream is None:
            rec, sizes_h, ev, n_seeds, cseg = enqueue()
        else:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                rec, sizes_h, ev, n_seeds, cseg = enqueue()
        return HeteroCallGroup(rec, walk, self._fs, self._seed_type, b0, self._input_id[b0 * B:b0 * B + n_seeds], sizes_h, ev,
                               self._stream, cseg)

    def _launch(self, i):
        if self._hetero:
            return self._launch_hetero(i)
        from ..sampler.sampler import hop_seed
        if i >= len(self._plan):
            return None
        b0, g, ragged = self._plan[i]
        smp, B, dev = self._smp, self._B, self._seeds.device
        H = len(smp.fanout)
        rs = [[hop_seed(self._rs + b0 + j, k) for j in range(g)] for k in range(H)]
        walk = smp._call_group_walk(B, g)

        def enqueue():
            if ragged is None:
                res = walk.run(self._seeds[b0 * B:(b0 + g) * B].contiguous(), rs)
                n_seeds = g * B
            else:   # the last, short mini-batch: a one-batch group over a ragged seed list
                ids = torch.zeros(B, dtype=self._seeds.dtype, device=dev)
                ids[:ragged] = self._seeds[b0 * B:b0 * B + ragged]
                seg = torch.tensor([0, ragged], dtype=torch.int32, device=dev)
                res = walk.run(ids, rs, seg, torch.zeros(B, dtype=torch.int32, device=dev))
                n_seeds = ragged
            G_ = res.n_batches
            pieces = [res.node_seg[G_:G_ + 1]] + [res.frontier_seg[k][G_:G_ + 1] for k in range(H + 1)]
            pieces += [res.offsets[k][res.frontier_seg[k][G_:G_ + 1].long()] for k in range(H)]
            sizes_d = torch.cat([p.to(torch.int32).reshape(-1) for p in pieces])
            sizes_h = torch.empty(sizes_d.shape, dtype=torch.int32, pin_memory=True)
            sizes_h.copy_(sizes_d, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            return res, sizes_h, ev, n_seeds

        if self._stream is None:
            res, sizes_h, ev, n_seeds = enqueue()
        else:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                res, sizes_h, ev, n_seeds = enqueue()
        return CallGroup(res, self._fs, smp.graph, b0, self._input_id[b0 * B:b0 * B + n_seeds], sizes_h, ev, self._stream)

    def __next__(self):
        if self._at == 0 and self._pending is None:
            self._pending = self._launch(0)
        cur = self._pending
        if cur is None:
            raise StopIteration
        self._at += 1
        self._pending = self._launch(self._at)     # group g + 1 is on the device before the host blocks on group g
        cur._wait()
        return cur
