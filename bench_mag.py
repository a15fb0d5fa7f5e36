"""bench.py --workload mag — BASELINE configs[4]: ogbn-mag-like heterogeneous 2-hop sampling + GATConv, one MI355X.

One "step" = `--groups-per-step` CALL GROUPS of `--call-group` mini-batches of 1024 paper seeds through the whole path:
  heterogeneous 2-hop walk over 6 edge types, fan-out [25, 10] each (HeteroPygWalk: one launch sequence per hop and edge
  type for the whole call group, no host sync)                                   [reference: cugraph_pyg NeighborLoader on
  a heterogeneous GraphStore, examples/mag_lp_mnmg.py:141; sampler/distributed_sampler.py:877-908]
  -> feature gather for every node type (fp32 [n_t, 128]; paper = the dataset's features, the other types = embedding
  tables, as examples/mag_lp_mnmg.py:120-136 does with learn_embeddings)
  -> 2 layers of HeteroConv{edge type: GATConv(in, 64, heads = 4)}, aggr = "sum", ReLU       [GATConv as the reference
  builds it: pylibwholegraph/torch/gnn_model.py:45-59; semantics SURVEY.md §8 row a18]
How the layers are computed (same outputs for the seeds as PyG's formulation, up to fp32 reassociation):
  * TRIMMED: layer 1 produces rows only for the vertices the seeds can see through layer 2 (those discovered by hops 0-1,
    kept in a compact per-type array), layer 2 only for the seeds (what torch_geometric.utils.trim_to_layer does);
  * AGGREGATE-FIRST: the attention-weighted sum is linear, so every (hop, edge type) is ONE launch of
    wgamd_gat_aggregate_heads_f32 over the UNTRANSFORMED source rows (10^5 .. 10^6 edges, rows = the hop's frontier entries),
    and the per-head weights are applied afterwards to the few destination rows (H small GEMMs) — the lin GEMM over every
    source row, 10-20x more rows, never runs.  alpha's inputs are x @ fold(W, att) ([n, 128] x [128, 4 per relation]).

`value` = sampled edges / wall time; `roofline` = the dominant GAT launch by SURVEY §8(d)'s single-pass byte count with the row
widths of this formulation, E (4F + 4H + 4) + N_dst (4HF + 4H + 8); `cpu_baseline` = the same composition (sampling + gather + both GAT layers) on the C
oracle + torch CPU GEMMs, one mini-batch at a time, bounded.
"""
import json
import os
import sys
import time

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0
MAG_NODES = {"paper": 736_389, "author": 1_134_649, "institution": 8_740, "field_of_study": 59_965}
MAG_RELS = {("author", "writes", "paper"): 7_145_660, ("paper", "cites", "paper"): 5_416_271,
            ("paper", "has_topic", "field_of_study"): 7_505_078, ("author", "affiliated_with", "institution"): 1_043_998,
            ("paper", "rev_writes", "author"): 7_145_660, ("field_of_study", "rev_has_topic", "paper"): 7_505_078}
F_IN, HEADS, CH = 128, 4, 64
HC = HEADS * CH


def build_mag_like(dev, nodes=None, rels=None, seed=11):
    """Synthetic ogbn-mag-like GraphStore (skewed endpoints: squared uniforms, so hubs exist) -> (graphs, num_nodes)."""
    from cugraph_pyg_amd.data import GraphStore
    nodes, rels = nodes or MAG_NODES, rels or MAG_RELS
    g = torch.Generator(device=dev).manual_seed(seed)
    gs = GraphStore()
    for (s_, r_, d_), m in rels.items():
        src = (torch.rand(m, generator=g, device=dev) ** 2 * nodes[s_]).long().clamp_(max=nodes[s_] - 1)
        dst = (torch.rand(m, generator=g, device=dev) ** 2 * nodes[d_]).long().clamp_(max=nodes[d_] - 1)
        gs[(s_, r_, d_), "coo", False, (nodes[s_], nodes[d_])] = torch.stack([src, dst])
    return gs._hetero_graphs, dict(nodes)


def make_params(etypes, ntypes, dev, seed=3):
    """Per layer and edge type: GATConv lin weight [in, H*C] (shared by both ends, as PyG's GATConv with one `lin`),
    att_src / att_dst [H, C]; per layer and node type a bias [H*C].  Uniform(-a, a), fixed seed."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    params = []
    for layer, fin in enumerate((F_IN, HC)):
        rel = {}
        for et in etypes:
            w = (torch.rand((fin, HC), generator=g) - 0.5) * (2.0 / np.sqrt(fin))
            att_s, att_d = (torch.rand((HEADS, CH), generator=g) - 0.5) * 0.5, (torch.rand((HEADS, CH), generator=g) - 0.5) * 0.5
            # alpha_src = ((x W).view(H, C) * att).sum(-1) = x (W . att): the [in, H] matrices are folded once
            v_s = (w.view(fin, HEADS, CH) * att_s).sum(-1)
            v_d = (w.view(fin, HEADS, CH) * att_d).sum(-1)
            rel[et] = {k: v.to(dev).contiguous() for k, v in dict(w=w, v_src=v_s, v_dst=v_d).items()}
        bias = {t: ((torch.rand(HC, generator=g) - 0.5) * 0.1).to(dev) for t in ntypes}
        params.append(dict(rel=rel, bias=bias))
    return params


class MagPipeline:
    """The measured path for one rank: call-group walk, per-type feature gather, two HeteroConv(GATConv) layers."""

    def __init__(self, graphs, num_nodes, tables, params, dev, B, G, fanout=(25, 10)):
        from wholegraph_amd import fused, nn
        self.nn, self.dev, self.B, self.G = nn, dev, B, G
        self.fused_tail = os.environ.get("WGAMD_GAT_TRANSFORM", "bf16x3") != "library"
        self.fused_layer = os.environ.get("WGAMD_GAT_LAYER", "fused") != "split"
        # the one-kernel relation keeps 10 neighbours of a row in registers and continues longer rows online, one neighbour
        # at a time: hops with a larger fan-out take the two-kernel path unless this says otherwise (measurement switch; the
        # fan-out-25 hop through the one-kernel relation: 0.77 ms instead of 0.39 + 0.16 per call group, 2.03 vs 2.06 G edges/s)
        self.fused_max_fanout = int(os.environ.get("WGAMD_GAT_FUSED_MAX_FANOUT", "10"))
        self.etypes = sorted(graphs)
        self.ntypes = sorted({t for et in self.etypes for t in (et[0], et[2])})
        self.fanout = {et: list(fanout) for et in self.etypes}
        self.hops = len(fanout)
        self.walk = fused.HeteroPygWalk(graphs, B, self.fanout, G, num_nodes=num_nodes, pad_unique=False)
        self.tables, self.params = tables, params
        self.walk_stream = torch.cuda.Stream(device=dev)
        self._rs = None

    # ---- walk -------------------------------------------------------------------------------------------------
    def sample(self, seeds, group_id):
        """Enqueue the walk of one call group on its own stream + ONE async D2H of every size the forward pass needs."""
        n_et = len(self.etypes)
        if self._rs is None:
            from cugraph_pyg_amd.sampler.sampler import _as_i64, hop_seed
            self._rs_base = torch.tensor([[_as_i64(hop_seed(7 + j, k)) for j in range(self.G)] for k in range(self.hops * n_et)],
                                         dtype=torch.int64, device=self.dev)
            self._rs = True
        self.walk_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.walk_stream):
            rec = self.walk.run("paper", seeds, self._rs_base + group_id * self.G)
            G = self.G
            pieces = [rec["state"][t]["seg"][G:G + 1] for t in self.ntypes]
            # vertices per type after hop 1 (the rows layer 1 has to produce), as a compact batch-major numbering
            rec["cseg"] = {}
            for t in self.ntypes:
                cs = torch.zeros(G + 1, dtype=torch.int64, device=self.dev)
                cs[1:] = torch.cumsum(rec["sizes"][1][t].long(), 0)
                rec["cseg"][t] = cs
                pieces.append(cs[G:G + 1])
            for c in rec["calls"]:
                if c is not None:
                    n_f = c["f_seg"][G:G + 1]
                    pieces += [n_f, c["offsets"][n_f.long()]]
            sizes_d = torch.cat([p.to(torch.int32) for p in pieces])
            sizes_h = torch.empty(sizes_d.shape, dtype=torch.int32, pin_memory=True)
            sizes_h.copy_(sizes_d, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.walk_stream)
        return rec, sizes_h, ev

    def _sizes(self, rec, sizes_h):
        it = iter(sizes_h.tolist())
        n_nodes = {t: next(it) for t in self.ntypes}
        self._n_compact = {t: next(it) for t in self.ntypes}
        live = []
        for c in rec["calls"]:
            live.append(None if c is None else (next(it), next(it)))      # (frontier entries, edges)
        return n_nodes, live

    # ---- forward ----------------------------------------------------------------------------------------------
    def _terms_keys(self, t):
        """Relation ends whose attention logit reads node type t, in the column order of ``_terms_matrix``."""
        keys = []
        for et in self.etypes:
            if et[0] == t:
                keys.append(("src", et))
            if et[2] == t:
                keys.append(("dst", et))
        return keys

    def _terms_matrix(self, layer, t):
        """[in, HEADS * relation ends of t]: the folded attention vectors of every relation end of node type t, side by side."""
        cache = self.__dict__.setdefault("_terms_cache", {})
        if (layer, t) not in cache:
            mats = [self.params[layer]["rel"][et]["v_src" if end == "src" else "v_dst"] for end, et in self._terms_keys(t)]
            cache[(layer, t)] = torch.cat(mats, 1).contiguous() if mats else None
        return cache[(layer, t)]

    def forward(self, rec, sizes_h, ev, timers=None):
        nn, G = self.nn, self.G
        ev.synchronize()
        n_nodes, live = self._sizes(rec, sizes_h)
        main = torch.cuda.current_stream()
        state = rec["state"]
        for t in self.ntypes:
            for k in ("nodes", "seg"):
                state[t][k].record_stream(main)

        def stage(name, fn):
            if timers is None:
                return fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = fn()
            e.record()
            timers.append((name, s, e))
            return out

        # feature fetch: one row gather per node type for the whole call group
        from wholegraph_amd.tensor import local_gather
        # Layer 1's attention logits need x_t @ [fold(W_r, att_src) | fold(W_r, att_dst) ...] for every gathered row: folded into
        # the gather (wgamd_gather_terms_f32: the rows pass through registers once and feed an exact-fp32 MFMA) instead of a
        # second pass over x (library GEMM with N = 8..20: 1.9 ms per call group next to the 1.4 ms gather).
        x, terms1 = {}, {}
        for t in self.ntypes:
            ids = state[t]["nodes"][:n_nodes[t]]
            v_t = self._terms_matrix(0, t)
            buf = torch.empty((n_nodes[t], F_IN), dtype=torch.float32, device=self.dev)
            if v_t is not None and n_nodes[t] > 0 and nn.gather_terms_supported(F_IN, v_t.shape[1]):
                x[t], terms1[t] = stage("gather+attn_terms1", lambda: nn.gather_with_terms(self.tables[t], ids, v_t, out=buf,
                                                                                           heads=HEADS if HEADS == 4 else 0))
            else:
                x[t] = stage("gather", lambda: local_gather(self.tables[t], ids, buf))

        # per (hop, edge type): where the hop's rows sit in the destination type's node list (full numbering for the
        # attention terms of layer 1, compact numbering for the layer-1 output), and the source rows of its edges (local
        # ids are per mini-batch; the lists are batch-major)
        n_c = self._n_compact
        cseg = rec["cseg"]

        def prep():
            from wholegraph_amd import _lib as L
            from wholegraph_amd.env import get_stream
            out = []
            for c, lv in zip(rec["calls"], live):
                if c is None or lv[0] == 0:
                    out.append(None)
                    continue
                n_f, n_e = lv      # (a hop with frontier entries but no sampled edge stays: its rows still get relu(bias))
                src_t, _, dst_t = c["et"]
                for k in ("offsets", "row", "f_batch", "f_seg", "f_local0"):
                    c[k].record_stream(main)
                first_hop = c["hop"] == 0
                dst_full = torch.empty(n_f, dtype=torch.int64, device=self.dev)
                dst_c = torch.empty(n_f, dtype=torch.int64, device=self.dev)
                col_full = torch.empty(max(n_e, 1), dtype=torch.int32, device=self.dev)
                col_c = torch.empty(max(n_e, 1), dtype=torch.int32, device=self.dev) if first_hop else None
                # one launch per hop and edge type (wgamd_call_group_hop_rows) instead of a dozen torch index ops
                L.check(L.lib().wgamd_call_group_hop_rows(
                    c["offsets"].data_ptr(), c["f_batch"].data_ptr(), c["f_seg"].data_ptr(), c["f_local0"].data_ptr(),
                    c["row"].data_ptr(), n_f, state[dst_t]["seg"].data_ptr(), cseg[dst_t].data_ptr(), state[src_t]["seg"].data_ptr(),
                    cseg[src_t].data_ptr() if first_hop else None, dst_full.data_ptr(), dst_c.data_ptr(), col_full.data_ptr(),
                    col_c.data_ptr() if first_hop else None, get_stream()), "wgamd_call_group_hop_rows")
                out.append(dict(et=c["et"], hop=c["hop"], off=c["offsets"][:n_f + 1], n_f=n_f, n_e=n_e, dst_full=dst_full,
                                dst_c=dst_c, col_full=col_full, col_c=col_c))
            return out
        calls = [c for c in stage("index_prep", prep) if c is not None]
        edges = sum(lv[1] for lv in live if lv is not None)

        def attention_terms(layer, xs, p, ready=None):
            """alpha's inputs for every relation: x_t @ [fold(W_r, att_src) | fold(W_r, att_dst) ...] — ONE pass over x_t
            (``ready[t]``: the product already made by the gather)."""
            a_src, a_dst = {}, {}
            for t in self.ntypes:
                if xs[t].shape[0] == 0:
                    continue
                keys = [(a_src if end == "src" else a_dst, et) for end, et in self._terms_keys(t)]
                if not keys:
                    continue
                # (a stand-alone narrow-matmul kernel — 64-row LDS tiles, K <= 32 — was built and measured: no faster than the
                #  library GEMM at F = 128, slower at F = 256; taken out again)
                if ready is not None and t in ready and ready[t].dim() == 3:     # [relation end][n][H] slabs from the gather
                    for k, (dst, et) in enumerate(keys):
                        dst[et] = ready[t][k]
                    continue
                vt = self._terms_matrix(layer, t)
                if not (ready is not None and t in ready) and xs[t].stride(1) == 1 and xs[t].stride(0) % 4 == 0 \
                        and xs[t].data_ptr() % 16 == 0 and nn.gather_terms_supported(int(xs[t].shape[1]), int(vt.shape[1])):
                    # hidden state of the previous layer: one streaming pass, slabs come out of the kernel (no library GEMM +
                    # transposing copy)
                    slabs = nn.rows_terms(xs[t], vt, heads=HEADS)
                    for k, (dst, et) in enumerate(keys):
                        dst[et] = slabs[k]
                    continue
                both = ready[t] if (ready is not None and t in ready) else xs[t] @ vt
                # one [relation end][n][H] copy per node type instead of one slice copy per relation end
                slabs = both.view(both.shape[0], len(keys), HEADS).permute(1, 0, 2).contiguous()
                for k, (dst, et) in enumerate(keys):
                    dst[et] = slabs[k]
            return a_src, a_dst

        def hetero_layer(layer, xs, hop_set, col_key, dst_key, n_out, launches):
            """One HeteroConv{GATConv} layer for the frontier rows of the hops in ``hop_set``; returns {type: compact rows}."""
            p = self.params[layer]
            a_src, a_dst = stage("attn_terms%d" % (layer + 1), lambda: attention_terms(layer, xs, p, terms1 if layer == 0 else None))
            # every compact row is a frontier entry of exactly one hop of its type, so the index_copy below writes all of them
            out = {t: torch.empty((n_out[t], HC), dtype=torch.float32, device=self.dev) for t in self.ntypes if n_out[t] > 0}
            for h in hop_set:
                for dt in self.ntypes:
                    mine = [c for c in calls if c["hop"] == h and c["et"][2] == dt]
                    if not mine:
                        continue
                    # HeteroConv's sum over the relations into `acc`: the first relation with edges WRITES it (beta = 0)
                    acc = torch.empty((mine[0]["n_f"], HC), dtype=torch.float32, device=self.dev)
                    live_rel = [c for c in mine if c["n_e"] > 0]   # (nothing sampled for a relation: it adds nothing to the sum)
                    # one-pass tail (wgamd_gat_transform_heads_bf16x3): the LAST relation's transform also adds the bias, applies
                    # the ReLU and places the hop's rows — no separate bias / ReLU pass over the layer output
                    one_pass = self.fused_tail and bool(live_rel) and \
                        nn.gat_transform_supported(xs[live_rel[0]["et"][0]].shape[1], HEADS, HC // HEADS)
                    if one_pass and layer == 1:
                        out[dt] = torch.empty((mine[0]["n_f"], HC), dtype=torch.float32, device=self.dev)
                    for j, c in enumerate(live_rel):
                        et = c["et"]
                        last = j == len(live_rel) - 1
                        # deep hop (fan-out <= 10) of layer 1: aggregation + dense tail as ONE kernel, the aggregate stays in LDS
                        if one_pass and self.fused_layer and self.fanout[et][h] <= self.fused_max_fanout and \
                                nn.gat_layer_fused_supported(xs[et[0]].shape[1], HEADS, HC // HEADS):
                            stage("gat%d+transform:%s hop %d (%d rows, %d edges)" % (layer + 1, et[1], h + 1, c["n_f"], c["n_e"]),
                                  lambda: nn.gat_layer_fused(
                                      c["off"], c[col_key], xs[et[0]], a_src[et], a_dst[et], p["rel"][et]["w"], HEADS,
                                      dst_rows=c[dst_key], acc_in=acc if j > 0 else None, bias=p["bias"][dt] if last else None,
                                      relu=last, out_rows=mine[0]["dst_c"] if (last and layer == 0) else None,
                                      out=out[dt] if last else acc))
                            if launches is not None:
                                launches.append((et, h, c["n_f"], c["n_e"], xs[et[0]].shape[1]))
                            continue
                        agg = stage("gat%d:%s hop %d (%d rows, %d edges)" % (layer + 1, et[1], h + 1, c["n_f"], c["n_e"]),
                                    lambda: nn.gat_aggregate_heads(c["off"], c[col_key], xs[et[0]], a_src[et], a_dst[et], HEADS,
                                                                   dst_rows=c[dst_key]))
                        if one_pass:
                            stage("transform%d" % (layer + 1), lambda: nn.gat_transform_heads_fused(
                                agg, p["rel"][et]["w"], HEADS, acc_in=acc if j > 0 else None,
                                bias=p["bias"][dt] if last else None, relu=last,
                                out_rows=mine[0]["dst_c"] if (last and layer == 0) else None,
                                out=out[dt] if last else acc))
                        else:
                            stage("transform%d" % (layer + 1), lambda: nn.gat_transform_heads(agg, p["rel"][et]["w"], HEADS, out=acc,
                                                                                              overwrite=j == 0))
                        if launches is not None:
                            launches.append((et, h, c["n_f"], c["n_e"], xs[et[0]].shape[1]))
                    if one_pass:
                        continue
                    if not live_rel:
                        acc.zero_()       # no relation of this type sampled an edge in this hop: relu(bias) rows
                    # bias + ReLU (+ the placement of the hop's rows in the compact list of layer 1) in one pass
                    if layer == 0:
                        stage("bias_relu", lambda: nn.bias_act_rows(acc, p["bias"][dt], True, mine[0]["dst_c"], out[dt]))
                    else:
                        out[dt] = stage("bias_relu", lambda: nn.bias_act_rows(acc, p["bias"][dt], True))
            return out

        launches = []
        # layer 1: rows for every vertex discovered by hops 0-1 (the frontiers of hops 1 and 2), sources = all vertices
        y1 = hetero_layer(0, x, (0, 1), "col_full", "dst_full", n_c, launches)
        for t in self.ntypes:
            y1.setdefault(t, torch.empty((0, HC), dtype=torch.float32, device=self.dev))
        # layer 2: rows for the seeds only (hop-1 frontier), sources = the compact layer-1 rows
        y2 = hetero_layer(1, y1, (0,), "col_c", "dst_c", {t: (G * self.B if t == "paper" else 0) for t in self.ntypes}, launches)
        return y2["paper"], edges, n_nodes, launches


def cpu_port_batch(hg, tables_h, params_h, seeds, fanout, hops, etypes, ntypes, batch_seed, fp64=False):
    """One mini-batch of the same path on the host — the same trimmed, aggregate-first computation: C oracle (sampling,
    renumbering, wgo_gat_aggregate_heads with OpenMP) + torch CPU GEMMs.  Returns (seed-row outputs [B, HC], sampled edges).
    ``fp64``: every floating-point step in float64 (features, attention terms, softmax, aggregation, transforms, bias) —
    the reference the pipeline's 1e-5 parity is asserted against (the fp32 port is what `cpu_baseline` times)."""
    import oracle
    from cugraph_pyg_amd.sampler.sampler import hop_seed
    node = {t: np.zeros(0, np.int64) for t in ntypes}
    node["paper"] = seeds.astype(np.int64)
    fstart = {t: 0 for t in ntypes}
    calls, edges = [], 0
    for h in range(hops):
        begin = {t: len(node[t]) for t in ntypes}
        for ti, et in enumerate(etypes):
            frontier = node[et[2]][fstart[et[2]]:begin[et[2]]]
            if len(frontier) == 0:
                continue
            rp_h, col_h = hg[et]
            off, nbr, _, _ = oracle.unweighted_sample(rp_h, col_h, frontier, fanout[et][h], hop_seed(batch_seed, h * len(etypes) + ti))
            node[et[0]], mp = oracle.append_unique(node[et[0]], nbr.astype(np.int64))
            calls.append((et, h, fstart[et[2]], off.astype(np.int32), mp.astype(np.int32)))
            edges += int(nbr.size)
        for t in ntypes:
            fstart[t] = begin[t]
        if h == 0:
            size1 = {t: len(node[t]) for t in ntypes}
    x = {t: oracle.gather_rows(tables_h[t], node[t]) for t in ntypes}
    n1 = {t: size1[t] for t in ntypes}                                    # vertices after hop 1 = rows layer 1 produces
    ft = np.float64 if fp64 else np.float32
    if fp64:
        x = {t: v.astype(np.float64) for t, v in x.items()}
    tt = lambda v: v.double() if fp64 else v      # noqa: E731
    aggregate = oracle.gat_aggregate_heads_f64 if fp64 else oracle.gat_aggregate_heads

    def layer(p, xs, hop_set, n_out):
        a = {}
        for et in etypes:
            a[et] = ((torch.from_numpy(xs[et[0]]) @ tt(p["rel"][et]["v_src"])).numpy(), (torch.from_numpy(xs[et[2]]) @ tt(p["rel"][et]["v_dst"])).numpy())
        out = {t: np.zeros((n_out[t], HC), ft) for t in ntypes}
        for et, hop, first, off, mp in calls:
            if hop not in hop_set:
                continue
            n_f = off.size - 1
            rows = np.arange(first, first + n_f, dtype=np.int64)
            agg = aggregate(off, mp, xs[et[0]], a[et][0], a[et][1], dst_rows=rows)      # [n_f, H, F]
            w = tt(p["rel"][et]["w"])
            F_ = agg.shape[2]
            res = torch.bmm(torch.from_numpy(agg).permute(1, 0, 2), w.view(F_, HEADS, CH).permute(1, 0, 2))   # [H, n_f, C]
            out[et[2]][first:first + n_f] += res.permute(1, 0, 2).reshape(n_f, HC).numpy()
        return {t: np.maximum(out[t] + tt(p["bias"][t]).numpy(), 0) for t in ntypes}

    y1 = layer(params_h[0], x, (0, 1), n1)           # layer 1: the vertices of hops 0-1 (local ids = a prefix of every list)
    y2 = layer(params_h[1], y1, (0,), {t: (len(seeds) if t == "paper" else 0) for t in ntypes})
    return y2["paper"], edges


def cpu_baseline(graphs, tables, params, seeds_h, B, fanout, hops, etypes, ntypes, budget_s):
    import oracle
    from bench import usable_cpus
    oracle.build()
    threads = usable_cpus()
    oracle.set_num_threads(threads)
    torch.set_num_threads(threads)
    hg = {et: (gr.row_ptr.cpu().numpy(), gr.col.cpu().numpy()) for et, gr in graphs.items()}
    tables_h = {t: v.cpu().numpy() for t, v in tables.items()}
    params_h = [dict(rel={et: {k: v.cpu() for k, v in w.items()} for et, w in p["rel"].items()},
                     bias={t: b.cpu() for t, b in p["bias"].items()}) for p in params]
    t0, edges, nb = time.perf_counter(), 0, 0
    while time.perf_counter() - t0 < budget_s and (nb + 1) * B <= len(seeds_h):
        _, e = cpu_port_batch(hg, tables_h, params_h, seeds_h[nb * B:(nb + 1) * B], fanout, hops, etypes, ntypes, 7 + nb)
        edges += e
        nb += 1
    dt = time.perf_counter() - t0
    return {"value": edges / dt, "unit": "sampled-edges/s", "cores": threads, "kind": "port",
            "sample": f"{nb} mini-batches of {B} paper seeds: 2-hop [25,10] x 6 edge types + feature gather + 2 HeteroConv(GATConv "
                      f"4x64) layers on the C oracle (OpenMP) + torch CPU GEMMs, {dt:.1f} s"}


def main(args):
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    assert int(os.environ.get("WORLD_SIZE", "1")) == 1, "--workload mag is the single-GPU configuration (BASELINE configs[4])"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    graphs, num_nodes = build_mag_like(dev)
    etypes = sorted(graphs)
    ntypes = sorted(num_nodes)
    g = torch.Generator(device=dev).manual_seed(5)
    tables = {t: torch.rand((num_nodes[t], F_IN), generator=g, device=dev) * 2 - 1 for t in ntypes}
    params = make_params(etypes, ntypes, dev)
    # call groups of 64 mini-batches (measured: 32 -> 1.31, 64 -> 1.40, 96 -> 1.39 G edges/s; the walk's ~120 launches per group
    # are what a larger group amortises)
    B, G, gps = 1024, args.call_group if args.call_group > 0 else 64, args.groups_per_step
    pipe = MagPipeline(graphs, num_nodes, tables, params, dev, B, G)
    groups = args.steps * gps
    warm = max(args.warmup * gps, 2)
    distinct = min(groups + warm, 16)
    gs_ = torch.Generator(device=dev).manual_seed(7)
    reps = -(-distinct * G * B // num_nodes["paper"])
    order = torch.cat([torch.randperm(num_nodes["paper"], generator=gs_, device=dev) for _ in range(reps)])
    batches = order[:distinct * G * B].view(distinct, G * B).contiguous()

    def run(first, last, timers=None):
        edges = 0
        pending = pipe.sample(batches[first % distinct], first)
        for gi in range(first, last):
            nxt = pipe.sample(batches[(gi + 1) % distinct], gi + 1) if gi + 1 < last else None
            _, e, _, _ = pipe.forward(*pending, timers=timers)
            edges += e
            pending = nxt
        return edges

    run(0, warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    edges = run(warm, warm + groups)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    # per-stage HIP-event pass (outside the timed region): one call group at a time, the walk alone on its stream
    acc, n_probe, launches, nn_sizes = {}, min(groups, 6), None, None
    per_shape = {}
    for gi in range(warm, warm + n_probe):
        timers = []
        torch.cuda.synchronize()
        w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0.record(pipe.walk_stream)
        pend = pipe.sample(batches[gi % distinct], gi)
        w1.record(pipe.walk_stream)
        _, _, nn_sizes, launches = pipe.forward(*pend, timers=timers)
        torch.cuda.synchronize()
        timers.append(("walk(2 hops x 6 edge types)", w0, w1))
        for name, a, b in timers:
            key = name.split(":")[0]
            acc[key] = acc.get(key, 0.0) + a.elapsed_time(b)
            if ":" in name:
                acc[name] = acc.get(name, 0.0) + a.elapsed_time(b)
    stage_ms = {k: v / n_probe for k, v in acc.items() if ":" not in k}
    # dominant GAT launch over the probed groups (shapes differ by a per cent between groups: the stage name carries the shape)
    import re
    gat = {k: v for k, v in acc.items() if k.startswith("gat") and ":" in k}
    roofline = None
    if gat:
        name = max(gat, key=lambda k: gat[k])
        n_f, n_e = (int(v) for v in re.search(r"\((\d+) rows, (\d+) edges\)", name).groups())
        F_ = F_IN if name.startswith("gat1") else HC
        ms = gat[name]       # this exact shape occurred once (its own group)
        one_kernel = "+transform" in name.split(":")[0]
        # one-kernel relation (wg_gat_fused.hip): the [N_dst, H F] aggregate never reaches HBM — what a destination row costs
        # is its OUTPUT row (H C floats) instead; the running HeteroConv sum a later relation reads back is not counted
        by = n_e * (4 * F_ + 4 * HEADS + 4) + n_f * ((4 * HC if one_kernel else 4 * HEADS * F_) + 4 * HEADS + 8)
        roofline = {"bound": "hbm", "kernel": "gat_layer_fused_kernel" if one_kernel else "gat_aggregate_heads_kernel", "stage": name,
                    "achieved": round(by / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                    "algorithmic_bytes_per_launch": int(by), "avg_launch_ms": round(ms, 5),
                    "bytes_formula": ("SURVEY §8(d) GAT, single pass, aggregate-first, aggregation + dense tail in one kernel: "
                                      "E (4F + 4H + 4) + N_dst (4HC + 4H + 8) with F = %d source floats per edge, H = 4 heads, C = 64"
                                      if one_kernel else
                                      "SURVEY §8(d) GAT, single pass, aggregate-first row widths: E (4F + 4H + 4) + N_dst (4HF + 4H "
                                      "+ 8) with F = %d source floats per edge, H = 4 heads") % F_,
                    "timing": "HIP events around the launch on the launch stream (one launch per hop and edge type per call group)"}
    if roofline is not None:
        # the same kernel over ALL its launches of the probed groups (sum of algorithmic bytes / sum of launch times): the
        # dominant launch flips between two relations whose launches last within 3 % of each other but move different bytes
        same, tot_by, tot_ms = roofline["stage"].split(":")[0], 0.0, 0.0
        for k, ms_k in gat.items():
            if k.split(":")[0] != same:
                continue
            r_, e_ = (int(v) for v in re.search(r"\((\d+) rows, (\d+) edges\)", k).groups())
            f_k = F_IN if k.startswith("gat1") else HC
            tot_by += e_ * (4 * f_k + 4 * HEADS + 4) + r_ * ((4 * HC if "+transform" in same else 4 * HEADS * f_k) + 4 * HEADS + 8)
            tot_ms += ms_k
        if tot_ms > 0:
            roofline["kernel_all_launches"] = {"stage": same, "launches": sum(1 for k in gat if k.split(":")[0] == same),
                                               "algorithmic_bytes": int(tot_by), "ms": round(tot_ms, 4),
                                               "frac": round(tot_by / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
    if roofline is not None and G == 64 and args.call_group <= 0:
        # HBM traffic and average duration of that launch shape from the committed profile of this command
        # (profiles/rNN/pmc_traffic_mag.json: the `#large` cluster = the largest launch of every call group; mag_kernel_stats.csv)
        from bench import load_pmc, load_profiled_avg
        hit = load_pmc(roofline["kernel"], want_void=False, workload="mag")
        if hit and hit.get("max_bytes"):
            # eleven launch shapes per call group: the dominant launch is the LARGEST single launch of the kernel in the PMC passes
            roofline["traffic"] = hit["max_bytes"]
            roofline["traffic_over_algorithmic"] = round(hit["max_bytes"] / roofline["algorithmic_bytes_per_launch"], 3)
            roofline["traffic_source"] = hit["source"] + " kernel " + hit["kernel"] + " (largest launch)"
        prof = load_profiled_avg(roofline["kernel"], "mag")
        if prof:   # (the summary averages ALL launches of the kernel, 11 shapes per call group: the max is the dominant launch)
            roofline["profiled_source"] = "%s (%d launches of all shapes, avg %.1f us)" % (prof["source"], prof["calls"], prof["avg_ns"] * 1e-3)
    cpu = None
    if not args.no_cpu_baseline:
        cpu = cpu_baseline(graphs, tables, params, order[:min(order.numel(), 256 * B)].cpu().numpy(), B, pipe.fanout, pipe.hops,
                           etypes, ntypes, args.cpu_budget)
    out = {"metric": "sampled-edges/sec (hetero 2-hop sample+renumber + feature gather + 2-layer HeteroConv(GATConv 4x64) fwd), "
                     "ogbn-mag-like fan-out [25, 10] x 6 edge types",
           "value": edges / dt, "unit": "sampled-edges/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "int64 ids + f32 features (edge softmax + aggregation: f32 HIP kernels; GATConv per-head lin: bf16x3-split MFMA, "
                    "f32 accumulate — the one-kernel relation / wgamd_gat_transform_heads_bf16x3)",
           "data": "synthetic",
           "config": {"workload": "ogbn-mag-like hetero (BASELINE configs[4]): 4 node types (736,389 / 1,134,649 / 8,740 / 59,965), "
                                  "6 edge types ~35.8 M edges, feat fp32 [n_t, 128] per type, batch 1024 paper seeds, 2-hop fan-out "
                                  "[25,10] per edge type, 2 x HeteroConv{GATConv(., 64, heads=4)} sum + ReLU, step = %d call groups "
                                  "of %d mini-batches" % (gps, G),
                      "parallelism": "1 GPU"},
           "call_group": G, "batches_per_step": G * gps, "timed_region_ms": round(dt * 1e3, 2), "timed_call_groups": groups,
           "ms_per_batch": dt / (groups * G) * 1e3, "edges_per_batch": edges / (groups * G),
           "nodes_per_call_group": nn_sizes, "stage_ms_per_call_group": {k: round(v, 4) for k, v in stage_ms.items()},
           "gat_launches": [{"edge_type": "%s-%s-%s" % et, "hop": h + 1, "rows": n_f, "edges": n_e, "src_row_floats": f_}
                            for et, h, n_f, n_e, f_ in (launches or [])],
           "roofline": roofline, "cpu_baseline": cpu}
    if cpu is not None:
        out["gpu_over_cpu"] = round(out["value"] / cpu["value"], 2)
    print(json.dumps(out), flush=True)
