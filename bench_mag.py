"""bench.py --workload mag — BASELINE configs[4]: ogbn-mag-like heterogeneous 2-hop sampling + GATConv, one MI355X.

One "step" = `--groups-per-step` CALL GROUPS of `--call-group` mini-batches of 1024 paper seeds through the whole path:
  heterogeneous 2-hop walk over the 8 edge types (4 + reverses), fan-out [25, 10] each (wholegraph_amd.fused.HeteroPygWalk behind the loader: one
  launch sequence per hop and edge type for the whole call group, no host sync)                                   [reference: cugraph_pyg NeighborLoader on
  a heterogeneous GraphStore, examples/mag_lp_mnmg.py:141; sampler/distributed_sampler.py:877-908]
  -> feature gather for every node type (fp32 [n_t, 128]; paper = the dataset's features, the other types = embedding
  tables, as examples/mag_lp_mnmg.py:120-136 does with learn_embeddings)
  -> 2 layers of HeteroConv{edge type: GATConv(in, 64, heads = 4)}, aggr = "sum", ReLU       [GATConv as the reference
  builds it: pylibwholegraph/torch/gnn_model.py:45-59; semantics SURVEY.md §8 row a18]
The whole path runs through the PACKAGE: GraphStore + FeatureStore -> cugraph_pyg_amd.loader.NeighborLoader.call_groups()
(HeteroCallGroup: lazy x_dict, trimmed per-layer relation hops) -> 2 x wholegraph_amd.nn.HeteroConv{GATConv}; this file holds
the workload, the timing and the CPU port only.
How the layers are computed (wholegraph_amd.nn.HeteroConv._forward_layer; same outputs for the seeds as PyG's formulation, up
to fp32 reassociation):
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
# ogbn-mag's four relations and their reverses (BASELINE.md S4 / SURVEY §8(d): "4 (+reverse) edge types", ~42 M directed edges).
# Rounds 3-5 ran six of the eight (no rev_cites / rev_affiliated_with, 35.8 M edges): MAG_RELS_R5, `--mag-rels r5`.
MAG_FWD = {("author", "writes", "paper"): 7_145_660, ("paper", "cites", "paper"): 5_416_271,
           ("paper", "has_topic", "field_of_study"): 7_505_078, ("author", "affiliated_with", "institution"): 1_043_998}
MAG_RELS = dict(MAG_FWD)
MAG_RELS.update({(d_, "rev_" + r_, s_): m for (s_, r_, d_), m in MAG_FWD.items()})
MAG_RELS_R5 = {k: v for k, v in MAG_RELS.items() if k[1] not in ("rev_cites", "rev_affiliated_with")}
F_IN, HEADS, CH = 128, 4, 64
HC = HEADS * CH


def build_mag_like(dev, nodes=None, rels=None, seed=11):
    """Synthetic ogbn-mag-like GraphStore (skewed endpoints: squared uniforms, so hubs exist) -> (graphs, num_nodes)."""
    from cugraph_pyg_amd.data import GraphStore
    nodes, rels = nodes or MAG_NODES, rels or MAG_RELS
    g = torch.Generator(device=dev).manual_seed(seed)
    gs = GraphStore()
    made = {}
    for (s_, r_, d_), m in rels.items():
        fwd = (d_, r_[4:], s_) if r_.startswith("rev_") else None
        if fwd in made:      # a reverse relation is the forward one flipped (PyG's ToUndirected on ogbn-mag)
            ei = made[fwd].flip(0)
        else:
            src = (torch.rand(m, generator=g, device=dev) ** 2 * nodes[s_]).long().clamp_(max=nodes[s_] - 1)
            dst = (torch.rand(m, generator=g, device=dev) ** 2 * nodes[d_]).long().clamp_(max=nodes[d_] - 1)
            ei = torch.stack([src, dst])
        made[(s_, r_, d_)] = ei
        gs[(s_, r_, d_), "coo", False, (nodes[s_], nodes[d_])] = ei
    build_mag_like.graph_store = gs           # (the loader path needs the store itself)
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
            rel[et] = {k: v.to(dev).contiguous() for k, v in dict(w=w, v_src=v_s, v_dst=v_d, att_src=att_s, att_dst=att_d).items()}
        bias = {t: ((torch.rand(HC, generator=g) - 0.5) * 0.1).to(dev) for t in ntypes}
        params.append(dict(rel=rel, bias=bias))
    return params


def build_model(params, etypes, ntypes, dev):
    """The measured model out of the package's own layers: 2 x ``wholegraph_amd.nn.HeteroConv({edge type: GATConv(in, 64,
    heads=4, add_self_loops=False)})`` holding exactly the parameters of ``make_params`` (a node type's bias sits in the first
    relation ending in it, the others carry none: HeteroConv adds the relations' outputs)."""
    from wholegraph_amd import nn
    layers = []
    for layer, fin in enumerate((F_IN, HC)):
        convs, seen = {}, set()
        for et in etypes:
            p = params[layer]["rel"][et]
            c = nn.GATConv(fin, CH, heads=HEADS, add_self_loops=False, bias=et[2] not in seen).to(dev)
            with torch.no_grad():
                c.lin.weight.copy_(p["w"].t())
                c.att_src.copy_(p["att_src"].view(1, HEADS, CH))
                c.att_dst.copy_(p["att_dst"].view(1, HEADS, CH))
                if c.bias is not None:
                    c.bias.copy_(params[layer]["bias"][et[2]])
            seen.add(et[2])
            convs[et] = c
        layers.append(nn.HeteroConv(convs, aggr="sum").to(dev))
    for m in layers:
        for q in m.parameters():
            q.requires_grad_(False)
    return layers


def make_loader(gs, tables, seeds, B, G, fanout=(25, 10), random_state=7):
    """GraphStore + FeatureStore -> ``cugraph_pyg_amd.loader.NeighborLoader`` over the paper seeds, call groups of G mini-batches
    (the reference's surface: loader/neighbor_loader.py:173-201 with a dict fan-out, sampler/sampler.py:231-502)."""
    from cugraph_pyg_amd.data import FeatureStore
    from cugraph_pyg_amd.loader import NeighborLoader
    fs = FeatureStore()
    for t, tab in tables.items():
        fs[t, "x", None] = tab
    return NeighborLoader((fs, gs), {et: list(fanout) for et in sorted(gs._hetero_graphs)}, input_nodes=("paper", seeds),
                          batch_size=B, shuffle=False, random_state=random_state, local_seeds_per_call=G * B)


def forward_group(model, grp):
    """Both layers over one call group -> the seeds' rows [G * B, HC] (x lazy: the first layer's gather makes its attention
    logits in the same pass)."""
    h = grp.x_dict
    for j, layer in enumerate(model):
        h = layer(h, grp.layer_graph(j), act="relu")
    return h["paper"]


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
            "sample": f"{nb} mini-batches of {B} paper seeds: 2-hop [25,10] x {len(etypes)} edge types + feature gather + 2 HeteroConv(GATConv "
                      f"4x64) layers on the C oracle (OpenMP) + torch CPU GEMMs, {dt:.1f} s"}


def train_pass(model, tables, num_nodes, order, B, G, dev, groups=6, warm=2):
    """The TRAINING step of the same configuration through the same API, outside the headline's timed region: call groups ->
    2 x nn.HeteroConv{GATConv} under autograd (aggregate-first: nn._GatAggregateHeads, wgamd_gat_aggregate_heads_bwd_f32; x lazy,
    attention terms of the tables' rows) -> linear head -> cross-entropy on synthetic labels -> backward -> SGD step per group."""
    from wholegraph_amd import nn as wnn
    params = [p for m in model for p in m.parameters()]
    for p in params:
        p.requires_grad_(True)
    head = torch.nn.Linear(HC, 16).to(dev)
    opt = torch.optim.SGD(params + list(head.parameters()), lr=1e-3)
    g = torch.Generator(device=dev).manual_seed(11)
    labels = torch.randint(0, 16, (num_nodes["paper"],), generator=g, device=dev)
    seeds = order[:(groups + warm) * G * B]
    loader = make_loader(build_mag_like.graph_store, tables, seeds, B, G)
    n, edges, t0, first, loss = 0, 0, None, None, None
    for grp in loader.call_groups():
        if n == warm:
            torch.cuda.synchronize()
            t0, edges = time.perf_counter(), 0
        out = head(forward_group(model, grp))
        loss = wnn.cross_entropy(out, labels[seeds[n * G * B:(n + 1) * G * B]])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        first = float(loss.detach()) if first is None else first
        edges += grp.num_edges
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for p in params:
        p.requires_grad_(False)
        p.grad = None
    return {"value": edges / dt, "unit": "sampled-edges/s", "ms_per_call_group": dt / (n - warm) * 1e3, "call_groups": n - warm,
            "loss_first_last": [round(first, 4), round(float(loss.detach()), 4)],
            "note": "NeighborLoader.call_groups() -> 2 x nn.HeteroConv{GATConv 4x64} (autograd, aggregate-first, x lazy) -> linear head -> "
                    "cross-entropy -> backward -> SGD step per call group of %d mini-batches" % G}


def main(args):
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    # BASELINE configs[4] is a single-GPU configuration.  `--gpus N` (N ranks, one per GPU) runs it data-parallel the way §5
    # shards every path: each rank walks its own shard of the seed order over a replica of the graph and of the feature
    # tables (0.6 GB), no data-path collective; the timed region is bracketed by barriers, the time is the MAX over ranks,
    # the edges the SUM.  The per-stage probe, the training variant and the CPU baseline stay with N = 1 (rank 0's line).
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    assert world == args.gpus, "launch through bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    local_rank = 0 if getattr(args, "share_gpu", False) else int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world,
                                **({"device_id": dev} if args.dist_backend == "nccl" else {}))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
    rels = MAG_RELS_R5 if getattr(args, "mag_rels", "all") == "r5" else MAG_RELS
    graphs, num_nodes = build_mag_like(dev, rels=rels)
    n_graph_edges = sum(rels.values())
    etypes = sorted(graphs)
    ntypes = sorted(num_nodes)
    g = torch.Generator(device=dev).manual_seed(5)
    tables = {t: torch.rand((num_nodes[t], F_IN), generator=g, device=dev) * 2 - 1 for t in ntypes}
    params = make_params(etypes, ntypes, dev)
    # call groups of 128 mini-batches: measured on one box at the end of round 5 — 64 -> 2.66, 128 -> 3.39, 192 -> 3.50 G edges/s
    # (0.122 / 0.096 / 0.093 ms per mini-batch: the walk's ~180 launches and the layers' eleven launches per group are what a
    # larger group amortises; rounds 3-4 ran 64).  The loader's own default for this configuration, sized from the walk's
    # buffer capacities within 2 % of the device memory, is 49 (HeteroNeighborSampler.seeds_per_call).
    B, G, gps = 1024, args.call_group if args.call_group > 0 else 128, args.groups_per_step
    from wholegraph_amd import nn
    model = build_model(params, etypes, ntypes, dev)
    for j, m in enumerate(model):
        m.stage_tag = str(j + 1)        # stage names of the probe pass: gat1... = layer 1 (reads x), gat2... = layer 2
    groups = args.steps * gps
    warm = max(args.warmup * gps, 2)
    n_probe = min(groups, 6)
    gs_ = torch.Generator(device=dev).manual_seed(7 + 1000003 * rank)     # every rank its own shard of seeds
    need = (groups + warm + n_probe + 1) * G * B        # (+ 1: the probe pass's own untimed first group)
    reps = -(-need // num_nodes["paper"])
    order = torch.cat([torch.randperm(num_nodes["paper"], generator=gs_, device=dev) for _ in range(reps)])
    fanout = {et: [25, 10] for et in etypes}
    hops = 2

    # ---- the timed region: the package API end to end (NeighborLoader.call_groups() -> HeteroConv x 2) --------------------
    loader = make_loader(build_mag_like.graph_store, tables, order[:(groups + warm) * G * B], B, G)
    edges, n, t0 = 0, 0, None
    with torch.no_grad():
        for grp in loader.call_groups():
            if n == warm:
                barrier()
                t0, edges = time.perf_counter(), 0
            forward_group(model, grp)
            edges += grp.num_edges
            n += 1
    barrier()
    dt = time.perf_counter() - t0
    assert n == groups + warm
    per_rank_value = [edges / dt]
    if dist is not None:
        red_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")
        t_dt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        t_ed = torch.tensor([float(edges)], dtype=torch.float64, device=red_dev)
        t_all = [torch.zeros(1, dtype=torch.float64, device=red_dev) for _ in range(world)]
        dist.all_gather(t_all, torch.tensor([edges / dt], dtype=torch.float64, device=red_dev))
        dist.all_reduce(t_dt, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_ed, op=dist.ReduceOp.SUM)
        dt, edges, per_rank_value = float(t_dt), float(t_ed), [float(v) for v in t_all]
        dist.barrier()
        if rank != 0:
            dist.destroy_process_group()
            return
        args.no_variants, args.no_cpu_baseline = True, True

    # per-stage HIP-event pass (outside the timed region): the layers' stages through nn.set_stage_hook, one call group at a
    # time; the walk alone (the loader's own walk object) between events on the main stream
    acc, launches, nn_sizes = {}, None, None
    timers, walk_ms = [], []

    def hook(name, fn):
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        out_ = fn()
        e_.record()
        timers.append((name, s_, e_))
        return out_
    probe = make_loader(build_mag_like.graph_store, tables, order[(groups + warm) * G * B:need], B, G, random_state=7 + (groups + warm) * G)
    nn.set_stage_hook(hook)
    try:
        with torch.no_grad():
            it = probe.call_groups(overlap=False)
            for gi in range(n_probe + 1):      # group 0 of the pass is not accumulated (first use of this pass's buffer sizes)
                torch.cuda.synchronize()
                del timers[:]
                # (overlap=False: next() enqueues the walk of the FOLLOWING group on this stream — two of them on the first
                #  call, none on the last)
                w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                w0.record()
                grp = next(it)
                w1.record()
                walk_groups = 1 if 0 < gi < n_probe else 0
                s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record()
                lg = [grp.layer_graph(j) for j in range(hops)]
                e_.record()
                timers.append(("index_prep", s_, e_))
                forward_group(model, grp)
                torch.cuda.synchronize()
                if gi == 0:
                    continue
                if walk_groups:
                    walk_ms.append(w0.elapsed_time(w1))
                for name, a_, b_ in timers:
                    key = name.split(":")[0]
                    acc[key] = acc.get(key, 0.0) + a_.elapsed_time(b_)
                    if ":" in name:
                        acc[name] = acc.get(name, 0.0) + a_.elapsed_time(b_)
                nn_sizes = dict(grp.num_nodes)
                launches = [(r.edge_type, r.hop, r.n_rows, r.n_edges, F_IN if j == 0 else HC) for j in range(hops)
                            for r in lg[j].relations if r.n_edges > 0]
    finally:
        nn.set_stage_hook(None)
    stage_ms = {k: v / n_probe for k, v in acc.items() if ":" not in k}
    if walk_ms:
        stage_ms["walk(2 hops x %d edge types)" % len(etypes)] = sum(walk_ms) / len(walk_ms)
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
    if roofline is not None and G == 128 and args.call_group <= 0:
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
    variants = None
    if not getattr(args, "no_variants", False):
        variants = {"train_step": train_pass(model, tables, num_nodes, order, B, G, dev)}
    cpu = None
    if not args.no_cpu_baseline:
        cpu = cpu_baseline(graphs, tables, params, order[:min(order.numel(), 256 * B)].cpu().numpy(), B, fanout, hops,
                           etypes, ntypes, args.cpu_budget)
    out = {"metric": "sampled-edges/sec (hetero 2-hop sample+renumber + feature gather + 2-layer HeteroConv(GATConv 4x64) fwd), "
                     "ogbn-mag-like fan-out [25, 10] x %d edge types" % len(etypes),
           "value": edges / dt, "unit": "sampled-edges/s", "n_gpus": world, "per_rank_value": per_rank_value, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "int64 ids + f32 features (edge softmax + aggregation: f32 HIP kernels; GATConv per-head lin: bf16x3-split MFMA, "
                    "f32 accumulate — the one-kernel relation / wgamd_gat_transform_heads_bf16x3)",
           "data": "synthetic",
           "config": {"workload": "ogbn-mag-like hetero (BASELINE configs[4]): 4 node types (736,389 / 1,134,649 / 8,740 / 59,965), "
                                  "%d directed edge types (%s) ~%.1f M edges, feat fp32 [n_t, 128] per type, batch 1024 paper seeds, 2-hop fan-out "
                                  "[25,10] per edge type, 2 x HeteroConv{GATConv(., 64, heads=4)} sum + ReLU, step = %d call groups "
                                  "of %d mini-batches" % (len(etypes), ", ".join(et[1] for et in etypes), n_graph_edges / 1e6, gps, G),
                      "parallelism": "1 GPU" if world == 1 else "dp%d (seeds sharded, graph + tables replicated, no data-path collective)" % world},
           "call_group": G, "batches_per_step": G * gps, "timed_region_ms": round(dt * 1e3, 2), "timed_call_groups": groups,
           "ms_per_batch": dt / (groups * G) * 1e3, "edges_per_batch": edges / (groups * G * world),
           "nodes_per_call_group": nn_sizes, "stage_ms_per_call_group": {k: round(v, 4) for k, v in stage_ms.items()},
           "gat_launches": [{"edge_type": "%s-%s-%s" % et, "hop": h + 1, "rows": n_f, "edges": n_e, "src_row_floats": f_}
                            for et, h, n_f, n_e, f_ in (launches or [])],
           "roofline": roofline, "cpu_baseline": cpu}
    if variants is not None:
        out["variants"] = variants
    if cpu is not None:
        out["gpu_over_cpu"] = round(out["value"] / cpu["value"], 2)
    print(json.dumps(out), flush=True)
