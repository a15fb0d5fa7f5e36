#!/usr/bin/env python
"""Per-operator throughput of the C-ABI entry points on one MI355X (companion of bench.py).

Each line of the result is one §8(a) row timed on its own through the same Python wrappers the parity tests use:
algorithmic bytes per SURVEY.md §8(d) ÷ wall time of the call (the ABI ops synchronise internally, so a call is
complete on return; HIP events bracket the enqueue-only ops).  ``frac`` is against the 8 TB/s HBM3E spec.  The sizes
are one call group of the products workload (64 mini-batches of 1024 seeds, fan-out [25, 10]) plus the reference's
``gather_scatter_bench`` shape (cpp/bench/wholememory_ops/gather_scatter_bench.cu:331-360: 1 M gathered rows).

    python bench_ops.py [--json profiles/rNN/ops_n1.json]
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

import torch  # noqa: E402

from bench import HBM_PEAK_GBPS, V_PRODUCTS, E_UNDIRECTED, rmat_csr  # noqa: E402


def timed(fn, iters=20, warm=14, events=False):
    """Seconds per call: MEDIAN of five timed chunks after `warm` untimed calls (the caching allocator settles on an op's
    block sizes after ~10 calls; a one-off 40 ms stall of the runtime still lands in one op's loop now and then — the median of
    chunks keeps it out of the table, a mean over one loop read 2 ms for a 0.1 ms op)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    per = max(iters // 5, 2)
    chunks = []
    for _ in range(5):
        if events:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(per):
                fn()
            e.record()
            torch.cuda.synchronize()
            chunks.append(s.elapsed_time(e) * 1e-3 / per)
        else:
            t0 = time.perf_counter()
            for _ in range(per):
                fn()
            torch.cuda.synchronize()
            chunks.append((time.perf_counter() - t0) / per)
    return sorted(chunks)[2]


def hetero_section(dev):
    """BASELINE configs[4] shape: ogbn-mag-like heterogeneous graph (4 node types, 4 relations + 2 reverse), 2-hop
    fan-out [25, 10] per edge type, batch 1024, sampled in call groups of 32 mini-batches; then GATConv (4 heads x 64)
    over the sampled author-writes-paper relation of one call group."""
    import numpy as np
    from cugraph_pyg_amd.data import GraphStore
    from cugraph_pyg_amd.sampler.sampler import HeteroNeighborSampler, hetero_neighbor_sample
    from wholegraph_amd import nn
    g = torch.Generator(device=dev).manual_seed(11)
    n = {"paper": 736_389, "author": 1_134_649, "institution": 8_740, "field_of_study": 59_965}
    rel = {("author", "writes", "paper"): 7_145_660, ("paper", "cites", "paper"): 5_416_271,
           ("paper", "has_topic", "field_of_study"): 7_505_078, ("author", "affiliated_with", "institution"): 1_043_998,
           ("paper", "rev_writes", "author"): 7_145_660, ("field_of_study", "rev_has_topic", "paper"): 7_505_078}
    gs = GraphStore()
    for (s_, r_, d_), m in rel.items():
        # skewed endpoints (squared uniform) so that hubs exist, as in the real graph
        src = (torch.rand(m, generator=g, device=dev) ** 2 * n[s_]).long().clamp_(max=n[s_] - 1)
        dst = (torch.rand(m, generator=g, device=dev) ** 2 * n[d_]).long().clamp_(max=n[d_] - 1)
        gs[(s_, r_, d_), "coo", False, (n[s_], n[d_])] = torch.stack([src, dst])
    graphs = gs._hetero_graphs
    fanout = {et: [25, 10] for et in rel}
    B, G, groups = 1024, 32, 6
    seeds = torch.randperm(n["paper"], generator=g, device=dev)[:B * G * groups]
    smp = HeteroNeighborSampler(graphs, fanout, local_seeds_per_call=B * G, num_nodes=n)
    list(smp.sample_batches("paper", seeds[:B * G], B, 1))          # warm-up (allocations, workspace)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    edges = nodes = nb = 0
    last = None
    for b, out in smp.sample_batches("paper", seeds, B, 7):
        edges += sum(sum(v) for v in out[5].values())
        nodes += sum(int(v.numel()) for v in out[0].values())
        nb += 1
        last = out
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for b in range(4):                                                # the one-batch-at-a-time route, for scale
        hetero_neighbor_sample(graphs, "paper", seeds[b * B:(b + 1) * B], smp.fanout, 7 + b)
    torch.cuda.synchronize()
    slow = (time.perf_counter() - t1) / 4
    et = ("author", "writes", "paper")
    node, row, col, _, _, _ = last
    n_dst, n_src, H, C = node["paper"].numel(), node["author"].numel(), 4, 64
    rp, cc = nn._to_csr(torch.stack([row[et], col[et]]), n_dst)
    x = torch.rand((n_src, H * C), generator=g, device=dev)
    a_s, a_d = torch.rand((n_src, H), generator=g, device=dev), torch.rand((n_dst, H), generator=g, device=dev)
    tg = timed(lambda: nn.gat_forward(rp, cc, x, a_s, a_d, H, 0.2, need_alpha=False), events=True)
    # roofline of the GAT kernel (single pass, online softmax): per edge the source row, its head scores and
    # the column index; per destination its head scores and the output row
    E_gat = int(cc.shape[0])
    gat_bytes = E_gat * (4 * H * C + 4 * H + 4) + n_dst * (4 * H + 4 * H * C + 8)
    gat_roof = {"bound": "hbm", "kernel": "gat_csr_kernel", "achieved": round(gat_bytes / tg / 1e9, 1), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(gat_bytes / tg / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                "algorithmic_bytes_per_launch": int(gat_bytes), "avg_launch_ms": round(tg * 1e3, 5),
                "shape": {"E": E_gat, "n_dst": n_dst, "n_src": n_src, "heads": H, "channels": C}}
    # CPU baseline (rank-0 host cores): the same heterogeneous composition on the C oracle, one mini-batch at a time
    cpu = None
    if os.environ.get("WGAMD_BENCH_NO_CPU") != "1":
        import oracle
        from cugraph_pyg_amd.sampler.sampler import hop_seed
        oracle.build()
        from bench import usable_cpus
        oracle.set_num_threads(usable_cpus())
        hg = {et: (gr.row_ptr.cpu().numpy(), gr.col.cpu().numpy()) for et, gr in graphs.items()}
        etypes = sorted(hg)
        ntypes = sorted({t for et in etypes for t in (et[0], et[2])})
        t0, c_edges, c_b = time.perf_counter(), 0, 0
        seeds_h = seeds.cpu().numpy()
        while time.perf_counter() - t0 < 8.0 and (c_b + 1) * B <= len(seeds_h):
            node = {t: np.zeros(0, np.int64) for t in ntypes}
            node["paper"] = seeds_h[c_b * B:(c_b + 1) * B].astype(np.int64)
            fstart = {t: 0 for t in ntypes}
            for h in range(2):
                begin = {t: len(node[t]) for t in ntypes}
                for ti, et in enumerate(etypes):
                    frontier = node[et[2]][fstart[et[2]]:begin[et[2]]]
                    if len(frontier) == 0:
                        continue
                    rp_h, col_h = hg[et]
                    _, nbr, _, _ = oracle.unweighted_sample(rp_h, col_h, frontier, fanout[et][h], hop_seed(7 + c_b, h * len(etypes) + ti))
                    node[et[0]], _ = oracle.append_unique(node[et[0]], nbr.astype(np.int64))
                    c_edges += int(nbr.size)
                for t in ntypes:
                    fstart[t] = begin[t]
            c_b += 1
        c_dt = time.perf_counter() - t0
        cpu = {"value": c_edges / c_dt, "unit": "sampled-edges/s", "cores": usable_cpus(), "kind": "port",
               "sample": f"{c_b} mini-batches of {B} paper seeds, 2-hop [25,10] x 6 edge types on the C oracle (OpenMP), {c_dt:.1f} s"}
    line = {"metric": "sampled-edges/sec (heterogeneous 2-hop sampling + renumber, ogbn-mag-like) + GATConv edge-softmax",
            "value": edges / dt, "unit": "sampled-edges/s", "n_gpus": 1, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int64 ids + f32 features", "data": "synthetic",
            "config": {"workload": "ogbn-mag-like hetero: 4 node types (736,389 / 1,134,649 / 8,740 / 59,965), 6 edge types "
                                   "~35.8 M edges, 2-hop fan-out [25,10] per edge type, batch 1024, call groups of 32; "
                                   "GATConv 4 heads x 64 on the sampled author-writes-paper relation"},
            "ms_per_batch": dt / nb * 1e3, "edges_per_batch": edges / nb, "roofline": gat_roof, "cpu_baseline": cpu,
            "gpu_over_cpu": None if cpu is None else round(edges / dt / cpu["value"], 2)}
    print(json.dumps(line), flush=True)
    return {"op": "hetero (ogbn-mag-like) call-group sampling, 2-hop [25,10] x 6 edge types, batch 1024",
            "reference": "pylibcugraph.heterogeneous_uniform_neighbor_sample via cugraph_pyg (a14)",
            "ms_per_batch": round(dt / nb * 1e3, 4), "edges_per_s": round(edges / dt, 1),
            "edges_per_batch": round(edges / nb, 1), "nodes_per_batch": round(nodes / nb, 1),
            "ms_per_batch_one_at_a_time": round(slow * 1e3, 3), "call_group": G,
            "gat_one_batch_ms": round(tg * 1e3, 4),
            "note": "per-batch outputs materialised as python tuples (finalize_batches); GAT H=4 C=64 forward on the "
                    "sampled author-writes-paper relation of one mini-batch"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--nodes", type=int, default=V_PRODUCTS)
    ap.add_argument("--edges", type=int, default=E_UNDIRECTED)
    ap.add_argument("--hetero", action="store_true", help="add the ogbn-mag-like heterogeneous loader measurement")
    ap.add_argument("--hetero-only", action="store_true",
                    help="BASELINE configs[4] only: print ONE JSON line (metric, roofline of the GAT kernel, cpu_baseline)")
    args = ap.parse_args()
    assert torch.cuda.is_available(), "bench_ops.py needs a GPU (no CPU fallback)"
    dev = torch.device("cuda", 0)
    if args.hetero_only:
        hetero_section(dev)
        return
    import wholegraph_amd as wg
    from wholegraph_amd import graph_ops, nn, wholegraph_ops
    from wholegraph_amd.tensor import local_gather, local_scatter

    row_ptr, col = rmat_csr(args.nodes, args.edges, 0, dev)
    V, E = args.nodes, col.shape[0]
    g = torch.Generator(device=dev).manual_seed(3)
    weight = torch.rand(E, generator=g, device=dev) + 0.01
    rows = []

    def add(name, ref, seconds, nbytes, units, unit_name, note=""):
        gbps = nbytes / seconds / 1e9
        rows.append({"op": name, "reference": ref, "ms": round(seconds * 1e3, 4), "algorithmic_bytes": int(nbytes),
                     "GBps": round(gbps, 1), "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS, 4),
                     unit_name + "_per_s": round(units / seconds, 1), "note": note})
        print(rows[-1], flush=True)

    # ---- a1: uniform sampling, hop 1 (65,536 seeds, M=25) and hop 2 (~600 k frontier, M=10) ------------------
    seeds = torch.randperm(V, generator=g, device=dev)[:64 * 1024]
    b = 8
    for name, centers, M in (("unweighted_sample hop1 M=25", seeds, 25), ("unweighted_sample hop2 M=10", None, 10)):
        if centers is None:
            centers = frontier
        out = wholegraph_ops.unweighted_sample_without_replacement(row_ptr, col, centers, M, random_seed=62)
        e_hop = out[1].shape[0]
        t = timed(lambda: wholegraph_ops.unweighted_sample_without_replacement(row_ptr, col, centers, M, random_seed=62))
        add(name, "wholegraph_op.h:31-42", t, centers.shape[0] * (b + 16 + 4) + e_hop * (2 * b + 4), e_hop, "edges",
            "ABI op: 2 internal host syncs (count, total)")
        if M == 25:
            hop1 = out
            uniq, mapping = graph_ops.append_unique(seeds, out[1], need_neighbor_raw_to_unique=True)
            frontier = uniq
    # ---- a4: weighted sampling ---------------------------------------------------------------------------------
    outw = wholegraph_ops.weighted_sample_without_replacement(row_ptr, col, weight, frontier, 10, random_seed=62)
    fdeg = row_ptr[frontier + 1] - row_ptr[frontier]
    deg_sum = int(fdeg.sum())
    # (a dozen untimed calls: the caching allocator settles on this op's block sizes only after ~10 of them — with 3 the timed
    #  loop still paid device allocations and read 4-5 ms for a 0.7 ms op)
    t = timed(lambda: wholegraph_ops.weighted_sample_without_replacement(row_ptr, col, weight, frontier, 10, random_seed=62),
              iters=10, warm=14)
    add("weighted_sample hop2 M=10", "wholegraph_op.h:61-73", t,
        frontier.shape[0] * (b + 16 + 4) + deg_sum * 4 + outw[1].shape[0] * (2 * b + 4), outw[1].shape[0], "edges",
        "reads every candidate weight (Σdeg = %d) to key it; %d rows > 1024 candidates (Σ = %d, max %d)"
        % (deg_sum, int((fdeg > 1024).sum()), int(fdeg[fdeg > 1024].sum()), int(fdeg.max())))
    # ---- a7: renumber --------------------------------------------------------------------------------------------
    hop2 = wholegraph_ops.unweighted_sample_without_replacement(row_ptr, col, frontier, 10, random_seed=63)
    T, Eh = frontier.shape[0], hop2[1].shape[0]
    u2, _ = graph_ops.append_unique(frontier, hop2[1], need_neighbor_raw_to_unique=True)
    t = timed(lambda: graph_ops.append_unique(frontier, hop2[1], need_neighbor_raw_to_unique=True))
    add("graph_append_unique hop2", "graph_op.h:27-33", t, (T + Eh) * b + u2.shape[0] * b + 4 * Eh, T + Eh, "keys",
        "hash-table traffic is overhead, not counted (§8d)")
    # ---- a9: self loops --------------------------------------------------------------------------------------------
    rp32, ci32 = hop2[0], torch.randint(0, T, (Eh,), device=dev, dtype=torch.int32)
    t = timed(lambda: graph_ops.add_csr_self_loop(rp32, ci32))
    add("csr_add_self_loop", "graph_op.h:44-48", t, (T + 1) * 8 + Eh * 4 + (Eh + T) * 4, Eh + T, "entries")
    # ---- a10/a11: gather / scatter ----------------------------------------------------------------------------
    for F, n_rows, label in ((100, u2.shape[0], "products rows"), (128, 1 << 20, "1M rows (gather_scatter_bench)"),
                             (256, 1 << 20, "1M rows"), (1024, 1 << 18, "256k rows")):
        table = torch.rand((V, F), generator=g, device=dev)
        idx = torch.randint(0, V, (n_rows,), generator=g, device=dev)
        out = torch.empty((n_rows, F), device=dev)
        t = timed(lambda: local_gather(table, idx, out), events=True)
        add("wholememory_gather F=%d, %s" % (F, label), "wholememory_op.h:25-30", t, n_rows * (b + 8 * F), n_rows, "rows",
            "reference-style figure n*4F/t = %.1f GB/s" % (n_rows * 4 * F / t / 1e9))
        perm = torch.randperm(V, generator=g, device=dev)[:n_rows]
        t = timed(lambda: local_scatter(out, perm, table), events=True)
        add("wholememory_scatter F=%d, %s" % (F, label), "wholememory_op.h:42-47", t, n_rows * (b + 8 * F), n_rows, "rows")
        if F == 100:
            half = table.half()
            t = timed(lambda: local_gather(half, idx, out), events=True)
            add("wholememory_gather fp16->fp32 F=100", "gather_scatter_func.cuh:242-505", t, n_rows * (b + 6 * F), n_rows,
                "rows")
        del table, out
    # ---- a18: aggregation ----------------------------------------------------------------------------------------
    n_src = u2.shape[0]
    col2 = graph_ops.append_unique(frontier, hop2[1], need_neighbor_raw_to_unique=True)[1]
    for F in (100, 256):
        x = torch.rand((n_src, F), generator=g, device=dev)
        t = timed(lambda: nn.spmm_csr_forward(hop2[0], col2, x, mean=True), events=True)
        add("SAGE mean SpMM F=%d" % F, "torch_geometric SAGEConv (external)", t, Eh * (4 * F + 4) + T * (4 * F + 8), Eh, "edges")
        gout = torch.rand((T, F), generator=g, device=dev)
        fn = nn._SpmmCsr.apply
        xr = x.clone().requires_grad_(True)
        y = fn(xr, hop2[0], col2, True)

        def bwd():
            xr.grad = None
            y.backward(gout, retain_graph=True)
        t = timed(bwd, events=True, iters=10)
        add("SAGE mean SpMM backward F=%d" % F, "—", t, Eh * (8 * F + 4) + T * (4 * F + 8) + n_src * 4 * F, Eh, "edges",
            "CSR transpose (stable sort) + forward gather kernel, no atomics")
        t = timed(lambda: nn.spmm_csr_backward(hop2[0], col2, gout, n_src, True, atomic=True), events=True, iters=10)
        add("SAGE mean SpMM backward (atomic variant) F=%d" % F, "—", t, Eh * (8 * F + 4) + T * (4 * F + 8) + n_src * 4 * F,
            Eh, "edges", "wgamd_spmm_csr_bwd_f32: one kernel, fp32 atomic scatter-add (incl. zeroing grad_x)")
    for H, C in ((4, 64), (1, 256)):
        x = torch.rand((n_src, H * C), generator=g, device=dev)
        a_s = torch.rand((n_src, H), generator=g, device=dev)
        a_d = torch.rand((T, H), generator=g, device=dev)
        t = timed(lambda: nn.gat_forward(hop2[0], col2, x, a_s, a_d, H, 0.2, need_alpha=False), events=True)
        add("GAT edge-softmax + SpMM H=%d C=%d" % (H, C), "torch_geometric GATConv (external)", t,
            Eh * (4 * H + 4) + T * 4 * H + Eh * (4 * H * C + 4 * H + 4) + T * 4 * H * C, Eh, "edges",
            "one fused kernel (online softmax): scores are never written")
        t = timed(lambda: nn.gat_forward(hop2[0], col2, x, a_s, a_d, H, 0.2, need_alpha=True), events=True)
        add("GAT (+alpha out for backward) H=%d C=%d" % (H, C), "—", t,
            Eh * (4 * H + 4) + T * 4 * H + Eh * (4 * H * C + 4 * H + 4) + T * 4 * H * C + Eh * 4 * H, Eh, "edges")
    # ---- aggregate-first GATConv relation on the products hop-2 shape (F = 128 source floats, 4 heads x 64): two kernels / one
    Fg, Hg, Cg = 128, 4, 64
    xg = torch.rand((n_src, Fg), generator=g, device=dev) - 0.5
    a_s = torch.rand((n_src, Hg), generator=g, device=dev)
    a_d = torch.rand((T, Hg), generator=g, device=dev)
    wg_ = (torch.rand((Fg, Hg * Cg), generator=g, device=dev) - 0.5) * 0.2
    bias_g = torch.rand(Hg * Cg, generator=g, device=dev)
    agg_bytes = Eh * (4 * Fg + 4 * Hg + 4) + T * (4 * Hg * Fg + 4 * Hg + 8)
    t = timed(lambda: nn.gat_aggregate_heads(hop2[0], col2, xg, a_s, a_d, Hg), events=True)
    add("GAT aggregate-first: aggregation H=4 F=128", "torch_geometric GATConv (external)", t, agg_bytes, Eh, "edges",
        "wgamd_gat_aggregate_heads_f32: agg[i, h, :] = sum_e alpha_e^h x[col[e], :]")
    agg = nn.gat_aggregate_heads(hop2[0], col2, xg, a_s, a_d, Hg)
    t = timed(lambda: nn.gat_transform_heads_fused(agg, wg_, Hg, bias=bias_g, relu=True), events=True)
    add("GAT aggregate-first: per-head transform + bias + ReLU, 4 x (128 -> 64)", "torch_geometric GATConv lin (external)", t,
        T * (4 * Hg * Fg + 4 * Hg * Cg), T, "rows", "wgamd_gat_transform_heads_bf16x3 (bf16x3-split MFMA, fp32 accumulate)")
    t = timed(lambda: nn.gat_transform_heads(agg, wg_, Hg, fused=False), events=True)
    add("GAT aggregate-first: per-head transform, library strided batched GEMM", "—", t, T * (4 * Hg * Fg + 4 * Hg * Cg), T,
        "rows", "torch.bmm through hipBLASLt (fp32 MFMA), no bias / ReLU")
    if nn.gat_layer_fused_supported(Fg, Hg, Cg):
        t = timed(lambda: nn.gat_layer_fused(hop2[0], col2, xg, a_s, a_d, wg_, Hg, bias=bias_g, relu=True), events=True)
        add("GAT aggregate-first relation as ONE kernel (aggregation + transform + bias + ReLU)", "torch_geometric GATConv (external)", t,
            Eh * (4 * Fg + 4 * Hg + 4) + T * (4 * Hg * Cg + 4 * Hg + 8), Eh, "edges",
            "wgamd_gat_layer_fused_bf16x3: the [rows, 4 x 128] aggregate stays in LDS")
    # ---- (f4): trainable embedding, sparse optimizer step (world of 1: routing is a local copy) ------------------
    comm = wg.create_group_communicator()
    n_rows, k = 1_000_000, 1_000_000
    for kind, n_state in (("sgd", 0), ("lazy_adam", 2)):
        for dim in (128,):
            emb = wg.create_embedding(comm, "distributed", "cuda", torch.float32, [n_rows, dim], random_init=True)
            opt = wg.create_wholememory_optimizer(emb, kind, {})
            idx = torch.randint(0, n_rows, (k,), generator=g, device=dev)
            grads = torch.rand((k, dim), generator=g, device=dev)
            uniq = int(torch.unique(idx).numel())

            def step():
                emb.add_gradients(idx, grads)
                emb.apply_gradients(0.01)
            t = timed(step, iters=10)
            nbytes = k * (8 + 4 * dim) + uniq * dim * 4 * 2 * (1 + n_state)
            add("embedding gather_gradient_apply %s dim=%d, %d pairs -> %d rows" % (kind, dim, k, uniq),
                "wholememory_embedding_gather_gradient_apply (f4)", t, nbytes, k, "pairs",
                "wall time incl. the op's host syncs; route (copy) + radix sort of (row, position) + fused sum/update kernel")
            t = timed(lambda: emb.gather(idx), events=True)
            add("embedding gather dim=%d (DISTRIBUTED handle, world 1)" % dim, "wholememory_embedding_gather", t,
                k * (8 + 8 * dim), k, "rows")
            wg.destroy_embedding(emb)
            wg.destroy_wholememory_optimizer(opt)
    # ---- (f4): a table in pinned HOST memory, read and trained in place over PCIe and through the READWRITE device cache ----
    n_host, dim, k = 4_000_000, 128, 500_000
    hot = (torch.empty(k, device=dev).exponential_(1.0, generator=g) * (n_host * 0.01)).long().clamp_(max=n_host - 1)
    for ratio in (None, 0.05):
        pol = None if ratio is None else wg.create_wholememory_cache_policy(comm, memory_type="distributed", memory_location="cuda",
                                                                            access_type="readwrite", ratio=ratio)
        emb = wg.create_embedding(comm, "distributed", "cpu", torch.float32, [n_host, dim], cache_policy=pol)
        emb.get_embedding_tensor().get_local_tensor()[0].normal_()
        opt = wg.create_wholememory_optimizer(emb, "lazy_adam", {})
        grads = torch.rand((k, dim), generator=g, device=dev)
        uniq = int(torch.unique(hot).numel())
        what = "host table [%d, %d] fp32 (pinned, %.1f GB), ids ~ exp(mean 1 %% of the rows), %d distinct of %d" % (
            n_host, dim, n_host * dim * 4 / 1e9, uniq, k)
        label = "no cache: every row crosses PCIe" if ratio is None else "READWRITE device cache, ratio %.2f (%d lines)" % (
            ratio, emb.cache_stats()[2])

        def step():
            emb.add_gradients(hot, grads)
            emb.apply_gradients(0.01)
        t = timed(lambda: emb.gather(hot), iters=10, warm=4)
        h0, l0, _ = emb.cache_stats()
        emb.gather(hot)
        h1, l1, _ = emb.cache_stats()
        add("embedding gather, %s" % label, "wholememory_embedding_gather (f4, device_cached_host_embedding)", t,
            k * (8 + 8 * dim), k, "rows", what + ("" if ratio is None else "; warm hit rate %.3f" % ((h1 - h0) / max(1, l1 - l0))))
        t = timed(step, iters=10, warm=4)
        add("embedding gather_gradient_apply lazy_adam, %s" % label, "wholememory_embedding_gather_gradient_apply (f4)", t,
            k * (8 + 4 * dim) + uniq * dim * 4 * 2 * 3, k, "pairs", what)
        if ratio is not None:
            t = timed(lambda: emb.writeback_all_cache(), iters=3, warm=1)
            add("embedding writeback_all_cache (%d lines, row + m + v)" % emb.cache_stats()[2], "wholememory_embedding_writeback_cache", t,
                0, emb.cache_stats()[2], "lines", "after the first call nothing is dirty: the scan of the tags")
        wg.destroy_embedding(emb)
        wg.destroy_wholememory_optimizer(opt)
        if pol is not None:
            wg.destroy_wholememory_cache_policy(pol)
    # ---- a14-a17: the cugraph_pyg-shaped loader end to end (GraphStore + FeatureStore -> NeighborLoader -> Data with x) ------
    from cugraph_pyg_amd.data import FeatureStore, GraphStore
    from cugraph_pyg_amd.loader import NeighborLoader
    gs_, fs_ = GraphStore(), FeatureStore()
    dst_ = torch.repeat_interleave(torch.arange(V, device=dev), row_ptr[1:] - row_ptr[:-1])
    gs_[("n", "e", "n"), "coo", False, (V, V)] = torch.stack([col, dst_])
    del dst_
    fs_["n", "x", None] = torch.rand((V, 100), generator=g, device=dev)
    for calls in (1, 64):
        ld_seeds = torch.randperm(V, generator=g, device=dev)[:1024 * 64 * 3]
        loader = NeighborLoader((fs_, gs_), [25, 10], input_nodes=ld_seeds, batch_size=1024,
                                local_seeds_per_call=1024 * calls, shuffle=False)
        it = iter(loader)
        next(it)
        torch.cuda.synchronize()
        t0, n_e, n_b = time.perf_counter(), 0, 0
        for batch in it:
            n_e += int(batch.edge_index.shape[1])
            n_b += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rows.append({"op": "NeighborLoader (PyG-shaped Data per batch, x gathered) batch 1024 fan-out [25,10], call group %d" % calls,
                     "reference": "cugraph_pyg.loader.NeighborLoader (a14-a17)", "ms_per_batch": round(dt / n_b * 1e3, 4),
                     "edges_per_s": round(n_e / dt, 1), "edges_per_batch": round(n_e / n_b, 1),
                     "note": "wall time of the Python iteration: sampling + renumbering in call groups, per-batch feature "
                             "fetch and Data construction"})
        print(rows[-1], flush=True)
    from cugraph_pyg_amd.loader import LinkNeighborLoader
    sel = torch.randint(0, E, (512 * 100,), generator=g, device=dev)
    eli = gs_.get_edge_index(("n", "e", "n"), "coo")[:, sel]
    for groups in (False, True):
        loader = LinkNeighborLoader((fs_, gs_), num_neighbors=[25, 10], edge_label_index=eli, batch_size=512,
                                    neg_sampling=("binary", 1.0), shuffle=False, call_groups=groups)
        it = iter(loader)
        next(it)
        torch.cuda.synchronize()
        t0, n_e, n_b = time.perf_counter(), 0, 0
        for batch in it:
            n_e += int(batch.edge_index.shape[1])
            n_b += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rows.append({"op": "LinkNeighborLoader batch 512 seed edges + 512 binary negatives, fan-out [25,10], %s"
                           % ("call groups of 16" if groups else "one batch per call"),
                     "reference": "cugraph_pyg.loader.LinkNeighborLoader (f1)", "ms_per_batch": round(dt / n_b * 1e3, 4),
                     "edges_per_s": round(n_e / dt, 1), "edges_per_batch": round(n_e / n_b, 1),
                     "note": "negatives + row-wise endpoint de-duplication + walk over ragged seed lists + feature fetch"})
        print(rows[-1], flush=True)
    del gs_, fs_
    if args.hetero:
        rows.append(hetero_section(dev))
        print(rows[-1], flush=True)
    result = {"device": torch.cuda.get_device_name(0), "graph": {"V": V, "E_directed": int(E)}, "hbm_peak_GBps": HBM_PEAK_GBPS,
              "rows": rows}
    if args.json:
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        with open(args.json, "w") as f:
            json.dump(result, f, indent=1)
    print(json.dumps({"ops": len(rows)}))


if __name__ == "__main__":
    main()
