"""PyG-style multi-hop neighbour sampling on the HIP kernels + the iterator that joins features.

Reference call stack: ``NeighborLoader`` → ``BaseSampler.sample_from_nodes``
(/root/reference/python/cugraph-pyg/cugraph_pyg/sampler/sampler.py:756-797) →
``DistributedNeighborSampler`` → ``pylibcugraph.homogeneous_*_neighbor_sample(renumber=True,
return_hops=True, retain_seeds=True, prior_sources_behavior='exclude', deduplicate_sources=True)``
(sampler/distributed_sampler.py:877-908; the arithmetic is in libcugraph, NOT in the reference tree,
so its RNG stream is "parity unpinned" — SURVEY.md §8(c)) → ``HomogeneousSampleReader._decode``
(:525-730) → ``SampleIterator`` (:51-165).

Here the same contract is produced by chaining the WholeGraph-parity kernels of libwholegraph_amd:
hop k expands only the vertices first discovered at hop k-1 (``prior_sources_behavior='exclude'``,
``deduplicate_sources``), the renumber map keeps the seeds first (``retain_seeds``), edges come out
hop by hop, ``row`` = local id of the sampled neighbour (PyG message source), ``col`` = local id
of the expanded vertex, ``edge`` = original edge id.  The random part is pinned to the oracle
(tests/test_gpu_pyg_loader.py composes the oracle the same way), the structural invariants are
the reference tests' own (tests/loader/test_neighbor_loader.py:20-133).
"""
from typing import Iterator, Optional, Sequence

import torch

from wholegraph_amd import graph_ops, wholegraph_ops

from .._compat import Data, HeteroData, HeteroSamplerOutput, NodeSamplerInput, SamplerOutput  # noqa: F401
from ..data.graph_store import CSRGraph

_GOLDEN = 0x9E3779B97F4A7C15
_MASK = 0xFFFFFFFFFFFFFFFF


def _as_i64(v: int) -> int:
    """two's-complement view of a 64-bit seed (torch has no uint64 tensors)"""
    v &= _MASK
    return v - (1 << 64) if v & (1 << 63) else v


def hop_seed(random_state: int, hop: int) -> int:
    """Seed of hop ``hop`` of a call whose ``random_state`` is given (batch b of a loader epoch uses
    ``random_state + b``, as ``random_state + rank`` in distributed_sampler.py:896)."""
    return (int(random_state) + hop * _GOLDEN) & _MASK


# ---------------------------------------------------------------------------------------------------------------------
# call-group sizing (the role of DistributedNeighborSampler.__calc_local_seeds_per_call,
# sampler/distributed_sampler.py:757,837-875)
# ---------------------------------------------------------------------------------------------------------------------
# The reference sizes a call from the device's total memory (0.11 output vertices per byte, measured on its own buffers)
# and falls back to 32,768 seeds when a fan-out is not positive.  Re-derived here for the buffers of THIS walk and for
# 288 GB of HBM3E: a call group may take CALL_GROUP_MEMORY_FRACTION of the device memory; what a seed costs is counted
# from the capacity-sized arrays of PygNoSyncWalk (per hop: 5 int32 + edge id + frontier id per sampled-edge slot, id +
# batch per node slot) plus the library's own workspace figure for the largest hop.
CALL_GROUP_MEMORY_FRACTION = 0.02
# The rows a call group FETCHES are not walk buffers but they are alive with them: one gather per stored attribute for all
# batches of the group.  Their worst case (every seed with fan-out^hops distinct neighbours) may take this fraction of the
# device memory — with 400-byte rows (products) that allows 255 mini-batches and the walk budget decides; with 4 KB rows it is
# 24 mini-batches instead of 191 and the group fetch stays a few GB instead of tens.
CALL_GROUP_FEATURE_FRACTION = 0.10
UNKNOWN_VERTICES_DEFAULT = 32768        # distributed_sampler.py:761-763
_CALL_GROUP_CAPACITY = (1 << 30) - 1    # node + edge slots one call may address (int32 rows inside the kernels)


def call_group_bytes_per_seed(fanout: Sequence[int], id_bytes: int = 8, disjoint: bool = False) -> float:
    """Device bytes one seed of a call group costs (capacity-sized buffers + workspace of the largest hop)."""
    from wholegraph_amd import _lib as L
    frontier, nodes, total = 1, 1, 0.0
    ws = 0
    wm_dtype = L.DT_INT64 if id_bytes == 8 else L.DT_INT
    for m in fanout:
        edges = frontier * m
        total += edges * (5 * 4 + 8 + id_bytes + 4) + (nodes + edges) * (id_bytes + 4) + (frontier + 1) * 4
        # workspace is shared by the hops: the largest one counts (queried at 4096 seeds, it is linear in the capacities)
        ws = max(ws, L.lib().wgamd_sample_hop_workspace_bytes(4096 * max(nodes, frontier), 4096 * edges, wm_dtype) / 4096.0)
        nodes, frontier = nodes + edges, edges
    total += ws
    if disjoint:   # no cross-seed de-duplication: every seed keeps its own tree (the reference scales by fanout[0] too)
        total *= max(int(fanout[0]), 1)
    return total


def store_row_bytes(feature_store):
    """``(node_row_bytes, edge_row_bytes)``: bytes of all stored attributes per node / per edge — the widest node type and
    edge type of a heterogeneous store (a call group's fetch is sized for its worst type)."""
    node, edge = {}, {}
    for attr in feature_store.get_all_tensor_attrs():
        t = feature_store[attr.group_name, attr.attr_name, None]
        b = torch.empty((), dtype=t.dtype).element_size()
        for d in tuple(t.shape)[1:]:
            b *= int(d)
        side = edge if isinstance(attr.group_name, tuple) else node
        side[attr.group_name] = side.get(attr.group_name, 0) + b
    return max(node.values(), default=0), max(edge.values(), default=0)


def default_local_seeds_per_call(fanout: Sequence[int], batch_size: int, id_bytes: int = 8, disjoint: bool = False,
                                 total_memory: Optional[int] = None, feature_row_bytes=(0, 0)) -> int:
    """Seeds per call group when the user gives none: the memory budget above, never less than one mini-batch, never more
    slots than one call can address, a whole number of mini-batches.  On a 288 GB MI355X, fan-out [25, 10], int64 ids:
    about 140 mini-batches of 1024 seeds (bench.py runs 64 and measures 256 as 5 % faster still)."""
    fanout = [int(f) for f in fanout]
    if any(f <= 0 for f in fanout):
        per_call = UNKNOWN_VERTICES_DEFAULT
    else:
        if total_memory is None:
            total_memory = (torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory
                            if torch.cuda.is_available() else 16 << 30)
        slots = 1
        frontier, nodes = 1, 1
        for m in fanout:
            nodes, frontier = nodes + frontier * m, frontier * m
            slots = nodes + frontier
        per_call = int(CALL_GROUP_MEMORY_FRACTION * total_memory / call_group_bytes_per_seed(fanout, id_bytes, disjoint))
        per_call = min(per_call, _CALL_GROUP_CAPACITY // max(slots, 1))
        fetched = nodes * int(feature_row_bytes[0]) + (nodes - 1) * int(feature_row_bytes[1])   # rows a seed can pull in
        if fetched > 0:
            per_call = min(per_call, int(CALL_GROUP_FEATURE_FRACTION * total_memory / fetched))
    return max(batch_size, per_call // batch_size * batch_size)


# ---------------------------------------------------------------------------------------------------------------------
# keeping the ranks of a partitioned FeatureStore in step (distributed_sampler.py:200-214,305-329)
# ---------------------------------------------------------------------------------------------------------------------
def _store_is_collective(feature_store) -> bool:
    """Is a fetch from this store a collective (some tensor partitioned over more than one rank)?"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return False
    for attr in feature_store.get_all_tensor_attrs():
        t = feature_store[attr.group_name, attr.attr_name, None]
        group = t.get_comm() if hasattr(t, "get_comm") else None
        if dist.get_world_size(group) > 1:
            return True
    return False


class FetchPadder:
    """With a FeatureStore partitioned over the ranks every feature fetch is a collective, and a loader makes one per call
    group plus one per mini-batch outside a group.  Ranks whose seed shards differ in size would make different numbers of
    them and leave each other waiting — the reference pads its call groups with empties for the same reason
    (``num_call_groups`` MAX-reduced, distributed_sampler.py:305-329) and warns about uneven batch counts (:200-214).
    Here the loader states how many fetches of either kind it WILL make, the counts are MAX-reduced once per epoch, and the
    rank that runs out first issues empty fetches (zero ids, same attributes, same collectives) until it has made as many."""

    def __init__(self, feature_store, n_group_fetches: int, n_single_fetches: int, n_batches: int, empty_group_ctx=None,
                 hetero: bool = False):
        import torch.distributed as dist
        self.fs, self.hetero, self.empty_ctx = feature_store, hetero, empty_group_ctx
        self.groups_done = self.singles_done = 0
        self.max_groups, self.max_singles = int(n_group_fetches), int(n_single_fetches)
        self.active = _store_is_collective(feature_store)
        if not self.active:
            return
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        mine = torch.tensor([n_group_fetches, n_single_fetches, n_batches], dtype=torch.int64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)
        table = torch.stack(every).cpu()
        self.max_groups, self.max_singles = int(table[:, 0].max()), int(table[:, 1].max())
        if dist.get_rank() == 0 and bool((table[:, 2] != table[0, 2]).any()):
            import warnings
            warnings.warn("Not all ranks received the same number of batches. Ranks with fewer batches are padded with "
                          "empty feature fetches so the collective fetches stay matched; a training loop with its own "
                          "collectives (gradient all-reduce) may still hang on uneven inputs. Batches per rank: "
                          f"{table[:, 2].tolist()}.")

    def group_done(self):
        self.groups_done += 1

    def single_done(self):
        self.singles_done += 1

    def _empty_index(self):
        return torch.empty(0, dtype=torch.int64, device="cuda" if torch.cuda.is_available() else "cpu")

    def _attrs(self):
        """The attributes a real fetch of this loader touches: all of them (homogeneous), or those stored under a node /
        edge type of the sampled graph (heterogeneous: ``empty_ctx`` = {"nodes": types, "edges": types})."""
        for attr in self.fs.get_all_tensor_attrs():
            g = attr.group_name
            if self.hetero and g not in self.empty_ctx["edges" if isinstance(g, tuple) else "nodes"]:
                continue
            yield attr

    def pad_groups(self):
        """Called when this rank has no call group left (before its first single fetch, or at the end)."""
        while self.active and self.groups_done < self.max_groups:
            for attr in self._attrs():
                _fetch_rows_agreed(self.fs[attr.group_name, attr.attr_name, None], self._empty_index())
            self.groups_done += 1

    def pad_singles(self):
        while self.active and self.singles_done < self.max_singles:
            for attr in self._attrs():
                self.fs[attr.group_name, attr.attr_name, None][self._empty_index()]
            self.singles_done += 1

    def finish(self):
        self.pad_groups()
        self.pad_singles()


class _BatchStream:
    """The sampler's batch generator + the fetch plan of the epoch (read by SampleIterator)."""

    def __init__(self, gen, fetch_padder=None):
        self._gen, self.fetch_padder = gen, fetch_padder

    def __iter__(self):
        return self

    def __next__(self):
        return next(self._gen)


_TEMPORAL_OK = {
    # candidate edge (time t_e) of a vertex reached at time t_v qualifies when ...
    "strictly_increasing": lambda te, tv: te > tv,
    "monotonically_increasing": lambda te, tv: te >= tv,
    "strictly_decreasing": lambda te, tv: te < tv,
    "monotonically_decreasing": lambda te, tv: te <= tv,
}


def _temporal_hop(graph: CSRGraph, frontier, frontier_time, fan, seed, biased, comparison):
    """One hop restricted to the edges whose timestamp passes ``comparison`` against the time at which their
    frontier vertex was reached (seeds: the input time; later vertices: the time of the edge that discovered them —
    the walk semantics the reference's tests pin, tests/loader/test_neighbor_loader.py:946-1057).  Runs on the
    weighted kernel: the qualifying candidates get weight w (or 1), the others 0, A-Res then draws ``fan`` of the
    qualifying ones without replacement and zero-weight picks (rows with fewer than ``fan`` qualifying edges) are
    dropped.  The effective-weight scratch is a persistent zero array touched only at the frontier's CSR ranges."""
    if comparison not in _TEMPORAL_OK:
        raise ValueError(f"unknown temporal_comparison {comparison!r}; expected one of {sorted(_TEMPORAL_OK)}")
    dev = graph.col.device
    scratch = getattr(graph, "_temporal_scratch", None)
    if scratch is None or scratch.shape[0] != graph.col.shape[0]:
        scratch = torch.zeros(graph.col.shape[0], dtype=torch.float32, device=dev)
        graph._temporal_scratch = scratch
    f = frontier.long()
    start = graph.row_ptr[f]
    deg = graph.row_ptr[f + 1] - start
    total = int(deg.sum())
    if total == 0:
        z = torch.zeros(0, dtype=torch.int64, device=dev)
        return frontier[:0], z.int(), z
    first = torch.cumsum(deg, 0) - deg
    pos = torch.repeat_interleave(start - first, deg, output_size=total) + torch.arange(total, device=dev)
    t_v = torch.repeat_interleave(frontier_time.to(dev).long(), deg, output_size=total)
    ok = _TEMPORAL_OK[comparison](graph.time[pos], t_v)
    scratch[pos] = ok.float() * (graph.weight[pos] if biased else 1.0)
    try:
        off, nbr, lid, gid = wholegraph_ops.weighted_sample_without_replacement(
            graph.row_ptr, graph.col, scratch, frontier, int(fan), seed, True, True)
        keep = scratch[gid] > 0
    finally:
        scratch[pos] = 0
    return nbr[keep], lid[keep], gid[keep]


def neighbor_sample(graph: CSRGraph, seeds: torch.Tensor, fanout: Sequence[int], random_state: int,
                    biased: bool = False, disjoint: bool = False, seed_time=None, temporal_comparison=None,
                    with_replacement: bool = False):
    """One mini-batch.  Returns (node, row, col, edge, num_sampled_nodes, num_sampled_edges).

    ``disjoint``: every seed grows its own tree and a vertex belongs to at most ONE tree of the batch — the tree of
    the first sampled edge that reaches it (first appearance = the renumbering order); sampled edges that lead into
    another tree's vertex are dropped.  This is the behaviour the reference's tests pin for libcugraph's disjoint
    sampling (tests/loader/test_neighbor_loader.py:840-943: seeds {0, 1} that share their only neighbour 2 yield ONE
    edge and n_id {0, 1, 2}; the per-seed vertex sets of a batch never intersect)."""
    seeds = seeds.to(device=graph.row_ptr.device, dtype=graph.col.dtype)
    nodes = seeds
    tree = torch.arange(seeds.shape[0], device=seeds.device) if disjoint else None   # tree id of every node
    temporal = seed_time is not None
    node_time = seed_time.to(seeds.device).long() if temporal else None               # time every node was reached at
    frontier, f_start = seeds, 0
    rows, cols, edges = [], [], []
    num_nodes, num_edges = [int(seeds.shape[0])], []
    for k, fan in enumerate(fanout):
        if frontier.shape[0] == 0:
            num_edges.append(0)
            num_nodes.append(0)
            continue
        if temporal:
            nbr, lid, gid = _temporal_hop(graph, frontier, node_time[f_start:f_start + frontier.shape[0]], fan,
                                          hop_seed(random_state, k), biased, temporal_comparison)
        elif biased:
            off, nbr, lid, gid = wholegraph_ops.weighted_sample_without_replacement(
                graph.row_ptr, graph.col, graph.weight, frontier, int(fan), hop_seed(random_state, k), True, True)
            # libcugraph's biased sampling never returns a zero-weight edge, even when the row is
            # shorter than the fan-out (tests/loader/test_neighbor_loader.py:99-133); the WholeGraph
            # kernel copies short rows whole, so drop those edges here.
            keep = graph.weight[gid] > 0
            if not bool(keep.all()):
                nbr, lid, gid = nbr[keep], lid[keep], gid[keep]
        elif with_replacement:
            # `replace=True` (reference: forwarded to libcugraph, distributed_sampler.py:775-792): exactly `fan` picks
            # per vertex that has neighbours, repeats allowed
            off, nbr, lid, gid = wholegraph_ops.unweighted_sample_with_replacement(
                graph.row_ptr, graph.col, frontier, int(fan), hop_seed(random_state, k), True, True)
        else:
            off, nbr, lid, gid = wholegraph_ops.unweighted_sample_without_replacement(
                graph.row_ptr, graph.col, frontier, int(fan), hop_seed(random_state, k), True, True)
        new_nodes, mapping = graph_ops.append_unique(nodes, nbr, need_neighbor_raw_to_unique=True)
        mapping = mapping.long()
        src_row = lid.long() + f_start
        n_old, n_new = int(nodes.shape[0]), int(new_nodes.shape[0])
        if (disjoint or temporal) and n_new > n_old:
            # the FIRST edge that reaches a new vertex decides its tree (disjoint) and its time (temporal)
            e_ids = torch.arange(mapping.shape[0], device=mapping.device)
            first = torch.full((n_new - n_old,), mapping.shape[0], dtype=torch.int64, device=mapping.device)
            is_new = mapping >= n_old
            first.scatter_reduce_(0, mapping[is_new] - n_old, e_ids[is_new], reduce="amin")
            if temporal:
                node_time = torch.cat([node_time, graph.time[gid[first]]])
            if disjoint:
                tree = torch.cat([tree, tree[src_row][first]])
        if disjoint and nbr.shape[0] > 0:
            keep = tree[mapping] == tree[src_row]
            mapping, src_row, gid, nbr = mapping[keep], src_row[keep], gid[keep], nbr[keep]
        rows.append(mapping)
        cols.append(src_row)
        edges.append(graph.edge_id[gid])
        num_edges.append(int(nbr.shape[0]))
        num_nodes.append(int(new_nodes.shape[0] - nodes.shape[0]))
        f_start = int(nodes.shape[0])
        frontier = new_nodes[f_start:]
        nodes = new_nodes
    cat = (lambda xs: torch.cat(xs) if xs else torch.zeros(0, dtype=torch.int64, device=seeds.device))
    return nodes, cat(rows), cat(cols), cat(edges), num_nodes, num_edges


def _one_hop(graph: CSRGraph, frontier, fan, seed, biased, with_replacement=False):
    """(neighbours, row index in `frontier`, CSR slot) of one hop on one CSR; zero-weight edges dropped
    for biased sampling."""
    if with_replacement:
        off, nbr, lid, gid = wholegraph_ops.unweighted_sample_with_replacement(
            graph.row_ptr, graph.col, frontier, int(fan), seed, True, True)
        return nbr, lid, gid
    if biased:
        off, nbr, lid, gid = wholegraph_ops.weighted_sample_without_replacement(
            graph.row_ptr, graph.col, graph.weight, frontier, int(fan), seed, True, True)
        keep = graph.weight[gid] > 0
        if not bool(keep.all()):
            nbr, lid, gid = nbr[keep], lid[keep], gid[keep]
    else:
        off, nbr, lid, gid = wholegraph_ops.unweighted_sample_without_replacement(
            graph.row_ptr, graph.col, frontier, int(fan), seed, True, True)
    return nbr, lid, gid


def hetero_neighbor_sample(graphs, seed_type, seeds, fanout, random_state: int, biased: bool = False,
                           seed_time=None, temporal_comparison=None, disjoint: bool = False,
                           with_replacement: bool = False):
    """Heterogeneous PyG-style sampling: per hop, for every edge type (src_t, rel, dst_t) in sorted
    order, the frontier vertices of type ``dst_t`` draw up to ``fanout[etype][hop]`` in-neighbours of
    type ``src_t``; vertices first seen during a hop form the next hop's frontier of their type.
    Seeds of hop-h / edge-type-index-t calls are ``hop_seed(random_state, h * n_etypes + t)`` — the
    flat ``[hop * num_etypes + etype]`` indexing of the reference's fan-out array
    (loader/neighbor_loader.py:192-201).  Ids are TYPE-LOCAL throughout.  ``seeds`` may also be a dict
    ``{node type: ids}`` (``seed_type`` is then ignored): link prediction seeds both endpoint types at once.

    ``disjoint``: as in ``neighbor_sample`` — every seed grows its own tree across ALL node types, a vertex joins the tree
    of the first sampled edge that reaches it (edge types in sorted order inside a hop) and edges into another tree's
    vertex are dropped (trees are numbered over the seed types in sorted order).  ``with_replacement``: ``replace=True``.

    Returns (node{type}, row{etype}, col{etype}, edge{etype}, num_sampled_nodes{type}[hops+1],
    num_sampled_edges{etype}[hops])."""
    etypes = sorted(graphs.keys())
    dev = next(iter(graphs.values())).row_ptr.device
    seed_dict = seeds if isinstance(seeds, dict) else {seed_type: seeds}
    ntypes = sorted({t for et in etypes for t in (et[0], et[2])} | set(seed_dict))
    empty = lambda: torch.zeros(0, dtype=torch.int64, device=dev)  # noqa: E731
    node = {t: empty() for t in ntypes}
    for t, ids in seed_dict.items():
        node[t] = ids.to(device=dev, dtype=torch.int64)
    tree = None
    if disjoint:   # tree id of every vertex of every type; seeds of the (sorted) seed types are numbered consecutively
        tree, base = {}, 0
        for t in ntypes:
            tree[t] = torch.arange(base, base + int(node[t].shape[0]), device=dev)
            base += int(node[t].shape[0])
    temporal = seed_time is not None
    if temporal:   # seed_time: tensor (single seed type) or {type: tensor}
        st = seed_time if isinstance(seed_time, dict) else {next(iter(seed_dict)): seed_time}
        node_time = {t: (st[t].to(dev).long() if t in st else torch.zeros(0, dtype=torch.int64, device=dev))
                     for t in ntypes}
    frontier_start = {t: 0 for t in ntypes}                  # first row of the current frontier in node[t]
    n_hops = len(next(iter(fanout.values())))
    rows = {et: [] for et in etypes}
    cols = {et: [] for et in etypes}
    edges = {et: [] for et in etypes}
    num_nodes = {t: [int(node[t].shape[0])] for t in ntypes}
    num_edges = {et: [] for et in etypes}
    for h in range(n_hops):
        hop_begin = {t: int(node[t].shape[0]) for t in ntypes}   # vertices added from here on are next frontier
        for ti, et in enumerate(etypes):
            src_t, _, dst_t = et
            fan = fanout.get(et, [0] * n_hops)[h]
            frontier = node[dst_t][frontier_start[dst_t]:hop_begin[dst_t]]
            if fan == 0 or frontier.shape[0] == 0:
                num_edges[et].append(0)
                continue
            if temporal:
                nbr, lid, gid = _temporal_hop(graphs[et], frontier, node_time[dst_t][frontier_start[dst_t]:hop_begin[dst_t]],
                                              fan, hop_seed(random_state, h * len(etypes) + ti), biased,
                                              temporal_comparison)
            else:
                nbr, lid, gid = _one_hop(graphs[et], frontier, fan, hop_seed(random_state, h * len(etypes) + ti), biased,
                                         with_replacement)
            n_old = int(node[src_t].shape[0])
            new_nodes, mapping = graph_ops.append_unique(node[src_t], nbr, need_neighbor_raw_to_unique=True)
            m = mapping.long()
            src_row = lid.long() + frontier_start[dst_t]
            if (temporal or disjoint) and new_nodes.shape[0] > n_old:
                first = torch.full((new_nodes.shape[0] - n_old,), m.shape[0], dtype=torch.int64, device=dev)
                is_new = m >= n_old
                first.scatter_reduce_(0, m[is_new] - n_old, torch.arange(m.shape[0], device=dev)[is_new], reduce="amin")
                if temporal:
                    node_time[src_t] = torch.cat([node_time[src_t], graphs[et].time[gid[first]]])
                if disjoint:   # the FIRST edge that reaches a new vertex decides its tree
                    tree[src_t] = torch.cat([tree[src_t], tree[dst_t][src_row[first]]])
            node[src_t] = new_nodes
            if disjoint and nbr.shape[0] > 0:
                keep = tree[src_t][m] == tree[dst_t][src_row]
                m, src_row, gid, nbr = m[keep], src_row[keep], gid[keep], nbr[keep]
            rows[et].append(m)
            cols[et].append(src_row)
            edges[et].append(graphs[et].edge_id[gid])
            num_edges[et].append(int(nbr.shape[0]))
        for t in ntypes:
            num_nodes[t].append(int(node[t].shape[0]) - hop_begin[t])
            frontier_start[t] = hop_begin[t]
    cat = lambda xs: torch.cat(xs) if xs else empty()  # noqa: E731
    return (node, {et: cat(rows[et]) for et in etypes}, {et: cat(cols[et]) for et in etypes},
            {et: cat(edges[et]) for et in etypes}, num_nodes, num_edges)


class HeteroNeighborSampler:
    """Heterogeneous counterpart of ``NeighborSampler`` (per-edge-type CSRs, dict fan-out)."""

    def __init__(self, graphs, fanout, biased: bool = False, with_replacement: bool = False,
                 disjoint: bool = False, temporal: bool = False, temporal_comparison: Optional[str] = None,
                 local_seeds_per_call: Optional[int] = None, num_nodes=None, **_ignored):
        self.local_seeds_per_call = local_seeds_per_call
        self.feature_row_bytes = (0, 0)   # (node, edge) row bytes of the FeatureStore the loader joins; set by BaseSampler
        self.num_nodes = num_nodes       # {node type: count}, optional (enables the packed renumber table)
        self._walks = {}
        self._positive_weights = None
        if with_replacement and (biased or temporal):
            raise NotImplementedError("sampling with replacement is uniform and non-temporal")
        self.with_replacement, self.disjoint = bool(with_replacement), bool(disjoint)
        if temporal and any(g.time is None for g in graphs.values()):
            raise ValueError("temporal sampling needs a time attribute on every edge type (time_attr=...)")
        self.temporal = bool(temporal)
        self.temporal_comparison = temporal_comparison or "monotonically_decreasing"
        n_hops = {len(v) for v in fanout.values()}
        if len(n_hops) != 1:
            raise ValueError("every edge type needs the same number of hops")
        if biased and any(g.weight is None for g in graphs.values()):
            raise ValueError("biased sampling needs a weight attribute on every edge type")
        self.graphs, self.fanout, self.biased = graphs, {k: [int(f) for f in v] for k, v in fanout.items()}, biased

    def _call_group_walk(self, batch_size: int, n_batches: int):
        from wholegraph_amd.fused import HeteroPygWalk
        key = (batch_size, n_batches)
        if key not in self._walks:
            # (the readers slice every list by the sizes read back: no -1 padding of the capacity slack)
            self._walks[key] = HeteroPygWalk(self.graphs, batch_size, self.fanout, n_batches, biased=self.biased,
                                             num_nodes=self.num_nodes, pad_unique=False)
        return self._walks[key]

    def call_groups_ok(self) -> bool:
        """Can this configuration run on the no-host-sync call-group kernels (see ``sample_batches``)?"""
        if self.biased and self._positive_weights is None:
            self._positive_weights = all(bool((g.weight > 0).all()) for g in self.graphs.values())
        biased_ok = (not self.biased) or (self._positive_weights and all(f <= 256 for v in self.fanout.values() for f in v))
        return biased_ok and (not self.temporal) and (not self.disjoint) and (not self.with_replacement) and all(
            g.col.dtype == torch.int64 for g in self.graphs.values())

    def _walk_capacity_per_seed(self):
        """(device bytes, largest node + edge slots of one call, node rows) ONE seed of a call group costs in the heterogeneous
        walk (``wholegraph_amd.fused.HeteroPygWalk``): its buffers are capacity-sized — a hop's frontier of a node type is
        everything the previous hop could have added to that type, an edge type's call holds frontier x fan-out edge slots —
        and all calls of a group stay alive until the group is consumed.  The worst seed type counts (the sampler does not
        know which type a loader seeds).  None when a fan-out is not positive (sample-all: no capacity bound)."""
        from wholegraph_amd import _lib as L
        hops = len(next(iter(self.fanout.values())))
        if any(f <= 0 for v in self.fanout.values() for f in v):
            return None
        types = sorted({t for et in self.graphs for t in (et[0], et[2])})
        worst = None
        for seed_type in types:
            gained = {t: 0 for t in types}
            cap = dict(gained)
            gained[seed_type] = cap[seed_type] = 1
            total, ws, slots = 0.0, 0.0, 1
            for h in range(hops):
                nxt = {t: 0 for t in types}
                for et in sorted(self.graphs):
                    fc = gained[et[2]]
                    if fc == 0 or et not in self.fanout:      # (an edge type without a fan-out entry is not sampled)
                        continue
                    ec, nc = fc * self.fanout[et][h], max(cap[et[0]], 1)
                    # per edge slot: local row / col + two scratch columns (int32), edge id, new node, its batch, next frontier
                    # entry, its batch; per node slot of the source type's list: id + batch; the frontier's offsets
                    total += ec * (4 * 4 + 8 + 8 + 4 + 8 + 4) + nc * (8 + 4) + (fc + 1) * 4
                    ws = max(ws, L.lib().wgamd_sample_hop_workspace_bytes(4096 * max(nc, fc), 4096 * ec, L.DT_INT64) / 4096.0)
                    slots = max(slots, nc + ec)
                    cap[et[0]] += ec
                    nxt[et[0]] += ec
                gained = nxt
            mine = (total + ws, slots, sum(cap.values()))
            worst = mine if worst is None or mine[0] > worst[0] else worst
        return worst

    def seeds_per_call(self, batch_size: int) -> int:
        """``local_seeds_per_call`` as given, else sized from device memory: the reference sums the per-hop fan-outs over the
        edge types as if they were one homogeneous hop (distributed_sampler.py:848-856 — for ogbn-mag's 6 edge types at [25, 10]
        that prices a seed at 9,000 hop-2 edges, 7x what this walk's buffers can hold, and gives groups of 5 mini-batches);
        here a seed is priced at the capacity the heterogeneous walk really allocates for it (``_walk_capacity_per_seed``),
        within the same memory fraction — about 50 mini-batches of 1024 seeds for that configuration on 288 GB."""
        if self.local_seeds_per_call:
            return int(self.local_seeds_per_call)
        cost = self._walk_capacity_per_seed() if not self.disjoint else None
        if cost is None:
            hops = len(next(iter(self.fanout.values())))
            per_hop = [sum(max(v[h], 0) for v in self.fanout.values()) if all(v[h] > 0 for v in self.fanout.values()) else -1
                       for h in range(hops)]
            return default_local_seeds_per_call(per_hop, batch_size, 8, self.disjoint, feature_row_bytes=self.feature_row_bytes)
        total_memory = (torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory
                        if torch.cuda.is_available() else 16 << 30)
        per_call = int(CALL_GROUP_MEMORY_FRACTION * total_memory / cost[0])
        per_call = min(per_call, _CALL_GROUP_CAPACITY // max(cost[1], 1))
        fetched = cost[2] * int(self.feature_row_bytes[0]) + max(cost[2] - 1, 0) * int(self.feature_row_bytes[1])
        if fetched > 0:
            per_call = min(per_call, int(CALL_GROUP_FEATURE_FRACTION * total_memory / fetched))
        return max(batch_size, per_call // batch_size * batch_size)

    def fetch_plan(self, n: int, batch_size: int, on_device: bool = True):
        """(call groups, batches outside a group, batches) ``sample_batches`` will produce for ``n`` seeds."""
        n_batches = -(-n // batch_size) if n else 0
        n_full = n // batch_size if (self.call_groups_ok() and on_device) else 0
        G = max(1, self.seeds_per_call(batch_size) // batch_size)
        return -(-n_full // G), n_batches - n_full, n_batches

    def sample_seed_lists(self, seed_lists, n_batches: int, random_state: int):
        """One call group over RAGGED per-batch seed lists of one or two node types (``seed_lists`` as in
        ``HeteroPygWalk.run``): batch j gets exactly ``hetero_neighbor_sample(graphs, None, its lists, fanout,
        random_state + j)``.  Returns (list of per-batch tuples, group context for ``_group_attribute_views``)."""
        n_et, hops = len(self.graphs), len(next(iter(self.fanout.values())))
        walk = self._call_group_walk(max(int(v[0].shape[0]) for v in seed_lists.values()) // max(n_batches, 1) or 1, n_batches)
        rs = torch.tensor([[_as_i64(hop_seed(random_state + j, k)) for j in range(n_batches)] for k in range(hops * n_et)],
                          dtype=torch.int64)
        rec = walk.run(None, None, rs, seed_lists=seed_lists)
        outs = walk.finalize_batches(rec)
        return outs, rec["group_context"]

    def sample_batches(self, seed_type, seeds, batch_size, random_state, seed_time=None):
        """Uniform, non-temporal sampling of full mini-batches runs in CALL GROUPS of ``local_seeds_per_call`` seeds
        on the batched no-sync kernel (one launch sequence per hop and edge type for the whole group); everything else
        goes one batch at a time through the C-ABI ops.  Both routes return identical results."""
        if self.temporal and seed_time is None:
            raise ValueError("temporal sampling needs input_time")
        n = seeds.shape[0]
        if self.biased and self._positive_weights is None:   # see NeighborSampler.sample_batches
            self._positive_weights = all(bool((g.weight > 0).all()) for g in self.graphs.values())
        biased_ok = (not self.biased) or (self._positive_weights and all(f <= 256 for v in self.fanout.values() for f in v))
        fast = biased_ok and (not self.temporal) and (not self.disjoint) and (not self.with_replacement) and seeds.is_cuda and all(
            g.col.dtype == torch.int64 for g in self.graphs.values())
        n_full = n // batch_size if fast else 0
        G = max(1, self.seeds_per_call(batch_size) // batch_size)
        n_et, hops = len(self.graphs), len(next(iter(self.fanout.values())))
        b = 0
        while b < n_full:
            g = min(G, n_full - b)
            walk = self._call_group_walk(batch_size, g)
            rs = torch.tensor([[_as_i64(hop_seed(random_state + b + j, k)) for j in range(g)] for k in range(hops * n_et)],
                              dtype=torch.int64)
            rec = walk.run(seed_type, seeds[b * batch_size:(b + g) * batch_size].to(torch.int64).contiguous(), rs)
            outs = walk.finalize_batches(rec)
            for j, out in enumerate(outs):
                self.current_group = (rec["group_context"], j)   # read by the consumer right after the yield
                yield b + j, out
            self.current_group = None
            b += g
        self.current_group = None
        for bb, start in enumerate(range(n_full * batch_size, n, batch_size), start=n_full):
            yield bb, hetero_neighbor_sample(
                self.graphs, seed_type, seeds[start:start + batch_size], self.fanout, random_state + bb, self.biased,
                seed_time[start:start + batch_size] if self.temporal else None, self.temporal_comparison,
                self.disjoint, self.with_replacement)


class NeighborSampler:
    """The role of ``DistributedNeighborSampler`` for the homogeneous case: owns the CSR, the fan-out
    and the flags; ``sample_batches`` is the hot loop."""

    def __init__(self, graph: CSRGraph, fanout: Sequence[int], biased: bool = False,
                 with_replacement: bool = False, disjoint: bool = False, heterogeneous: bool = False,
                 temporal: bool = False, local_seeds_per_call: Optional[int] = None,
                 temporal_comparison: Optional[str] = None, **_ignored):
        if with_replacement and (biased or temporal):
            raise NotImplementedError("sampling with replacement is uniform and non-temporal")
        self.with_replacement = bool(with_replacement)
        if heterogeneous:
            raise NotImplementedError("heterogeneous graphs go through HeteroNeighborSampler")
        if temporal and graph.time is None:
            raise ValueError("temporal sampling needs a time attribute (time_attr=...)")
        self.temporal = bool(temporal)
        self.temporal_comparison = temporal_comparison or "monotonically_decreasing"
        if biased and graph.weight is None:
            raise ValueError("biased sampling needs a weight attribute (weight_attr=...)")
        self.graph, self.fanout, self.biased, self.disjoint = graph, [int(f) for f in fanout], biased, bool(disjoint)
        self.local_seeds_per_call = local_seeds_per_call
        self.feature_row_bytes = (0, 0)   # (node, edge) row bytes of the FeatureStore the loader joins; set by BaseSampler
        self._walks = {}
        self._positive_weights = None

    def _call_group_walk(self, batch_size: int, n_batches: int):
        from wholegraph_amd.fused import PygNoSyncWalk
        key = (batch_size, n_batches)
        if key not in self._walks:
            self._walks[key] = PygNoSyncWalk(self.graph.row_ptr, self.graph.col, batch_size, self.fanout, n_batches,
                                             csr_weight=self.graph.weight if self.biased else None, pad_unique=False)
        return self._walks[key]

    def call_groups_ok(self) -> bool:
        """Can this configuration run on the no-host-sync call-group kernels (see ``sample_batches``)?"""
        if self.biased and self._positive_weights is None:
            self._positive_weights = bool((self.graph.weight > 0).all())
        biased_ok = (not self.biased) or (self._positive_weights and all(f <= 256 for f in self.fanout))
        return (biased_ok and (not self.disjoint) and (not self.temporal) and (not self.with_replacement)
                and all(f > 0 for f in self.fanout))

    def seeds_per_call(self, batch_size: int) -> int:
        """``local_seeds_per_call`` as given, else sized from device memory (``default_local_seeds_per_call``)."""
        if self.local_seeds_per_call:
            return int(self.local_seeds_per_call)
        return default_local_seeds_per_call(self.fanout, batch_size, self.graph.col.element_size(), self.disjoint,
                                            feature_row_bytes=self.feature_row_bytes)

    def fetch_plan(self, n: int, batch_size: int, on_device: bool = True):
        """(call groups, batches outside a group, batches) ``sample_batches`` will produce for ``n`` seeds."""
        n_batches = -(-n // batch_size) if n else 0
        n_full = n // batch_size if (self.call_groups_ok() and on_device) else 0
        G = max(1, self.seeds_per_call(batch_size) // batch_size)
        return -(-n_full // G), n_batches - n_full, n_batches

    def sample_seed_lists(self, seeds: torch.Tensor, seed_seg: torch.Tensor, seed_batch: torch.Tensor, max_seeds: int,
                          n_batches: int, random_state: int):
        """One call group over RAGGED per-batch seed lists (``seeds`` = the lists back to back, padded to
        ``n_batches * max_seeds``; ``seed_seg`` / ``seed_batch`` as in ``PygNoSyncWalk.run``): batch j gets exactly
        ``neighbor_sample(graph, its list, fanout, random_state + j)``.  Returns (list of per-batch tuples, group context
        for ``group_attribute_views``)."""
        walk = self._call_group_walk(max_seeds, n_batches)
        rs = [[hop_seed(random_state + j, k) for j in range(n_batches)] for k in range(len(self.fanout))]
        res = walk.run(seeds.to(self.graph.col.dtype).contiguous(), rs, seed_seg, seed_batch)
        outs = res.finalize_batches(self.graph.edge_id)
        return outs, res.group_context

    def sample_batches(self, seeds: torch.Tensor, batch_size: int, random_state: int, seed_time=None) -> Iterator:
        """Yields ``(batch index, (node, row, col, edge, num_sampled_nodes, num_sampled_edges))``.

        Uniform and biased (strictly positive weights, fan-outs <= 256) sampling with positive fan-outs runs in CALL
        GROUPS (``local_seeds_per_call`` seeds per launch sequence, by default sized from the device memory like the
        reference's, ``default_local_seeds_per_call`` — the reference splits its seeds the same way,
        sampler/distributed_sampler.py:391-410) on the no-host-sync kernels; everything else
        (zero weights, fan-out -1, disjoint / temporal, the ragged last batch) goes through the one-batch-at-a-time C-ABI
        ops.  Both routes return identical results (tests/test_gpu_pyg_loader.py)."""
        n = seeds.shape[0]
        if self.temporal and seed_time is None:
            raise ValueError("temporal sampling needs input_time")
        # biased call groups need strictly positive weights (the kernel copies short rows whole, zero-weight edges
        # included, and libcugraph never returns those: the one-batch path filters them) and fan-outs <= 256
        if self.biased and self._positive_weights is None:
            self._positive_weights = bool((self.graph.weight > 0).all())
        biased_ok = (not self.biased) or (self._positive_weights and all(f <= 256 for f in self.fanout))
        fast = (biased_ok and (not self.disjoint) and (not self.temporal) and (not self.with_replacement)
                and all(f > 0 for f in self.fanout) and seeds.is_cuda)
        n_full = n // batch_size if fast else 0
        G = max(1, self.seeds_per_call(batch_size) // batch_size)
        seeds = seeds.to(self.graph.col.dtype)
        b = 0
        while b < n_full:
            g = min(G, n_full - b)
            walk = self._call_group_walk(batch_size, g)
            rs = [[hop_seed(random_state + b + j, k) for j in range(g)] for k in range(len(self.fanout))]
            res = walk.run(seeds[b * batch_size:(b + g) * batch_size].contiguous(), rs)
            outs = res.finalize_batches(self.graph.edge_id)
            for j, out in enumerate(outs):
                self.current_group = (res.group_context, j)   # read by the consumer right after the yield
                yield b + j, out
            self.current_group = None
            b += g
        self.current_group = None
        for bb, start in enumerate(range(n_full * batch_size, n, batch_size), start=n_full):
            yield bb, neighbor_sample(self.graph, seeds[start:start + batch_size], self.fanout, random_state + bb,
                                      self.biased, self.disjoint,
                                      seed_time[start:start + batch_size] if self.temporal else None,
                                      self.temporal_comparison, self.with_replacement)


class BaseSampler:
    """``sample_from_nodes(NodeSamplerInput)`` → iterator of ``SamplerOutput`` (sampler.py:756-797)."""

    def __init__(self, sampler: NeighborSampler, data, batch_size: int = 16):
        self.__sampler = sampler
        self.__feature_store, self.__graph_store = data
        self.__batch_size = batch_size
        # the call-group size also answers for the rows a group fetches (sampler.default_local_seeds_per_call)
        if self.__feature_store is not None and hasattr(sampler, "feature_row_bytes"):
            sampler.feature_row_bytes = store_row_bytes(self.__feature_store)

    @property
    def core(self):
        """The sampler that does the work (``NeighborSampler`` / ``HeteroNeighborSampler``)."""
        return self.__sampler

    def sample_from_nodes(self, index: NodeSamplerInput, random_state: int = 62, **kwargs) -> Iterator[SamplerOutput]:
        """Iterator of ``SamplerOutput`` (sampler.py:756-797).  It also carries the epoch's FETCH PLAN: how many feature
        fetches per call group / per single batch this rank will make, MAX-reduced over the ranks when the FeatureStore is
        partitioned, so that ``SampleIterator`` can pad a rank that runs out of batches early (``FetchPadder``)."""
        smp = self.__sampler
        hetero = isinstance(smp, HeteroNeighborSampler)
        n_groups, n_singles, n_batches = smp.fetch_plan(int(index.node.shape[0]), self.__batch_size, index.node.is_cuda)
        ctx = None
        if hetero:
            ntypes = {t for et in smp.graphs for t in (et[0], et[2])}
            ctx = {"nodes": ntypes, "edges": set(smp.graphs)}
        padder = FetchPadder(self.__feature_store, n_groups, n_singles, n_batches, ctx, hetero)
        return _BatchStream(self.__sample_from_nodes(index, random_state), padder)

    def __sample_from_nodes(self, index: NodeSamplerInput, random_state: int):
        nodes = index.node
        input_id = index.input_id
        bs = self.__batch_size
        if isinstance(self.__sampler, HeteroNeighborSampler):
            it = index.input_type
            for b, (node, row, col, edge, nn, ne) in self.__sampler.sample_batches(it, nodes, bs, random_state,
                                                                                   seed_time=index.time):
                n_seeds = nn[it][0]
                ids = input_id[b * bs: b * bs + n_seeds]
                out = HeteroSamplerOutput(
                    node=node, row=row, col=col, edge=edge, batch={it: node[it][:n_seeds]},
                    num_sampled_nodes={k: torch.tensor(v) for k, v in nn.items()},
                    num_sampled_edges={k: torch.tensor(v) for k, v in ne.items()},
                    metadata=((it, ids), None if index.time is None else index.time[b * bs: b * bs + n_seeds]))
                # (call-group context, index of the batch in it) when the batch came out of a call group
                out._call_group = getattr(self.__sampler, "current_group", None)
                yield out
            return
        for b, (node, row, col, edge, nn, ne) in self.__sampler.sample_batches(nodes, bs, random_state,
                                                                               seed_time=index.time):
            n_seeds = nn[0]
            ids = input_id[b * bs: b * bs + n_seeds]
            grp = getattr(self.__sampler, "current_group", None)
            ready = grp is not None and "num_sampled_nodes" in grp[0]
            out = SamplerOutput(
                node=node, row=row, col=col, edge=edge, batch=node[:n_seeds],
                num_sampled_nodes=grp[0]["num_sampled_nodes"][grp[1]] if ready else torch.tensor(nn),
                num_sampled_edges=grp[0]["num_sampled_edges"][grp[1]] if ready else torch.tensor(ne),
                metadata=(ids, None if index.time is None else index.time[b * bs: b * bs + n_seeds]))
            out._call_group = grp
            yield out

    def sample_from_edges(self, index, neg_sampling=None, random_state: int = 62, **kwargs) -> Iterator[SamplerOutput]:
        """``EdgeSamplerInput`` (+ optional negative sampling) -> iterator of ``SamplerOutput`` whose metadata is
        ``(input_id, edge_label_index, edge_label, seed_time)`` (sampler.py:799-896, decode :583-628): the endpoints of a
        batch's seed edges and of its negatives are de-duplicated, expanded like node seeds, and ``edge_label_index`` indexes
        the batch's ``node`` list.  Homogeneous graphs; typed edge seeds go through ``LinkNeighborLoader`` directly."""
        if isinstance(self.__sampler, HeteroNeighborSampler):
            raise NotImplementedError("typed edge seeds are served by cugraph_pyg_amd.loader.LinkNeighborLoader")
        from ..loader.link_loader import LinkLoader   # the batch machinery (negatives, de-duplication, call groups) lives there
        loader = LinkLoader((self.__feature_store, self.__graph_store), self.__sampler,
                            edge_label_index=torch.stack([torch.as_tensor(index.row), torch.as_tensor(index.col)]),
                            edge_label=index.label, edge_label_time=index.time, neg_sampling=neg_sampling,
                            input_id=index.input_id, batch_size=self.__batch_size, shuffle=False, random_state=random_state,
                            as_sampler_output=True, **kwargs)
        yield from loader


def filter_store(feature_store, graph_store, node, row, col, edge) -> Data:
    """``filter_cugraph_pyg_store`` (sampler/sampler_utils.py:40-63): edge_index + every stored
    node attribute gathered at ``node`` (edge attributes at ``edge``)."""
    data = Data()
    data.edge_index = torch.stack([row, col], dim=0)
    for attr in feature_store.get_all_tensor_attrs():
        is_edge_attr = isinstance(attr.group_name, tuple)
        index = edge if is_edge_attr else node
        data[attr.attr_name] = feature_store[attr.group_name, attr.attr_name, None][index]
        if not is_edge_attr:
            data.num_nodes = index.size(0)
    return data


_GROUP_FETCH_BYTES = 4 << 30


from wholegraph_amd.pool import GrowOnlyPool as _GroupRowsPool  # noqa: E402  (the pool moved next to its second user, nn.py)


_group_rows = _GroupRowsPool()


def _fetch_rows_agreed(t, index):
    """``t[index]`` for a whole call group, in pieces of at most _GROUP_FETCH_BYTES, into a buffer of ``_group_rows``.  With a
    multi-rank tensor every ``t[...]`` is a collective, so the NUMBER of pieces must be the same on every rank of the
    tensor's group: it is MAX-reduced over that group first (ranks see different frontier sizes; a per-rank decision would
    pair one fetch on rank A with several on rank B and hang or mis-route rows).  Single-rank tensors decide locally."""
    import torch.distributed as dist
    row_bytes = torch.empty((), dtype=t.dtype).element_size()
    for d in tuple(t.shape)[1:]:
        row_bytes *= int(d)
    pieces = max(1, -(-int(index.numel()) * row_bytes // _GROUP_FETCH_BYTES))
    group = t.get_comm() if hasattr(t, "get_comm") else None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        worst = torch.tensor([pieces], dtype=torch.int64, device=dev)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX, group=group)
        pieces = int(worst.item())
    if hasattr(t, "gather_into"):
        # ONE output for the whole group, filled piece by piece (a torch.cat of pieces would hold the group's rows twice)
        out = _group_rows.take((int(index.numel()),) + tuple(t.shape)[1:], t.dtype, index.device)
        at = 0
        for part in ([index] if pieces == 1 else torch.tensor_split(index, pieces)):
            t.gather_into(part, out[at:at + part.numel()])
            at += part.numel()
        return out
    if pieces == 1:
        return t[index]
    parts = torch.tensor_split(index, pieces)
    first = t[parts[0]]
    out = torch.empty((int(index.numel()),) + tuple(first.shape)[1:], dtype=first.dtype, device=first.device)
    out[:parts[0].numel()] = first
    del first
    at = parts[0].numel()
    for part in parts[1:]:
        out[at:at + part.numel()] = t[part]
        at += part.numel()
    return out


def _group_attribute_views(feature_store, ctx):
    """Every stored attribute gathered ONCE for a whole call group (``ctx`` of HeteroPygWalk.finalize_batches), split into
    per-batch views."""
    views = {}
    for attr in feature_store.get_all_tensor_attrs():
        g = attr.group_name
        src = ctx["edges"] if isinstance(g, tuple) else ctx["nodes"]
        if g not in src:
            continue
        index, sizes = src[g]
        views[g, attr.attr_name] = torch.split(_fetch_rows_agreed(feature_store[g, attr.attr_name, None], index), sizes)
    return views


def group_attribute_views(feature_store, ctx):
    """One feature fetch per CALL GROUP instead of one per mini-batch (homogeneous graphs): the batches of a group are
    consecutive segments of one node list / one edge list (``ctx`` of PygWalkResult.finalize_batches), so every stored
    attribute is gathered once and split into per-batch views."""
    views = {}
    for attr in feature_store.get_all_tensor_attrs():
        is_edge = isinstance(attr.group_name, tuple)
        index, sizes = (ctx["edges"], ctx["edge_sizes"]) if is_edge else (ctx["nodes"], ctx["node_sizes"])
        views[attr.group_name, attr.attr_name] = torch.split(
            _fetch_rows_agreed(feature_store[attr.group_name, attr.attr_name, None], index), sizes)
    return views


def filter_store_from_group(feature_store, views, j, node, row, col, edge, ctx=None) -> Data:
    """``filter_store`` for batch j of a call group whose attributes were fetched by ``group_attribute_views``.  ``ctx``: the
    group's context — its ``edge_index`` / ``csr`` lists hold every batch's edge list and destination-major CSR as views of
    arrays made once for the group."""
    data = Data()
    ready = ctx.get("edge_index") if ctx is not None else None
    data.edge_index = ready[j] if ready is not None else torch.stack([row, col], dim=0)
    # a call-group walk emits hop after hop, a hop's edges in the CSR order of its frontier, every hop's destinations behind
    # the previous hop's: destination-major (wholegraph_amd.nn._to_csr then skips its sort)
    data.edge_index._wgamd_dst_sorted = data.edge_index._version     # (nn._to_csr: valid while the tensor is not edited in place)
    if ready is not None:
        data.edge_index._wgamd_csr = (data.edge_index._version, int(node.size(0))) + ctx["csr"][j]
    for attr in feature_store.get_all_tensor_attrs():
        is_edge = isinstance(attr.group_name, tuple)
        v = views[attr.group_name, attr.attr_name]
        data[attr.attr_name] = (v[j] if v is not None else
                                feature_store[attr.group_name, attr.attr_name, None][edge if is_edge else node])
        if not is_edge:
            data.num_nodes = node.size(0)
    return data


def build_hetero_data(feature_store, s: HeteroSamplerOutput, group_views=None, j: int = 0) -> HeteroData:
    """HeteroSamplerOutput -> HeteroData with every stored attribute joined (sampler.py:96-165).  ``group_views`` (from
    _group_attribute_views) + the batch's index in its call group replace the per-batch gathers."""
    data = HeteroData()
    for et in s.row:
        data[et].edge_index = torch.stack([s.row[et], s.col[et]], dim=0)
        data[et].e_id = s.edge[et].to(torch.long)
    for nt, ids in s.node.items():
        data[nt].n_id = ids
        data[nt].num_nodes = ids.size(0)
    for attr in feature_store.get_all_tensor_attrs():
        g = attr.group_name
        v = group_views.get((g, attr.attr_name)) if group_views is not None else None
        if isinstance(g, tuple):
            if g in s.edge:
                data[g][attr.attr_name] = v[j] if v is not None else feature_store[g, attr.attr_name, None][s.edge[g]]
        elif g in s.node:
            data[g][attr.attr_name] = v[j] if v is not None else feature_store[g, attr.attr_name, None][s.node[g]]
    data.set_value_dict("batch", s.batch)
    data.set_value_dict("num_sampled_nodes", s.num_sampled_nodes)
    data.set_value_dict("num_sampled_edges", s.num_sampled_edges)
    if s.metadata is not None and s.metadata[0] is not None:
        input_type, input_id = s.metadata[0]
        data[input_type].input_id = input_id
        data[input_type].batch_size = input_id.size(0)
        data[input_type].seed_time = s.metadata[1]
    return data


class SampleIterator:
    """Joins features to sampler outputs and emits PyG ``Data`` (sampler.py:51-165)."""

    def __init__(self, data, output_iter: Iterator[SamplerOutput]):
        self.__feature_store, self.__graph_store = data
        self.__output_iter = output_iter
        # the epoch's fetch plan, when the batches come from BaseSampler.sample_from_nodes (see FetchPadder)
        self.__padder = getattr(output_iter, "fetch_padder", None)

    def __next_hetero(self, s):
        group = getattr(s, "_call_group", None)
        if group is None:
            return build_hetero_data(self.__feature_store, s)
        cache = getattr(self, "_SampleIterator__hetero_cache", None)
        if cache is None or cache[0] is not group[0]:
            cache = (group[0], _group_attribute_views(self.__feature_store, group[0]))
            self.__hetero_cache = cache
            if self.__padder is not None:
                self.__padder.group_done()
        return build_hetero_data(self.__feature_store, s, cache[1], group[1])

    def __filter_from_group(self, ctx, j, s) -> Data:
        cache = getattr(self, "_SampleIterator__group_cache", None)
        if cache is None or cache[0] is not ctx:
            cache = (ctx, group_attribute_views(self.__feature_store, ctx))
            self.__group_cache = cache
            if self.__padder is not None:
                self.__padder.group_done()
        return filter_store_from_group(self.__feature_store, cache[1], j, s.node, s.row, s.col, s.edge, ctx)

    def __next__(self):
        pad = self.__padder
        try:
            s = next(self.__output_iter)
        except StopIteration:
            if pad is not None:     # this rank is out of batches: match the fetches the others still make
                pad.finish()
            raise
        if pad is not None and getattr(s, "_call_group", None) is None:
            pad.pad_groups()        # batches outside a call group come last: the group phase of this rank is over
            pad.single_done()
        if isinstance(s, HeteroSamplerOutput):
            return self.__next_hetero(s)
        group = getattr(s, "_call_group", None)
        if group is not None:
            data = self.__filter_from_group(group[0], group[1], s)
        else:
            data = filter_store(self.__feature_store, self.__graph_store, s.node, s.row, s.col, s.edge)
        if "n_id" not in data:
            data.n_id = s.node
        if s.edge is not None and "e_id" not in data:
            data.e_id = s.edge.to(torch.long)
        data.batch = s.batch
        data.num_sampled_nodes = s.num_sampled_nodes
        data.num_sampled_edges = s.num_sampled_edges
        data.input_id = s.metadata[0]
        data.batch_size = data.input_id.size(0)
        if len(s.metadata) == 4:     # edge seeds (sampler.py:104-112)
            data.edge_label_index, data.seed_time = s.metadata[1], s.metadata[3]
            if s.metadata[2] is not None:
                data.edge_label = s.metadata[2]
        else:
            data.seed_time = s.metadata[1]
        return data

    def __iter__(self):
        return self
