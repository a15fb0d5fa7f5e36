"""Duck-typed ``torch_geometric.loader.LinkLoader`` / ``LinkNeighborLoader``: mini-batches seeded by
EDGES (link prediction), optional negative sampling.

Reference: loader/link_loader.py:17-234, loader/link_neighbor_loader.py, and the edge-seed path of the
sampler (/root/reference/python/cugraph-pyg/cugraph_pyg/sampler/sampler.py:799-896,
sampler/distributed_sampler.py:428-638, negative sampling sampler/sampler_utils.py:93-336).  Per batch:
the endpoints of the seed edges (plus negatives) are deduplicated in first-appearance order — the
renumbering kernel with an empty target list does exactly that and returns the inverse map, which IS
``edge_label_index`` — then the unique endpoints are expanded like node seeds.  Full batches (homogeneous graphs and typed
edge seeds of heterogeneous ones, uniform or biased, not temporal / disjoint) run in CALL GROUPS (``local_seeds_per_call``, by default sized from device memory): the endpoints of all batches are de-duplicated row-wise in one
pass, the ragged per-batch seed lists go through ONE no-host-sync walk (``NeighborSampler.sample_seed_lists``) and every
stored attribute is fetched once for the group; batch by batch the result equals the one-batch path
(``call_groups=False``; tests/test_gpu_pyg_loader.py).

Implemented: homogeneous and heterogeneous graphs (edge seeds of ONE edge type, endpoints of both node types
seeded together), ``neg_sampling`` = None | "binary" | "triplet" (uniform negatives inside the endpoint types'
id ranges, at least one per batch — sampler_utils.py:112-116; both modes deliver ``edge_label_index`` + ``edge_label``
with the positives first, as the reference's readers do, sampler.py:583-598 — "triplet" draws the negative sources from
the batch's positive sources, sampler.py:833-840 — here negative k starts at the source of positive k mod n_pos), ``disjoint``, temporal seeds (``edge_label_time`` + ``time_attr``);
with a node-level ``time_attr`` in the feature store the negatives are redrawn until both endpoints exist at the seed
edge's time (5 attempts, then the earliest node of the type: sampler_utils.py:213-311).
"""
from math import ceil
import warnings
from typing import Optional, Tuple

import torch

from wholegraph_amd import graph_ops

from ..data.graph_store import GraphStore
from ..sampler.sampler import (FetchPadder, HeteroNeighborSampler, NeighborSampler, build_hetero_data, filter_store,
                               filter_store_from_group, group_attribute_views, _group_attribute_views, hetero_neighbor_sample,
                               neighbor_sample)
from .._compat import HeteroSamplerOutput, SamplerOutput
from .node_loader import generate_seed


class _NoStore:
    """A feature store with nothing in it (sampler-output mode: the loader fetches no features)."""

    def get_all_tensor_attrs(self):
        return []


def _first_occurrence(inverse, values, n_unique):
    """values[first position i with inverse[i] == u] for every u in [0, n_unique)."""
    first = torch.full((n_unique,), inverse.numel(), dtype=torch.int64, device=inverse.device)
    first.scatter_reduce_(0, inverse, torch.arange(inverse.numel(), device=inverse.device), reduce="amin")
    return values[first]


def _node_time(feature_store, node_type, time_attr, n):
    """Per-node timestamps of ``node_type`` if the feature store holds ``(node_type, time_attr)``, else None."""
    if time_attr is None:
        return None
    try:
        t = feature_store[node_type, time_attr, None]
    except Exception:  # noqa: BLE001 - a missing attribute raises KeyError / AttributeError depending on the store
        return None
    if t is None:
        return None
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    t = t if torch.is_tensor(t) else t[torch.arange(n, device=dev)]
    return t.to(dev).long().view(-1)


def _draw_negatives(n_neg, num_src, num_dst, gen, dev, neg_time=None, src_time=None, dst_time=None):
    """Uniform (src, dst) pairs; with node times, pair k is redrawn until both ends are no later than neg_time[k]."""
    src = torch.randint(0, num_src, (n_neg,), generator=gen, device=dev)
    dst = torch.randint(0, num_dst, (n_neg,), generator=gen, device=dev)
    if neg_time is None or (src_time is None and dst_time is None) or n_neg == 0:
        return src, dst

    def late(ids, times):
        return torch.zeros_like(ids, dtype=torch.bool) if times is None else times[ids] > neg_time

    for _ in range(5):
        bad = late(src, src_time) | late(dst, dst_time)
        k = int(bad.sum())
        if k == 0:
            return src, dst
        src = torch.where(bad, torch.randint(0, num_src, (n_neg,), generator=gen, device=dev), src)
        dst = torch.where(bad, torch.randint(0, num_dst, (n_neg,), generator=gen, device=dev), dst)
    if src_time is not None:
        src = torch.where(late(src, src_time), src_time.argmin().expand_as(src), src)
    if dst_time is not None:
        dst = torch.where(late(dst, dst_time), dst_time.argmin().expand_as(dst), dst)
    return src, dst


def _batched_first_unique(ends: torch.Tensor, n_ids: int):
    """Row-wise first-appearance de-duplication of ``ends`` [G, S] (ids < n_ids), for a whole call group at once.
    -> (unique ids of all rows back to back, padded with zeros to G*S; int32 offsets [G+1]; int32 row of every live
    entry, padded; ``local`` [G, S] = position of every element in its row's unique list) — per row exactly what
    ``graph_ops.append_unique`` with an empty target list returns."""
    G, S = ends.shape
    dev = ends.device
    rows = torch.arange(G, device=dev).view(G, 1)
    uk, inv = torch.unique((rows * n_ids + ends).view(-1), return_inverse=True)       # sorted by (row, id)
    pos = torch.arange(G * S, device=dev)
    first = torch.full((uk.numel(),), G * S, dtype=torch.int64, device=dev).scatter_reduce_(0, inv, pos, reduce="amin")
    order = torch.argsort(first)                       # positions are row-major, so the rows stay grouped
    rank = torch.empty_like(order)
    rank[order] = torch.arange(order.numel(), device=dev)
    uk = uk[order]
    row_of = torch.div(uk, n_ids, rounding_mode="floor")
    seg = torch.zeros(G + 1, dtype=torch.int64, device=dev)
    seg[1:] = torch.cumsum(torch.bincount(row_of, minlength=G), 0)
    uniq = torch.zeros(G * S, dtype=torch.int64, device=dev)
    uniq[:uk.numel()] = uk - row_of * n_ids
    batch = torch.zeros(G * S, dtype=torch.int32, device=dev)
    batch[:uk.numel()] = row_of.to(torch.int32)
    local = (rank[inv].view(G, S) - seg[:-1].view(G, 1))
    return uniq, seg.to(torch.int32).contiguous(), batch, local


def _parse_neg_sampling(neg_sampling) -> Tuple[Optional[str], float]:
    if neg_sampling is None:
        return None, 0.0
    if isinstance(neg_sampling, str):
        return neg_sampling, 1.0
    if isinstance(neg_sampling, dict):
        return neg_sampling.get("mode", "binary"), float(neg_sampling.get("amount", 1))
    if isinstance(neg_sampling, (tuple, list)):
        return neg_sampling[0], float(neg_sampling[1])
    mode = getattr(neg_sampling, "mode", None)          # torch_geometric.sampler.NegativeSampling
    return getattr(mode, "value", mode), float(getattr(neg_sampling, "amount", 1))


class LinkLoader:
    def __init__(self, data, link_sampler: NeighborSampler, edge_label_index=None, edge_label=None,
                 edge_label_time=None, neg_sampling=None, neg_sampling_ratio=None, transform=None,
                 transform_sampler_output=None, filter_per_worker=None, custom_cls=None, input_id=None,
                 batch_size: int = 1, shuffle: bool = False, drop_last: bool = False,
                 random_state: Optional[int] = None, time_attr: Optional[str] = None, call_groups: bool = True,
                 as_sampler_output: bool = False, **kwargs):
        if not isinstance(data, (list, tuple)) or not isinstance(data[1], GraphStore):
            raise NotImplementedError("Currently can't accept non-cugraph graphs")
        for name, val in (("filter_per_worker", filter_per_worker), ("custom_cls", custom_cls), ("transform", transform),
                          ("transform_sampler_output", transform_sampler_output)):
            if val:           # same notice NodeLoader gives: these hooks are accepted for signature parity, not applied
                warnings.warn(f"{name} is currently ignored")
        if kwargs:
            warnings.warn("ignored LinkLoader arguments: %s" % sorted(kwargs))
        self.__time_attr = time_attr
        # False: one batch at a time through the C-ABI ops (tuning / tests); the call-group walk only exists on the GPU
        self.__call_groups = bool(call_groups) and torch.cuda.is_available()
        # True: yield torch_geometric-style SamplerOutput objects (metadata = (input_id, edge_label_index, edge_label,
        # seed_time), the reference's sampler.py:621-628) instead of Data — what BaseSampler.sample_from_edges hands to a
        # SampleIterator; features are then joined there
        self.__raw = bool(as_sampler_output)
        if not isinstance(link_sampler, (NeighborSampler, HeteroNeighborSampler)):
            raise NotImplementedError("Must provide a cuGraph sampler")
        if neg_sampling_ratio is not None:
            warnings.warn("neg_sampling_ratio is deprecated; use neg_sampling=('binary', ratio)")
            neg_sampling = ("binary", float(neg_sampling_ratio))
        self.__mode, self.__amount = _parse_neg_sampling(neg_sampling)
        if self.__mode not in (None, "binary", "triplet"):
            raise ValueError(f"unknown negative sampling mode {self.__mode!r}")
        graph_store = data[1]
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        self.__etype = None
        if isinstance(edge_label_index, (tuple, list)) and len(edge_label_index) == 2 and not torch.is_tensor(
                edge_label_index[0]) and isinstance(edge_label_index[0], (tuple, list)):
            self.__etype = tuple(edge_label_index[0])         # (edge_type, tensor | None)
            edge_label_index = edge_label_index[1]
        self.__hetero = isinstance(link_sampler, HeteroNeighborSampler)
        if self.__hetero and self.__etype is None:
            raise ValueError("heterogeneous graphs need edge_label_index=(edge_type, 2 x N tensor)")
        if edge_label_index is None:                          # all edges of the (given) type
            et = self.__etype or graph_store.get_all_edge_attrs()[0].edge_type
            edge_label_index = graph_store.get_edge_index(et, "coo")
        if isinstance(edge_label_index, (tuple, list)):
            edge_label_index = torch.stack([torch.as_tensor(t) for t in edge_label_index])
        self.__eli = torch.as_tensor(edge_label_index).to(dev).long()
        if self.__eli.dim() != 2 or self.__eli.shape[0] != 2:
            raise ValueError("edge_label_index must be a 2 x N tensor")
        n = self.__eli.shape[1]
        self.__label = None if edge_label is None else torch.as_tensor(edge_label).to(dev)
        self.__time = None if edge_label_time is None else torch.as_tensor(edge_label_time).to(dev).long()
        if getattr(link_sampler, "temporal", False) and self.__time is None:
            raise ValueError("temporal link sampling needs edge_label_time")
        self.__input_id = torch.arange(n, device=dev) if input_id is None else torch.as_tensor(input_id).to(dev)
        if n < batch_size and drop_last:
            raise ValueError("The number of input edges is less than the batch size and drop_last is True.")
        self.__data, self.__sampler = data, link_sampler
        if data[0] is not None and hasattr(link_sampler, "feature_row_bytes"):   # the group fetch counts in the group size
            from ..sampler.sampler import store_row_bytes
            link_sampler.feature_row_bytes = store_row_bytes(data[0])
        self.__batch_size, self.__shuffle, self.__drop_last = batch_size, shuffle, drop_last
        self.__random_state = random_state
        nv = graph_store._num_vertices()
        if self.__hetero:
            self.__num_src, self.__num_dst = int(nv[self.__etype[0]]), int(nv[self.__etype[2]])
            types = (self.__etype[0], self.__etype[2])
        else:
            self.__num_nodes = graph_store._graph.num_vertices
            self.__num_src = self.__num_dst = self.__num_nodes
            types = (sorted(nv.keys())[0],) * 2
        # node-level timestamps gate the negatives of temporal link prediction (sampler_utils.py:213-241)
        self.__src_time = self.__dst_time = None
        if self.__mode is not None and self.__time is not None:
            self.__src_time = _node_time(data[0], types[0], time_attr, self.__num_src)
            self.__dst_time = self.__src_time if types[0] == types[1] else _node_time(data[0], types[1], time_attr,
                                                                                      self.__num_dst)

    def __with_negatives(self, src, dst, ix, gen):
        """(src_all, dst_all, n_neg, time_all): positives first, then the negatives of the batch."""
        n_pos, dev = ix.numel(), src.device
        t_pos = None if self.__time is None else self.__time[ix]
        if self.__mode is None:
            return src, dst, 0, t_pos
        n_neg = max(int(ceil(self.__amount * n_pos)), 1)                     # at least one negative per batch
        t_neg = None if t_pos is None else t_pos[torch.arange(n_neg, device=dev) % n_pos]
        neg_src, neg_dst = _draw_negatives(n_neg, self.__num_src, self.__num_dst, gen, dev, t_neg, self.__src_time,
                                           self.__dst_time)
        if self.__mode == "triplet":   # negative k starts at the source of "its" positive, k mod n_pos
            neg_src = src[torch.arange(n_neg, device=dev) % n_pos]
        t_all = None if t_pos is None else torch.cat([t_pos, t_neg])
        return torch.cat([src, neg_src]), torch.cat([dst, neg_dst]), n_neg, t_all

    def __emit(self, node, row, col, edge, nn, ne, ix, inverse, n_pos, n_neg, views=None, j=0, group=None):
        """One homogeneous mini-batch as ``Data`` (features joined) or, in sampler mode, as ``SamplerOutput``."""
        dev = self.__eli.device
        half = n_pos + n_neg
        edge_label_index = torch.stack([inverse[:half], inverse[half:]])
        if self.__mode is not None:
            pos = torch.ones(n_pos, device=dev) if self.__label is None else (self.__label[ix] + 1)
            edge_label = torch.cat([pos, torch.zeros(n_neg, device=dev, dtype=pos.dtype)])
        else:
            edge_label = None if self.__label is None else self.__label[ix]
        if self.__raw:
            out = SamplerOutput(node=node, row=row, col=col, edge=edge, batch=node[:nn[0]],
                                num_sampled_nodes=torch.tensor(nn), num_sampled_edges=torch.tensor(ne),
                                metadata=(self.__input_id[ix], edge_label_index, edge_label,
                                          None if self.__time is None else self.__time[ix]))
            out._call_group = group
            return out
        fs, gs = self.__data
        data = (filter_store_from_group(fs, views, j, node, row, col, edge) if views is not None
                else filter_store(fs, gs, node, row, col, edge))
        data.n_id, data.e_id = node, edge
        data.num_sampled_nodes, data.num_sampled_edges = torch.tensor(nn), torch.tensor(ne)
        data.input_id = self.__input_id[ix]
        data.batch_size = n_pos
        data.edge_label_index = edge_label_index
        if edge_label is not None:
            data.edge_label = edge_label
        if self.__mode == "triplet":   # PyG's triplet view of the same batch
            data.src_index, data.dst_pos_index = inverse[:n_pos], inverse[half:half + n_pos]
            neg = inverse[half + n_pos:]
            data.dst_neg_index = neg.view(-1, n_pos).t() if n_neg % n_pos == 0 and n_neg > n_pos else neg
        return data

    def __len__(self):
        n = self.__eli.shape[1]
        return n // self.__batch_size if self.__drop_last else (n + self.__batch_size - 1) // self.__batch_size

    def __batches(self):
        n = self.__eli.shape[1]
        dev = self.__eli.device
        perm = torch.randperm(n, device=dev) if self.__shuffle else torch.arange(n, device=dev)
        if self.__drop_last and n % self.__batch_size:
            perm = perm[: n - n % self.__batch_size]
        seed = self.__random_state if self.__random_state is not None else generate_seed()
        fs, gs = self.__data
        if self.__hetero:
            yield from self.__hetero_batches(perm, seed)
            return
        graph = self.__sampler.graph
        bs = self.__batch_size
        n_full = perm.numel() // bs if (self.__call_groups and self.__sampler.call_groups_ok()) else 0
        G = self.__batches_per_group(bs)
        pad = self.__padder(n_full, G, perm.numel(), bs)
        b0 = 0
        while b0 < n_full:   # CALL GROUPS of full batches: one launch sequence per hop for g mini-batches
            g = min(G, n_full - b0)
            yield from self.__group(perm, seed, b0, g)
            pad.group_done()
            b0 += g
        pad.pad_groups()     # a rank with fewer call groups than the others makes empty fetches until it has as many
        for b, start in enumerate(range(n_full * bs, perm.numel(), bs), start=n_full):
            pad.single_done()
            ix = perm[start:start + self.__batch_size]
            src, dst = self.__eli[0, ix], self.__eli[1, ix]
            n_pos = ix.numel()
            gen = torch.Generator(device=dev).manual_seed((seed + b) & 0x7FFFFFFFFFFFFFFF)
            src_all, dst_all, n_neg, t_all = self.__with_negatives(src, dst, ix, gen)
            ends = torch.cat([src_all, dst_all])
            # first-appearance dedup + inverse map = renumbering with no targets
            uniq, inverse = graph_ops.append_unique(ends[:0].contiguous(), ends.contiguous(),
                                                    need_neighbor_raw_to_unique=True)
            seed_time = None
            if self.__sampler.temporal:   # every endpoint of seed edge i (and of its negatives) starts at time i
                seed_time = _first_occurrence(inverse.long(), torch.cat([t_all, t_all]), uniq.numel())
            node, row, col, edge, nn, ne = neighbor_sample(graph, uniq, self.__sampler.fanout, seed + b,
                                                           self.__sampler.biased, self.__sampler.disjoint, seed_time,
                                                           self.__sampler.temporal_comparison,
                                                           getattr(self.__sampler, "with_replacement", False))
            yield self.__emit(node, row, col, edge, nn, ne, ix, inverse.long(), n_pos, n_neg)
        pad.pad_singles()

    def __batches_per_group(self, bs):
        """Mini-batches per call group: the sampler's seeds-per-call budget over the seeds a batch of ``bs`` seed edges
        really brings (both endpoints of the positives and of the negatives)."""
        n_neg = max(int(ceil(self.__amount * bs)), 1) if self.__mode is not None else 0
        seeds_per_batch = 2 * (bs + n_neg)
        return max(1, self.__sampler.seeds_per_call(seeds_per_batch) // seeds_per_batch)

    def __padder(self, n_full, G, n, bs):
        """The epoch's fetch plan (``FetchPadder``): call groups, batches outside a group, batches.  In sampler-output mode
        nothing is fetched here, so nothing is padded."""
        n_batches = -(-n // bs) if n else 0
        if self.__raw:
            return FetchPadder(_NoStore(), 0, 0, n_batches)
        ctx = None
        if self.__hetero:
            smp = self.__sampler
            ctx = {"nodes": {t for et in smp.graphs for t in (et[0], et[2])}, "edges": set(smp.graphs)}
        return FetchPadder(self.__data[0], -(-n_full // G), n_batches - n_full, n_batches, ctx, self.__hetero)

    def __group(self, perm, seed, b0, g):
        """Batches b0 .. b0+g-1 (all full) of a homogeneous graph as one call group: negatives per batch with the batch's
        own generator (same draws as the one-batch path), endpoints de-duplicated row-wise for the whole group, ONE walk
        over the ragged seed lists.  Yields the same ``Data`` objects as the loop below, batch by batch."""
        fs, gs = self.__data
        dev, bs = self.__eli.device, self.__batch_size
        ixs, ends, n_negs = [], [], []
        for j in range(g):
            ix = perm[(b0 + j) * bs:(b0 + j + 1) * bs]
            gen = torch.Generator(device=dev).manual_seed((seed + b0 + j) & 0x7FFFFFFFFFFFFFFF)
            src_all, dst_all, n_neg, _ = self.__with_negatives(self.__eli[0, ix], self.__eli[1, ix], ix, gen)
            ixs.append(ix)
            ends.append(torch.cat([src_all, dst_all]))
            n_negs.append(n_neg)
        ends = torch.stack(ends)                                    # [g, 2 * (bs + n_neg)]: every batch has the same n_neg
        S = ends.shape[1]
        uniq, seg, batch, local = _batched_first_unique(ends, self.__num_nodes)
        outs, ctx = self.__sampler.sample_seed_lists(uniq, seg, batch, S, g, seed + b0)
        views = None if self.__raw else group_attribute_views(fs, ctx)   # every stored attribute: one fetch for the group
        for j, (node, row, col, edge, nn, ne) in enumerate(outs):
            yield self.__emit(node, row, col, edge, nn, ne, ixs[j], local[j], bs, n_negs[j], views, j, (ctx, j))

    def __hetero_batches(self, perm, seed):
        """Edge seeds of one edge type (src_t, rel, dst_t): the src endpoints seed type src_t, the dst endpoints type
        dst_t (one joint list when both are the same type); ids are type-local (sampler.py:280-490 decode contract:
        ``edge_label_index`` under the seed edge type, local ids into ``n_id`` of the endpoint types)."""
        fs, gs = self.__data
        src_t, _, dst_t = self.__etype
        dev = self.__eli.device
        smp = self.__sampler
        bs = self.__batch_size
        n_full = perm.numel() // bs if (self.__call_groups and smp.call_groups_ok()) else 0
        G = self.__batches_per_group(bs)
        pad = self.__padder(n_full, G, perm.numel(), bs)
        b0 = 0
        while b0 < n_full:   # CALL GROUPS of full batches
            g = min(G, n_full - b0)
            yield from self.__hetero_group(perm, seed, b0, g)
            pad.group_done()
            b0 += g
        pad.pad_groups()
        for b, start in enumerate(range(n_full * bs, perm.numel(), bs), start=n_full):
            pad.single_done()
            ix = perm[start:start + self.__batch_size]
            src, dst = self.__eli[0, ix], self.__eli[1, ix]
            n_pos = ix.numel()
            gen = torch.Generator(device=dev).manual_seed((seed + b) & 0x7FFFFFFFFFFFFFFF)
            src_all, dst_all, n_neg, t_all = self.__with_negatives(src, dst, ix, gen)
            empty = src_all[:0].contiguous()
            if src_t == dst_t:
                uniq, inv = graph_ops.append_unique(empty, torch.cat([src_all, dst_all]).contiguous(),
                                                    need_neighbor_raw_to_unique=True)
                seeds = {src_t: uniq}
                inv_src, inv_dst = inv[:src_all.numel()].long(), inv[src_all.numel():].long()
            else:
                us, inv_src = graph_ops.append_unique(empty, src_all.contiguous(), need_neighbor_raw_to_unique=True)
                ud, inv_dst = graph_ops.append_unique(empty, dst_all.contiguous(), need_neighbor_raw_to_unique=True)
                seeds = {src_t: us, dst_t: ud}
                inv_src, inv_dst = inv_src.long(), inv_dst.long()
            seed_time = None
            if smp.temporal:
                t_src = t_dst = t_all
                if src_t == dst_t:
                    seed_time = {src_t: _first_occurrence(torch.cat([inv_src, inv_dst]), torch.cat([t_src, t_dst]),
                                                          seeds[src_t].numel())}
                else:
                    seed_time = {src_t: _first_occurrence(inv_src, t_src, seeds[src_t].numel()),
                                 dst_t: _first_occurrence(inv_dst, t_dst, seeds[dst_t].numel())}
            node, row, col, edge, nn, ne = hetero_neighbor_sample(smp.graphs, None, seeds, smp.fanout, seed + b, smp.biased,
                                                                  seed_time, smp.temporal_comparison,
                                                                  getattr(smp, "disjoint", False),
                                                                  getattr(smp, "with_replacement", False))
            out = HeteroSamplerOutput(node=node, row=row, col=col, edge=edge,
                                      batch={t: node[t][:v.numel()] for t, v in seeds.items()},
                                      num_sampled_nodes={k: torch.tensor(v) for k, v in nn.items()},
                                      num_sampled_edges={k: torch.tensor(v) for k, v in ne.items()}, metadata=None)
            yield self.__emit_hetero(build_hetero_data(fs, out), ix, inv_src, inv_dst, n_pos, n_neg)
        pad.pad_singles()

    def __emit_hetero(self, data, ix, inv_src, inv_dst, n_pos, n_neg):
        dev = self.__eli.device
        st = data[self.__etype]
        st.edge_label_index = torch.stack([inv_src, inv_dst])
        st.input_id = self.__input_id[ix]
        st.batch_size = n_pos
        if self.__mode is not None:
            pos = torch.ones(n_pos, device=dev) if self.__label is None else (self.__label[ix] + 1)
            st.edge_label = torch.cat([pos, torch.zeros(n_neg, device=dev, dtype=pos.dtype)])
            if self.__mode == "triplet":
                st.src_index, st.dst_pos_index, st.dst_neg_index = inv_src[:n_pos], inv_dst[:n_pos], inv_dst[n_pos:]
        elif self.__label is not None:
            st.edge_label = self.__label[ix]
        return data

    def __hetero_group(self, perm, seed, b0, g):
        """Batches b0 .. b0+g-1 (all full) of a heterogeneous graph as one call group (see ``__group``): the source
        endpoints seed type src_t, the destination endpoints type dst_t (one joint list when both are the same type)."""
        fs, gs = self.__data
        src_t, _, dst_t = self.__etype
        dev, bs, smp = self.__eli.device, self.__batch_size, self.__sampler
        ixs, srcs, dsts, n_negs = [], [], [], []
        for j in range(g):
            ix = perm[(b0 + j) * bs:(b0 + j + 1) * bs]
            gen = torch.Generator(device=dev).manual_seed((seed + b0 + j) & 0x7FFFFFFFFFFFFFFF)
            src_all, dst_all, n_neg, _ = self.__with_negatives(self.__eli[0, ix], self.__eli[1, ix], ix, gen)
            ixs.append(ix)
            srcs.append(src_all)
            dsts.append(dst_all)
            n_negs.append(n_neg)
        srcs, dsts = torch.stack(srcs), torch.stack(dsts)           # [g, bs + n_neg] each
        half = srcs.shape[1]
        if src_t == dst_t:
            uniq, seg, batch, local = _batched_first_unique(torch.cat([srcs, dsts], 1), self.__num_src)
            lists = {src_t: (uniq, seg, batch)}
            inv_src, inv_dst = local[:, :half], local[:, half:]
        else:
            us, sseg, sbatch, inv_src = _batched_first_unique(srcs, self.__num_src)
            ud, dseg, dbatch, inv_dst = _batched_first_unique(dsts, self.__num_dst)
            lists = {src_t: (us, sseg, sbatch), dst_t: (ud, dseg, dbatch)}
        outs, ctx = smp.sample_seed_lists(lists, g, seed + b0)
        views = _group_attribute_views(fs, ctx)
        for j, (node, row, col, edge, nn, ne) in enumerate(outs):
            out = HeteroSamplerOutput(node=node, row=row, col=col, edge=edge,
                                      batch={t: node[t][:nn[t][0]] for t in lists},
                                      num_sampled_nodes={k: torch.tensor(v) for k, v in nn.items()},
                                      num_sampled_edges={k: torch.tensor(v) for k, v in ne.items()}, metadata=None)
            yield self.__emit_hetero(build_hetero_data(fs, out, views, j), ixs[j], inv_src[j], inv_dst[j], bs, n_negs[j])

    def __iter__(self):
        return self.__batches()


class LinkNeighborLoader(LinkLoader):
    """Link-prediction loader with GraphSAGE neighbour sampling around the seed edges' endpoints."""

    def __init__(self, data, num_neighbors, edge_label_index=None, edge_label=None, edge_label_time=None,
                 replace: bool = False, subgraph_type: str = "directional", disjoint: bool = False,
                 temporal_strategy: str = "uniform", neg_sampling=None, neg_sampling_ratio=None,
                 time_attr: Optional[str] = None, weight_attr: Optional[str] = None, transform=None,
                 transform_sampler_output=None, is_sorted: bool = False, filter_per_worker=None,
                 neighbor_sampler=None, directed: bool = True, batch_size: int = 16, compression=None,
                 local_seeds_per_call=None, temporal_comparison: Optional[str] = None, **kwargs):
        if getattr(subgraph_type, "value", subgraph_type) != "directional" or not directed:
            raise ValueError("Only directional subgraphs are currently supported")
        if neighbor_sampler is not None:
            raise ValueError("Passing a neighbor sampler is currently unsupported")
        if not isinstance(data, (list, tuple)) or not isinstance(data[1], GraphStore):
            raise NotImplementedError("Currently can't accept non-cugraph graphs")
        feature_store, graph_store = data
        is_temporal = time_attr is not None
        if is_temporal:
            graph_store._set_time_attr((feature_store, time_attr))
        if weight_attr is not None:
            graph_store._set_weight_attr((feature_store, weight_attr))
        if graph_store._is_single_relation and not isinstance(num_neighbors, dict):
            sampler = NeighborSampler(graph_store._graph, fanout=num_neighbors, biased=(weight_attr is not None),
                                      with_replacement=replace, disjoint=disjoint, temporal=is_temporal,
                                      temporal_comparison=temporal_comparison, local_seeds_per_call=local_seeds_per_call)
        else:
            etypes = [a.edge_type for a in graph_store.get_all_edge_attrs()]
            if not isinstance(num_neighbors, dict):
                num_neighbors = {et: list(num_neighbors) for et in etypes}
            unknown = [k for k in num_neighbors if k not in etypes]
            if unknown:
                raise ValueError(f"fan-out given for unknown edge types: {unknown}")
            sampler = HeteroNeighborSampler(graph_store._hetero_graphs, num_neighbors, biased=(weight_attr is not None),
                                            with_replacement=replace, disjoint=disjoint, temporal=is_temporal,
                                            temporal_comparison=temporal_comparison, local_seeds_per_call=local_seeds_per_call,
                                            num_nodes=graph_store._num_vertices())
        super().__init__((feature_store, graph_store), sampler, edge_label_index=edge_label_index,
                         edge_label=edge_label, edge_label_time=edge_label_time, neg_sampling=neg_sampling,
                         neg_sampling_ratio=neg_sampling_ratio, transform=transform,
                         transform_sampler_output=transform_sampler_output, filter_per_worker=filter_per_worker,
                         batch_size=batch_size, time_attr=time_attr, **kwargs)
