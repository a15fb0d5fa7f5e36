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
from typing import Iterator, List, Optional, Sequence

import torch

from wholegraph_amd import graph_ops, wholegraph_ops

from .._compat import Data, NodeSamplerInput, SamplerOutput
from ..data.graph_store import CSRGraph

_GOLDEN = 0x9E3779B97F4A7C15
_MASK = 0xFFFFFFFFFFFFFFFF


def hop_seed(random_state: int, hop: int) -> int:
    """Seed of hop ``hop`` of a call whose ``random_state`` is given (batch b of a loader epoch uses
    ``random_state + b``, as ``random_state + rank`` in distributed_sampler.py:896)."""
    return (int(random_state) + hop * _GOLDEN) & _MASK


def neighbor_sample(graph: CSRGraph, seeds: torch.Tensor, fanout: Sequence[int], random_state: int,
                    biased: bool = False):
    """One mini-batch.  Returns (node, row, col, edge, num_sampled_nodes, num_sampled_edges)."""
    seeds = seeds.to(device=graph.row_ptr.device, dtype=graph.col.dtype)
    nodes = seeds
    frontier, f_start = seeds, 0
    rows, cols, edges = [], [], []
    num_nodes, num_edges = [int(seeds.shape[0])], []
    for k, fan in enumerate(fanout):
        if frontier.shape[0] == 0:
            num_edges.append(0)
            num_nodes.append(0)
            continue
        if biased:
            off, nbr, lid, gid = wholegraph_ops.weighted_sample_without_replacement(
                graph.row_ptr, graph.col, graph.weight, frontier, int(fan), hop_seed(random_state, k), True, True)
            # libcugraph's biased sampling never returns a zero-weight edge, even when the row is
            # shorter than the fan-out (tests/loader/test_neighbor_loader.py:99-133); the WholeGraph
            # kernel copies short rows whole, so drop those edges here.
            keep = graph.weight[gid] > 0
            if not bool(keep.all()):
                nbr, lid, gid = nbr[keep], lid[keep], gid[keep]
        else:
            off, nbr, lid, gid = wholegraph_ops.unweighted_sample_without_replacement(
                graph.row_ptr, graph.col, frontier, int(fan), hop_seed(random_state, k), True, True)
        new_nodes, mapping = graph_ops.append_unique(nodes, nbr, need_neighbor_raw_to_unique=True)
        rows.append(mapping.long())
        cols.append(lid.long() + f_start)
        edges.append(graph.edge_id[gid])
        num_edges.append(int(nbr.shape[0]))
        num_nodes.append(int(new_nodes.shape[0] - nodes.shape[0]))
        f_start = int(nodes.shape[0])
        frontier = new_nodes[f_start:]
        nodes = new_nodes
    cat = (lambda xs: torch.cat(xs) if xs else torch.zeros(0, dtype=torch.int64, device=seeds.device))
    return nodes, cat(rows), cat(cols), cat(edges), num_nodes, num_edges


class NeighborSampler:
    """The role of ``DistributedNeighborSampler`` for the homogeneous case: owns the CSR, the fan-out
    and the flags; ``sample_batches`` is the hot loop."""

    def __init__(self, graph: CSRGraph, fanout: Sequence[int], biased: bool = False,
                 with_replacement: bool = False, disjoint: bool = False, heterogeneous: bool = False,
                 temporal: bool = False, **_ignored):
        if with_replacement:
            raise NotImplementedError("sampling with replacement is not implemented (kernels sample without)")
        if disjoint or heterogeneous or temporal:
            raise NotImplementedError("disjoint / heterogeneous / temporal sampling: SURVEY.md §8(f) 'next'")
        if biased and graph.weight is None:
            raise ValueError("biased sampling needs a weight attribute (weight_attr=...)")
        self.graph, self.fanout, self.biased = graph, [int(f) for f in fanout], biased

    def sample_batches(self, seeds: torch.Tensor, batch_size: int, random_state: int) -> Iterator:
        n = seeds.shape[0]
        for b, start in enumerate(range(0, n, batch_size)):
            yield b, neighbor_sample(self.graph, seeds[start:start + batch_size], self.fanout, random_state + b,
                                     self.biased)


class BaseSampler:
    """``sample_from_nodes(NodeSamplerInput)`` → iterator of ``SamplerOutput`` (sampler.py:756-797)."""

    def __init__(self, sampler: NeighborSampler, data, batch_size: int = 16):
        self.__sampler = sampler
        self.__feature_store, self.__graph_store = data
        self.__batch_size = batch_size

    def sample_from_nodes(self, index: NodeSamplerInput, random_state: int = 62, **kwargs) -> Iterator[SamplerOutput]:
        nodes = index.node
        input_id = index.input_id
        bs = self.__batch_size
        for b, (node, row, col, edge, nn, ne) in self.__sampler.sample_batches(nodes, bs, random_state):
            n_seeds = nn[0]
            ids = input_id[b * bs: b * bs + n_seeds]
            yield SamplerOutput(
                node=node, row=row, col=col, edge=edge, batch=node[:n_seeds],
                num_sampled_nodes=torch.tensor(nn), num_sampled_edges=torch.tensor(ne),
                metadata=(ids, None))

    def sample_from_edges(self, index, neg_sampling=None, **kwargs):
        raise NotImplementedError("link loaders / negative sampling: SURVEY.md §8(f) rank 1")


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


class SampleIterator:
    """Joins features to sampler outputs and emits PyG ``Data`` (sampler.py:51-165)."""

    def __init__(self, data, output_iter: Iterator[SamplerOutput]):
        self.__feature_store, self.__graph_store = data
        self.__output_iter = output_iter

    def __next__(self):
        s = next(self.__output_iter)
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
        data.seed_time = s.metadata[1]
        return data

    def __iter__(self):
        return self
