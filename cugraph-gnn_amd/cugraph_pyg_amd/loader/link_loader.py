"""Duck-typed ``torch_geometric.loader.LinkLoader`` / ``LinkNeighborLoader``: mini-batches seeded by
EDGES (link prediction), optional negative sampling.

Reference: loader/link_loader.py:17-234, loader/link_neighbor_loader.py, and the edge-seed path of the
sampler (/root/reference/python/cugraph-pyg/cugraph_pyg/sampler/sampler.py:799-896,
sampler/distributed_sampler.py:428-638, negative sampling sampler/sampler_utils.py:93-336).  Per batch:
the endpoints of the seed edges (plus negatives) are deduplicated in first-appearance order — the
renumbering kernel with an empty target list does exactly that and returns the inverse map, which IS
``edge_label_index`` — then the unique endpoints are expanded like node seeds.

Implemented: homogeneous graphs, ``neg_sampling`` = None | "binary" | "triplet" (uniform negative
destinations).  Not implemented: temporal constraints, heterogeneous edge seeds.
"""
import warnings
from typing import Optional, Tuple, Union

import torch

from wholegraph_amd import graph_ops

from ..data.graph_store import GraphStore
from ..sampler.sampler import NeighborSampler, SampleIterator, filter_store, neighbor_sample
from .._compat import Data
from .node_loader import generate_seed


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
                 random_state: Optional[int] = None, **kwargs):
        if not isinstance(data, (list, tuple)) or not isinstance(data[1], GraphStore):
            raise NotImplementedError("Currently can't accept non-cugraph graphs")
        if not isinstance(link_sampler, NeighborSampler):
            raise NotImplementedError("Must provide a cuGraph sampler")
        if edge_label_time is not None:
            raise NotImplementedError("temporal link sampling is not implemented")
        if neg_sampling_ratio is not None:
            warnings.warn("neg_sampling_ratio is deprecated; use neg_sampling=('binary', ratio)")
            neg_sampling = ("binary", float(neg_sampling_ratio))
        self.__mode, self.__amount = _parse_neg_sampling(neg_sampling)
        if self.__mode not in (None, "binary", "triplet"):
            raise ValueError(f"unknown negative sampling mode {self.__mode!r}")
        graph_store = data[1]
        if not graph_store.is_homogeneous:
            raise NotImplementedError("heterogeneous edge seeds are not implemented")
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        if isinstance(edge_label_index, (tuple, list)) and len(edge_label_index) == 2 and not torch.is_tensor(
                edge_label_index[0]):
            edge_label_index = edge_label_index[1]            # (edge_type, tensor)
        if edge_label_index is None:                          # all edges of the graph
            edge_label_index = graph_store.get_edge_index(graph_store.get_all_edge_attrs()[0].edge_type, "coo")
        self.__eli = torch.as_tensor(edge_label_index).to(dev).long()
        if self.__eli.dim() != 2 or self.__eli.shape[0] != 2:
            raise ValueError("edge_label_index must be a 2 x N tensor")
        n = self.__eli.shape[1]
        self.__label = None if edge_label is None else torch.as_tensor(edge_label).to(dev)
        self.__input_id = torch.arange(n, device=dev) if input_id is None else torch.as_tensor(input_id).to(dev)
        if n < batch_size and drop_last:
            raise ValueError("The number of input edges is less than the batch size and drop_last is True.")
        self.__data, self.__sampler = data, link_sampler
        self.__batch_size, self.__shuffle, self.__drop_last = batch_size, shuffle, drop_last
        self.__random_state = random_state
        self.__num_nodes = graph_store._graph.num_vertices

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
        graph = self.__sampler.graph
        for b, start in enumerate(range(0, perm.numel(), self.__batch_size)):
            ix = perm[start:start + self.__batch_size]
            src, dst = self.__eli[0, ix], self.__eli[1, ix]
            n_pos = ix.numel()
            gen = torch.Generator(device=dev).manual_seed((seed + b) & 0x7FFFFFFFFFFFFFFF)
            n_neg = int(round(n_pos * self.__amount)) if self.__mode else 0
            neg_dst = torch.randint(0, self.__num_nodes, (n_neg,), generator=gen, device=dev) if n_neg else None
            if self.__mode == "binary":
                neg_src = torch.randint(0, self.__num_nodes, (n_neg,), generator=gen, device=dev)
                ends = torch.cat([src, neg_src, dst, neg_dst])
            elif self.__mode == "triplet":
                ends = torch.cat([src, dst, neg_dst])
            else:
                ends = torch.cat([src, dst])
            # first-appearance dedup + inverse map = renumbering with no targets
            uniq, inverse = graph_ops.append_unique(ends[:0].contiguous(), ends.contiguous(),
                                                    need_neighbor_raw_to_unique=True)
            node, row, col, edge, nn, ne = neighbor_sample(graph, uniq, self.__sampler.fanout, seed + b,
                                                           self.__sampler.biased)
            data = filter_store(fs, gs, node, row, col, edge)
            data.n_id, data.e_id = node, edge
            data.num_sampled_nodes, data.num_sampled_edges = torch.tensor(nn), torch.tensor(ne)
            data.input_id = self.__input_id[ix]
            data.batch_size = n_pos
            inverse = inverse.long()
            if self.__mode == "triplet":
                data.src_index = inverse[:n_pos]
                data.dst_pos_index = inverse[n_pos:2 * n_pos]
                data.dst_neg_index = inverse[2 * n_pos:].view(n_pos, -1) if n_neg % max(n_pos, 1) == 0 and n_neg > n_pos \
                    else inverse[2 * n_pos:]
            else:
                half = n_pos + (n_neg if self.__mode == "binary" else 0)
                data.edge_label_index = torch.stack([inverse[:half], inverse[half:]])
                if self.__mode == "binary":
                    pos = torch.ones(n_pos, device=dev) if self.__label is None else (self.__label[ix] + 1)
                    data.edge_label = torch.cat([pos, torch.zeros(n_neg, device=dev, dtype=pos.dtype)])
                elif self.__label is not None:
                    data.edge_label = self.__label[ix]
            yield data

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
                 local_seeds_per_call=None, **kwargs):
        if getattr(subgraph_type, "value", subgraph_type) != "directional" or not directed:
            raise ValueError("Only directional subgraphs are currently supported")
        if neighbor_sampler is not None:
            raise ValueError("Passing a neighbor sampler is currently unsupported")
        if time_attr is not None:
            raise NotImplementedError("temporal sampling is not implemented")
        if not isinstance(data, (list, tuple)) or not isinstance(data[1], GraphStore):
            raise NotImplementedError("Currently can't accept non-cugraph graphs")
        feature_store, graph_store = data
        if weight_attr is not None:
            graph_store._set_weight_attr((feature_store, weight_attr))
        sampler = NeighborSampler(graph_store._graph, fanout=num_neighbors, biased=(weight_attr is not None),
                                  with_replacement=replace, disjoint=disjoint)
        super().__init__((feature_store, graph_store), sampler, edge_label_index=edge_label_index,
                         edge_label=edge_label, edge_label_time=edge_label_time, neg_sampling=neg_sampling,
                         neg_sampling_ratio=neg_sampling_ratio, transform=transform,
                         transform_sampler_output=transform_sampler_output, filter_per_worker=filter_per_worker,
                         batch_size=batch_size, **kwargs)
