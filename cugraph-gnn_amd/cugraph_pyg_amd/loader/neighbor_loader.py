"""Duck-typed ``torch_geometric.loader.NeighborLoader`` — the GraphSAGE neighbour-sampling loader
(/root/reference/python/cugraph-pyg/cugraph_pyg/loader/neighbor_loader.py:20-236)."""
import warnings
from typing import Callable, Dict, List, Optional, Union

from ..data.graph_store import GraphStore
from ..sampler import BaseSampler, HeteroNeighborSampler, NeighborSampler
from .node_loader import NodeLoader


class NeighborLoader(NodeLoader):
    def __init__(self, data, num_neighbors: Union[List[int], Dict], input_nodes=None, input_time=None,
                 replace: bool = False, subgraph_type: str = "directional", disjoint: bool = False,
                 temporal_strategy: str = "uniform", time_attr: Optional[str] = None,
                 weight_attr: Optional[str] = None, transform: Optional[Callable] = None,
                 transform_sampler_output: Optional[Callable] = None, is_sorted: bool = False,
                 filter_per_worker: Optional[bool] = None, neighbor_sampler=None, directed: bool = True,
                 batch_size: int = 16, compression: Optional[str] = None,
                 local_seeds_per_call: Optional[int] = None, temporal_comparison: Optional[str] = None, **kwargs):
        subgraph_type = getattr(subgraph_type, "value", subgraph_type)
        if not directed:
            subgraph_type = "induced"
            warnings.warn("The 'directed' argument is deprecated. Use subgraph_type='induced' instead.")
        if subgraph_type != "directional":
            raise ValueError("Only directional subgraphs are currently supported")
        if temporal_strategy != "uniform":
            warnings.warn("Only the uniform temporal strategy is currently supported")
        if neighbor_sampler is not None:
            raise ValueError("Passing a neighbor sampler is currently unsupported")
        if is_sorted:
            warnings.warn("The 'is_sorted' argument is ignored by cuGraph.")
        if not isinstance(data, (list, tuple)) or not isinstance(data[1], GraphStore):
            raise NotImplementedError("Currently can't accept non-cugraph graphs")
        feature_store, graph_store = data
        if compression is not None and compression not in ["CSR", "COO"]:
            raise ValueError("Invalid value for compression (expected 'CSR' or 'COO')")
        is_temporal = time_attr is not None
        if is_temporal:
            # edge timestamps come from the feature store; seeds without input_time take their node's timestamp
            # (neighbor_loader.py:173-187)
            graph_store._set_time_attr((feature_store, time_attr))
            if input_time is None:
                # resolve the seeds the way NodeLoader does: ('type', ids | None), a bare type name, ids, or None
                nv = graph_store._num_vertices()
                if isinstance(input_nodes, (tuple, list)) and len(input_nodes) == 2 and isinstance(input_nodes[0], str):
                    in_type, in_nodes = input_nodes
                elif isinstance(input_nodes, str):
                    in_type, in_nodes = input_nodes, None
                else:
                    in_type, in_nodes = sorted(nv.keys())[0], input_nodes
                if in_nodes is None:
                    import torch
                    in_nodes = torch.arange(nv[in_type], dtype=torch.int64)
                input_time = feature_store[in_type, time_attr, None][in_nodes]
        if weight_attr is not None:
            graph_store._set_weight_attr((feature_store, weight_attr))
        if graph_store._is_single_relation and not isinstance(num_neighbors, dict):
            core = NeighborSampler(graph_store._graph, fanout=num_neighbors, biased=(weight_attr is not None),
                                   with_replacement=replace, disjoint=disjoint, heterogeneous=False,
                                   temporal=is_temporal, temporal_comparison=temporal_comparison,
                                   local_seeds_per_call=local_seeds_per_call)
        else:
            if compression is not None and compression != "COO":
                raise ValueError("Only COO format is supported for heterogeneous graphs!")
            etypes = [a.edge_type for a in graph_store.get_all_edge_attrs()]
            if not isinstance(num_neighbors, dict):      # a plain list applies to every edge type (PyG)
                num_neighbors = {et: list(num_neighbors) for et in etypes}
            unknown = [k for k in num_neighbors if k not in etypes]
            if unknown:
                raise ValueError(f"fan-out given for unknown edge types: {unknown}")
            core = HeteroNeighborSampler(graph_store._hetero_graphs, num_neighbors, biased=(weight_attr is not None),
                                         with_replacement=replace, disjoint=disjoint, temporal=is_temporal,
                                         temporal_comparison=temporal_comparison,
                                         local_seeds_per_call=local_seeds_per_call,
                                         num_nodes=graph_store._num_vertices())
        sampler = BaseSampler(core, (feature_store, graph_store), batch_size=batch_size)
        super().__init__((feature_store, graph_store), sampler, input_nodes=input_nodes, input_time=input_time,
                         transform=transform, transform_sampler_output=transform_sampler_output,
                         filter_per_worker=filter_per_worker, batch_size=batch_size, **kwargs)
