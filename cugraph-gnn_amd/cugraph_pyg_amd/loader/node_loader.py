"""Duck-typed ``torch_geometric.loader.NodeLoader``
(/root/reference/python/cugraph-pyg/cugraph_pyg/loader/node_loader.py:16-178)."""
import warnings
from typing import Callable, Optional

import torch
import torch.distributed as dist

from .._compat import NodeSamplerInput
from ..data.graph_store import GraphStore
from ..sampler import BaseSampler, SampleIterator


def generate_seed() -> int:
    """rank 0 draws, everybody receives, each rank adds its rank (loader/utils.py:9-20)."""
    ws = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if ws == 1:
        return int(torch.randint(0, 2**31 - 1, (1,)))
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    t = torch.randint(0, 2**31 - 1, (1,), device=dev)
    dist.broadcast(t, src=0)
    return int(t) + dist.get_rank()


class NodeLoader:
    def __init__(self, data, node_sampler: BaseSampler, input_nodes=None, input_time=None,
                 transform: Optional[Callable] = None, transform_sampler_output: Optional[Callable] = None,
                 filter_per_worker: Optional[bool] = None, custom_cls=None, input_id=None, batch_size: int = 1,
                 shuffle: bool = False, drop_last: bool = False, random_state: Optional[int] = None, **kwargs):
        if not isinstance(data, (list, tuple)) or not isinstance(data[1], GraphStore):
            raise NotImplementedError("Currently can't accept non-cugraph graphs")
        if not isinstance(node_sampler, BaseSampler):
            raise NotImplementedError("Must provide a cuGraph sampler")
        for name, val in (("filter_per_worker", filter_per_worker), ("custom_cls", custom_cls),
                          ("transform", transform), ("transform_sampler_output", transform_sampler_output)):
            if val:
                warnings.warn(f"{name} is currently ignored")
        graph_store = data[1]
        input_type = None
        self.__has_explicit_input_nodes = not (input_nodes is None or isinstance(input_nodes, str))
        if isinstance(input_nodes, (list, tuple)) and len(input_nodes) == 2 and isinstance(input_nodes[0], str):
            input_type, input_nodes = input_nodes
            self.__has_explicit_input_nodes = input_nodes is not None
        elif isinstance(input_nodes, str):
            input_type, input_nodes = input_nodes, None
        if input_nodes is None:   # all vertices of the (only / named) type
            nv = graph_store._num_vertices()
            vt = input_type if input_type is not None else sorted(nv.keys())[0]
            input_nodes = torch.arange(nv[vt], dtype=torch.int64)
        input_nodes = torch.as_tensor(input_nodes).detach().clone().to(torch.int64)
        if input_nodes.numel() < batch_size and drop_last:
            raise ValueError("The number of input nodes is less than the batch size and drop_last is True. "
                             "This will result in all batches being dropped. Either set drop_last to False or "
                             "increase the number of nodes in input_nodes.")
        if input_type is not None and graph_store._is_single_relation:
            input_nodes = input_nodes + graph_store._vertex_offsets[input_type]   # (0: one vertex type)
        if input_type is None and not graph_store._is_single_relation:
            # several relations go through the heterogeneous sampler (reference sampler.py:781-785), which needs the
            # seed type: with one node type it is that type, otherwise the caller has to name it
            if not graph_store.is_homogeneous:
                raise ValueError("heterogeneous graphs need input_nodes=(node_type, ids)")
            input_type = sorted(graph_store._num_vertices().keys())[0]
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        self.__input_data = NodeSamplerInput(
            input_id=torch.arange(len(input_nodes), dtype=torch.int64, device=dev) if input_id is None else input_id,
            node=input_nodes.to(dev), time=None if input_time is None else torch.as_tensor(input_time).to(dev),
            input_type=input_type)
        self.__data, self.__node_sampler = data, node_sampler
        self.__batch_size, self.__shuffle, self.__drop_last = batch_size, shuffle, drop_last
        self.__random_state = random_state

    def __iter__(self):
        n = self.__input_data.node.numel()
        perm = torch.randperm(n) if self.__shuffle else torch.arange(n)
        if self.__drop_last and n % self.__batch_size > 0:
            perm = perm[: n - n % self.__batch_size]
        perm = perm.to(self.__input_data.node.device)
        input_data = NodeSamplerInput(
            input_id=self.__input_data.input_id[perm], node=self.__input_data.node[perm],
            time=None if self.__input_data.time is None else self.__input_data.time[perm],
            input_type=self.__input_data.input_type)
        seed = self.__random_state if self.__random_state is not None else generate_seed()
        return SampleIterator(self.__data, self.__node_sampler.sample_from_nodes(input_data, random_state=seed))

    def call_groups(self, overlap: bool = True):
        """The same epoch as ``iter(self)`` — same shuffle, same seeds, same samples — handed out in CALL GROUPS:
        ``local_seeds_per_call`` seeds (by default sized from device memory) per ``CallGroup``, each the block-diagonal
        union of its mini-batches with a lazy ``x`` and per-layer trimmed graphs (``loader/call_group.py``).  For loops
        that want the device's speed rather than one ``Data`` per 1024 seeds.  Homogeneous graphs."""
        from .call_group import CallGroupIterator
        n = self.__input_data.node.numel()
        perm = torch.randperm(n) if self.__shuffle else torch.arange(n)
        if self.__drop_last and n % self.__batch_size > 0:
            perm = perm[: n - n % self.__batch_size]
        perm = perm.to(self.__input_data.node.device)
        input_data = NodeSamplerInput(
            input_id=self.__input_data.input_id[perm], node=self.__input_data.node[perm],
            time=None if self.__input_data.time is None else self.__input_data.time[perm],
            input_type=self.__input_data.input_type)
        seed = self.__random_state if self.__random_state is not None else generate_seed()
        return CallGroupIterator(self.__data, self.__node_sampler.core, input_data, seed, self.__batch_size, overlap=overlap)

    def __len__(self):
        if not self.__has_explicit_input_nodes:
            raise ValueError("len(loader) is only supported when the loader was constructed with an explicit "
                             "number of seeds via input_nodes for now.")
        n = self.__input_data.node.numel()
        return n // self.__batch_size if self.__drop_last else (n + self.__batch_size - 1) // self.__batch_size
