"""torch_geometric stand-ins.

The reference duck-types PyG (``import_optional('torch_geometric')``,
/root/reference/python/cugraph-pyg/cugraph_pyg/utils/imports.py:8-89) but needs the real package at
run time.  PyG is not installable in this image, so when it is missing the few PyG value types the
loader path touches are provided here with the same field names; when ``torch_geometric`` IS
importable the real classes are used, so the stores/loaders plug into a PyG training loop unchanged.
"""
from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Optional, Tuple

try:  # pragma: no cover - not available in the build image
    import torch_geometric  # noqa: F401
    from torch_geometric.data import Data, HeteroData
    from torch_geometric.data.feature_store import TensorAttr
    from torch_geometric.data.graph_store import EdgeAttr, EdgeLayout
    from torch_geometric.sampler import EdgeSamplerInput, HeteroSamplerOutput, NodeSamplerInput, SamplerOutput
    HAS_PYG = True
except ImportError:
    HAS_PYG = False

    class EdgeLayout(Enum):
        COO = "coo"
        CSC = "csc"
        CSR = "csr"

    @dataclass
    class EdgeAttr:
        """torch_geometric.data.graph_store.EdgeAttr"""
        edge_type: Tuple[str, str, str]
        layout: Any = EdgeLayout.COO
        is_sorted: bool = False
        size: Optional[Tuple[int, int]] = None

        def __post_init__(self):
            self.layout = EdgeLayout(self.layout) if not isinstance(self.layout, EdgeLayout) else self.layout

    _UNSET = object()

    @dataclass
    class TensorAttr:
        """torch_geometric.data.feature_store.TensorAttr"""
        group_name: Any = _UNSET
        attr_name: Any = _UNSET
        index: Any = _UNSET

        def is_set(self, key):
            return getattr(self, key) is not _UNSET

        def is_fully_specified(self):
            return all(self.is_set(k) for k in ("group_name", "attr_name", "index"))

    class Data:
        """Attribute bag with the subset of torch_geometric.data.Data the loader fills."""

        def __init__(self, **kwargs):
            self.__dict__["_store"] = dict(kwargs)

        def __getattr__(self, k):
            try:
                return self.__dict__["_store"][k]
            except KeyError:
                raise AttributeError(k) from None

        def __setattr__(self, k, v):
            self._store[k] = v

        def __getitem__(self, k):
            return self._store[k]

        def __setitem__(self, k, v):
            self._store[k] = v

        def __contains__(self, k):
            return k in self._store

        def keys(self):
            return list(self._store.keys())

        def to(self, device):
            import torch
            for k, v in self._store.items():
                if isinstance(v, torch.Tensor):
                    self._store[k] = v.to(device)
            return self

        def __repr__(self):
            import torch
            parts = [f"{k}={list(v.shape) if isinstance(v, torch.Tensor) else v}" for k, v in self._store.items()]
            return "Data(" + ", ".join(parts) + ")"

    class HeteroData:
        """``data[node_type]`` / ``data[src, rel, dst]`` -> attribute bags (subset of
        torch_geometric.data.HeteroData the loader fills)."""

        def __init__(self):
            self._stores = {}

        def __getitem__(self, key):
            key = tuple(key) if isinstance(key, (tuple, list)) else key
            if key not in self._stores:
                self._stores[key] = Data()
            return self._stores[key]

        @property
        def node_types(self):
            return [k for k in self._stores if not isinstance(k, tuple)]

        @property
        def edge_types(self):
            return [k for k in self._stores if isinstance(k, tuple)]

        def set_value_dict(self, name, value_dict):
            for k, v in (value_dict or {}).items():
                self[k][name] = v

        def __getattr__(self, name):
            # data.<attr>_dict -> {type: value} over the stores that hold <attr> (PyG's collect())
            if name.endswith("_dict") and not name.startswith("_"):
                attr = name[:-5]
                return {k: st[attr] for k, st in self._stores.items() if attr in st}
            raise AttributeError(name)

        def __repr__(self):
            return "HeteroData(" + ", ".join(f"{k}={v}" for k, v in self._stores.items()) + ")"

    @dataclass
    class NodeSamplerInput:
        """torch_geometric.sampler.NodeSamplerInput"""
        input_id: Any
        node: Any
        time: Any = None
        input_type: Optional[str] = None

    @dataclass
    class EdgeSamplerInput:
        """torch_geometric.sampler.EdgeSamplerInput"""
        input_id: Any
        row: Any
        col: Any
        label: Any = None
        time: Any = None
        input_type: Any = None

    @dataclass
    class SamplerOutput:
        """torch_geometric.sampler.SamplerOutput"""
        node: Any
        row: Any
        col: Any
        edge: Any
        batch: Any = None
        num_sampled_nodes: Any = None
        num_sampled_edges: Any = None
        orig_row: Any = None
        orig_col: Any = None
        metadata: Any = field(default=None)

    @dataclass
    class HeteroSamplerOutput:
        """torch_geometric.sampler.HeteroSamplerOutput"""
        node: Any
        row: Any
        col: Any
        edge: Any
        batch: Any = None
        num_sampled_nodes: Any = None
        num_sampled_edges: Any = None
        orig_row: Any = None
        orig_col: Any = None
        metadata: Any = field(default=None)
