"""``GraphStore`` — PyG GraphStore whose graph lives as a device CSR on MI355X.

Behavioural spec: /root/reference/python/cugraph-pyg/cugraph_pyg/data/graph_store.py:50-631.
Kept from the reference: COO-only ``put_edge_index`` per edge type, lazy construction of the
sampling graph, ``finalize``, per-type vertex offsets by lexicographically sorted type name
(:372-383), edge ids that restart at 0 for every edge type and continue across ranks (:541-575),
and the DIRECTION REVERSAL (:508-539,609-614): the sampling graph's source is PyG's
``edge_index[1]`` (the message target), so expanding a seed follows its in-edges and returns
``edge_index[0]`` endpoints — what a GNN layer aggregates.

Not kept: pylibcugraph ``SGGraph/MGGraph`` (third party, distributed graph).  The sampling graph is
a ``CSRGraph`` replicated on every GPU (288 GB HBM holds all BASELINE graphs): each rank
all-gathers the other ranks' edge slices once at construction, so sampling needs no collective.
"""
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from .._compat import HAS_PYG, EdgeAttr, EdgeLayout

if HAS_PYG:  # pragma: no cover
    from torch_geometric.data import GraphStore as _PygGraphStore
else:
    _PygGraphStore = object


def _world():
    return (dist.get_world_size(), dist.get_rank()) if dist.is_available() and dist.is_initialized() else (1, 0)


@dataclass
class CSRGraph:
    """What the sampler consumes (the role of ``pylibcugraph.SGGraph``): rows = vertices being
    expanded (PyG message targets), ``col`` = their in-neighbours, ``edge_id``/``edge_type`` = the
    original (per-type) edge id / numeric edge type of every CSR slot, ``weight`` optional."""
    row_ptr: torch.Tensor              # int64 [V+1]
    col: torch.Tensor                  # int64 [E]
    edge_id: torch.Tensor              # int64 [E]
    edge_type: Optional[torch.Tensor]  # int32 [E] (None: one edge type)
    weight: Optional[torch.Tensor]     # float32 [E]
    num_vertices: int
    time: Optional[torch.Tensor] = None  # int64 [E] edge timestamps (temporal sampling)


class GraphStore(_PygGraphStore):
    def __init__(self, location: str = "cuda"):
        if location not in ["cpu", "cuda"]:
            raise ValueError("location must be 'cpu' or 'cuda'")
        self.__edge_indices: Dict[Tuple[str, str, str], torch.Tensor] = {}
        self.__sizes = {}
        self.__finalized = False
        self.__weight_attr = None
        self.__time_attr = None
        self.__clear_graph()
        if HAS_PYG:  # pragma: no cover
            super().__init__()

    def __clear_graph(self):
        if self.__finalized:
            raise NotImplementedError("Modifying a finalized GraphStore is not supported.")
        self.__graph = None
        self.__hetero = None
        self.__vertex_offsets = None
        self.__numeric_edge_types = None

    # ---- PyG GraphStore interface ---------------------------------------------------------
    def put_edge_index(self, edge_index, *args, **kwargs) -> bool:
        return self._put_edge_index(edge_index, EdgeAttr(*args, **kwargs))

    def get_edge_index(self, *args, **kwargs):
        return self._get_edge_index(EdgeAttr(*args, **kwargs))

    def remove_edge_index(self, *args, **kwargs) -> bool:
        return self._remove_edge_index(EdgeAttr(*args, **kwargs))

    def __setitem__(self, key, value):
        key = key if isinstance(key, tuple) and isinstance(key[0], tuple) else (key,)
        self.put_edge_index(value, *key)

    def __getitem__(self, key):
        key = key if isinstance(key, tuple) and isinstance(key[0], tuple) else (key,)
        return self.get_edge_index(*key)

    def _put_edge_index(self, edge_index, edge_attr) -> bool:
        if self.__finalized:
            raise NotImplementedError("Adding edges to a finalized GraphStore is not supported.")
        layout = edge_attr.layout if isinstance(edge_attr.layout, EdgeLayout) else EdgeLayout(edge_attr.layout)
        if layout != EdgeLayout.COO:
            raise ValueError("Only COO format supported")
        if isinstance(edge_index, (list, tuple)):
            edge_index = torch.stack([torch.as_tensor(e) for e in edge_index])
        edge_index = torch.as_tensor(edge_index)
        if edge_index.numel() == 0:
            edge_index = torch.zeros((2, 0), dtype=torch.int64)
        if edge_index.shape[0] != 2:
            raise ValueError("Edge index must be of length 2")
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        self.__edge_indices[edge_attr.edge_type] = edge_index.to(device=dev, dtype=torch.int64)
        self.__sizes[edge_attr.edge_type] = edge_attr.size
        self.__clear_graph()
        return True

    def _get_edge_index(self, edge_attr):
        ei = self.__edge_indices[edge_attr.edge_type]   # this rank's slice, COO
        layout = edge_attr.layout if isinstance(edge_attr.layout, EdgeLayout) else EdgeLayout(edge_attr.layout)
        if layout == EdgeLayout.COO:
            return ei
        n_row, n_col = self.__sizes[edge_attr.edge_type] or (int(ei[0].max()) + 1, int(ei[1].max()) + 1)
        major, minor, n = (ei[0], ei[1], n_row) if layout == EdgeLayout.CSR else (ei[1], ei[0], n_col)
        order = torch.sort(major, stable=True).indices
        ptr = torch.zeros(n + 1, dtype=torch.int64, device=ei.device)
        ptr[1:] = torch.cumsum(torch.bincount(major, minlength=n), 0)
        return (ptr, minor[order]) if layout == EdgeLayout.CSR else (minor[order], ptr)

    def _remove_edge_index(self, edge_attr) -> bool:
        if self.__finalized:
            raise NotImplementedError("Removing edges from a finalized GraphStore is not supported.")
        del self.__edge_indices[edge_attr.edge_type]
        self.__clear_graph()
        return True

    def get_all_edge_attrs(self) -> List[EdgeAttr]:
        return [EdgeAttr(edge_type=et, layout="coo", is_sorted=False, size=self.__sizes[et])
                for et in self.__edge_indices.keys()]

    # ---- hooks the loaders rely on (graph_store.py:240-262,333-500) --------------------------
    @property
    def is_multi_gpu(self):
        return _world()[0] > 1

    @property
    def is_homogeneous(self) -> bool:
        return len(self._vertex_offsets) == 1

    @property
    def _is_single_relation(self) -> bool:
        """The criterion the reference uses to pick its homogeneous reader (sampler.py:781-785): exactly ONE edge type
        whose endpoints share a node type.  One node type with several relations is NOT homogeneous for sampling — edge
        ids restart per relation and the output must keep the relations apart (HeteroData)."""
        ets = [a.edge_type for a in self.get_all_edge_attrs()]
        return len(ets) == 1 and ets[0][0] == ets[0][2]

    def finalize(self, weight_attr=None, time_attr=None):
        """Build the device CSR now and drop the COO slices; the store is read-only afterwards."""
        if self.__finalized:
            raise RuntimeError("This GraphStore object has already been finalized.")
        if weight_attr is not None:
            self._set_weight_attr(weight_attr)
        if time_attr is not None:
            self._set_time_attr(time_attr)
        self.__construct_graph()
        self._hetero_graphs  # noqa: B018
        self._vertex_offsets  # noqa: B018  cache before the slices go away
        self._numeric_edge_types  # noqa: B018
        self.__edge_indices = {k: None for k in self.__edge_indices}
        self.__finalized = True
        return self

    def _set_weight_attr(self, attr):
        """``(feature_store, attr_name)``: edge weights for biased sampling (graph_store.py:430-446)."""
        if attr != self.__weight_attr:
            if self.__finalized:   # the COO slices are gone: the CSR cannot be rebuilt with another attribute
                raise NotImplementedError("Modifying a finalized GraphStore is not supported.")
            self.__graph = None
            self.__hetero = None
        self.__weight_attr = attr

    def _set_time_attr(self, attr):
        """``(feature_store, attr_name)``: edge timestamps for temporal sampling (graph_store.py:448-464)."""
        if attr != self.__time_attr:
            if self.__finalized:
                raise NotImplementedError("Modifying a finalized GraphStore is not supported.")
            self.__graph = None
            self.__hetero = None
        self.__time_attr = attr

    def __edge_values(self, attr, key, n_edges, dev, dtype):
        """Per-edge values of one edge type in edge-id order, from a ``(feature_store, name)`` attribute."""
        fs, name = attr
        v = fs[key, name, None]
        v = v[torch.arange(n_edges, device=dev)] if not isinstance(v, torch.Tensor) else v
        return v.to(device=dev, dtype=dtype).view(-1)

    def _num_vertices(self) -> Dict[str, int]:
        if self.__finalized:
            return dict(self.__num_vertices_cache)
        num = {}
        for attr in self.get_all_edge_attrs():
            src_t, _, dst_t = attr.edge_type
            if attr.size is not None:
                num[src_t] = max(num.get(src_t, 0), int(attr.size[0]))
                num[dst_t] = max(num.get(dst_t, 0), int(attr.size[1]))
            else:
                ei = self.__edge_indices[attr.edge_type]
                if ei.numel():
                    num[src_t] = max(num.get(src_t, 0), int(ei[0].max()) + 1)
                    num[dst_t] = max(num.get(dst_t, 0), int(ei[1].max()) + 1)
        if self.is_multi_gpu:
            for k in sorted(num):
                t = torch.tensor(num[k], device="cuda" if torch.cuda.is_available() else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                num[k] = int(t)
        return num

    @property
    def _vertex_offsets(self) -> Dict[str, int]:
        if self.__vertex_offsets is None:
            num = self._num_vertices()
            self.__vertex_offsets, off = {}, 0
            self.__num_vertices_cache = num
            for vtype in sorted(num.keys()):
                self.__vertex_offsets[vtype] = off
                off += num[vtype]
        return dict(self.__vertex_offsets)

    @property
    def _vertex_offset_array(self) -> torch.Tensor:
        offs = self._vertex_offsets
        keys = sorted(offs.keys())
        total = sum(self.__num_vertices_cache.values())
        return torch.tensor([offs[k] for k in keys] + [total], dtype=torch.int64,
                            device="cuda" if torch.cuda.is_available() else "cpu")

    @property
    def _numeric_edge_types(self):
        """(sorted edge types, src vertex-type ids, dst vertex-type ids) in cuGraph orientation."""
        if self.__numeric_edge_types is None:
            sorted_keys = sorted(self.__edge_indices.keys())
            vtypes = sorted(self._vertex_offsets.keys())
            srcs = [vtypes.index(k[2]) for k in sorted_keys]   # cuGraph src = PyG message target
            dsts = [vtypes.index(k[0]) for k in sorted_keys]
            self.__numeric_edge_types = (sorted_keys, torch.tensor(srcs, dtype=torch.int32),
                                         torch.tensor(dsts, dtype=torch.int32))
        return self.__numeric_edge_types

    @property
    def _graph(self) -> CSRGraph:
        return self.__construct_graph()

    def __gathered_edge_index(self, key, dev):
        """All ranks' COO slices of one edge type, concatenated in rank order (= global edge-id order)."""
        ei = self.__edge_indices[key]
        ws, _ = _world()
        if ws == 1:
            return ei
        n_local = torch.tensor([ei.shape[1]], dtype=torch.int64, device=dev)
        sizes = torch.empty(ws, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(sizes, n_local)
        sizes_h = sizes.tolist()
        pad = max(max(sizes_h), 1)
        buf = torch.zeros((2, pad), dtype=torch.int64, device=dev)
        buf[:, : ei.shape[1]] = ei
        allbuf = torch.empty((ws, 2, pad), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allbuf.view(-1), buf.view(-1))
        return torch.cat([allbuf[r, :, : sizes_h[r]] for r in range(ws)], dim=1)

    @property
    def _hetero_graphs(self) -> Dict[Tuple[str, str, str], CSRGraph]:
        """One CSR per edge type in TYPE-LOCAL vertex ids: rows = vertices of the PyG target type
        (``edge_type[2]``), ``col`` = in-neighbours of the source type (``edge_type[0]``).  This is what
        heterogeneous sampling expands (per-edge-type fan-out)."""
        if getattr(self, "_GraphStore__hetero", None) is None:
            dev = "cuda" if torch.cuda.is_available() else "cpu"
            nv = self._num_vertices()
            out = {}
            for key in sorted(self.__edge_indices.keys()):
                ei = self.__gathered_edge_index(key, dev)
                n_rows = nv[key[2]]
                order = torch.sort(ei[1], stable=True).indices
                row_ptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
                row_ptr[1:] = torch.cumsum(torch.bincount(ei[1], minlength=n_rows), 0)
                w = None
                if self.__weight_attr is not None:
                    fs, name = self.__weight_attr
                    wt = fs[key, name, None]
                    wt = wt[torch.arange(ei.shape[1], device=dev)] if not isinstance(wt, torch.Tensor) else wt
                    w = wt.to(device=dev, dtype=torch.float32).view(-1)[order].contiguous()
                tm = None
                if self.__time_attr is not None:
                    tm = self.__edge_values(self.__time_attr, key, ei.shape[1], dev, torch.int64)[order].contiguous()
                out[key] = CSRGraph(row_ptr=row_ptr, col=ei[0][order].contiguous(), edge_id=order.contiguous(),
                                    edge_type=None, weight=w, num_vertices=n_rows, time=tm)
            self.__hetero = out
        return self.__hetero

    def __construct_graph(self) -> CSRGraph:
        if self.__graph is not None:
            return self.__graph
        ws, rank = _world()
        sorted_keys = sorted(self.__edge_indices.keys())
        offs = self._vertex_offsets
        V = sum(self.__num_vertices_cache.values())
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        rows, cols, eids, etps, wgts, tms = [], [], [], [], [], []
        for t, key in enumerate(sorted_keys):
            ei = self.__gathered_edge_index(key, dev)
            src_t, _, dst_t = key
            rows.append(ei[1] + offs[dst_t])     # expanded vertex = PyG message target
            cols.append(ei[0] + offs[src_t])
            eids.append(torch.arange(ei.shape[1], dtype=torch.int64, device=dev))
            etps.append(torch.full((ei.shape[1],), t, dtype=torch.int32, device=dev))
            if self.__weight_attr is not None:
                fs, name = self.__weight_attr
                w = fs[key, name, None]
                w = w[torch.arange(ei.shape[1], device=dev)] if not isinstance(w, torch.Tensor) else w
                wgts.append(w.to(device=dev, dtype=torch.float32).view(-1))
            if self.__time_attr is not None:
                tms.append(self.__edge_values(self.__time_attr, key, ei.shape[1], dev, torch.int64))
        row = torch.cat(rows) if rows else torch.zeros(0, dtype=torch.int64, device=dev)
        col = torch.cat(cols) if cols else torch.zeros(0, dtype=torch.int64, device=dev)
        order = torch.sort(row, stable=True).indices         # CSR order; ties keep edge-id order
        row_ptr = torch.zeros(V + 1, dtype=torch.int64, device=dev)
        row_ptr[1:] = torch.cumsum(torch.bincount(row, minlength=V), 0)
        self.__graph = CSRGraph(
            row_ptr=row_ptr, col=col[order].contiguous(),
            edge_id=torch.cat(eids)[order].contiguous() if eids else col,
            edge_type=torch.cat(etps)[order].contiguous() if len(sorted_keys) > 1 else None,
            weight=torch.cat(wgts)[order].contiguous() if wgts else None, num_vertices=V,
            time=torch.cat(tms)[order].contiguous() if tms else None)
        return self.__graph
