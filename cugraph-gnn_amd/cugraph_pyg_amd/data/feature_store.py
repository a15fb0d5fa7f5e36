"""``FeatureStore`` — PyG FeatureStore over HBM-resident (optionally range-partitioned) tables; ``location="cpu"`` keeps
the rows in pinned host memory instead (the reference's default placement, feature_store.py:42-58), read in place by the same
kernels.

Behavioural spec: /root/reference/python/cugraph-pyg/cugraph_pyg/data/feature_store.py:24-239 —
``store[group, attr, None] = tensor`` stores this rank's slice (ranks' slices are concatenated in
rank order, :169-173), ``store[group, attr, None]`` returns the distributed tensor whose ``[idx]``
gathers global rows, ``store[group, attr, idx]`` gathers directly.  Backed by
``cugraph_pyg_amd.tensor.DistTensor/DistEmbedding`` → ``wholegraph_amd.WholeMemoryTensor``
(HIP gather/scatter kernels; RCCL all-to-all across GPUs)."""
import warnings
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from .._compat import HAS_PYG, TensorAttr
from ..tensor import DistEmbedding, DistTensor

if HAS_PYG:  # pragma: no cover
    from torch_geometric.data import FeatureStore as _PygFeatureStore
else:
    _PygFeatureStore = object


def _world():
    return (dist.get_world_size(), dist.get_rank()) if dist.is_available() and dist.is_initialized() else (1, 0)


class FeatureStore(_PygFeatureStore):
    def __init__(self, memory_type=None, location="cuda"):
        if HAS_PYG:  # pragma: no cover
            super().__init__()
        self.__features = {}
        if location not in ("cpu", "cuda"):
            raise ValueError("location must be 'cpu' or 'cuda'")
        self.__location = location
        if memory_type is not None:
            warnings.warn("The memory_type argument is deprecated. Memory type is now automatically inferred.")

    def __make_tensor(self, tensor: torch.Tensor, ix=None):
        ws, rank = _world()
        if tensor.dim() not in (1, 2):
            raise ValueError("Tensor must be 1D or 2D.")
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        cls = DistTensor if tensor.dim() == 1 else DistEmbedding
        if ws == 1 and ix is None:
            return cls(tensor, device=self.__location)
        n_local = torch.tensor([tensor.shape[0], tensor.dim(), tensor.shape[1] if tensor.dim() == 2 else 1],
                               dtype=torch.int64, device=dev)
        meta = torch.empty((ws, 3), dtype=torch.int64, device=dev)
        if ws > 1:
            dist.all_gather_into_tensor(meta.view(-1), n_local)
        else:
            meta[0] = n_local
        meta_h = meta.tolist()
        if any(m[1] != meta_h[0][1] for m in meta_h):
            raise ValueError("Tensor dimension must be the same across ranks")
        if any(m[2] != meta_h[0][2] for m in meta_h if m[0] > 0):
            raise ValueError("Trailing dimensions must be the same across ranks")
        sizes = [m[0] for m in meta_h]
        total = int(ix.max()) + 1 if ix is not None and ws == 1 else sum(sizes)
        shape = (total,) if tensor.dim() == 1 else (total, tensor.shape[1])
        tx = cls(None, shape=shape, dtype=tensor.dtype, device=self.__location)
        if ix is None:
            off = sum(sizes[:rank])
            ix = torch.arange(off, off + tensor.shape[0], dtype=torch.int64, device=dev)
        if tensor.shape[0] != ix.shape[0]:
            raise ValueError("Shape mismatch")
        tx[ix] = tensor
        return tx

    # ---- PyG FeatureStore interface ----------------------------------------------------------
    @staticmethod
    def _attr(key) -> TensorAttr:
        if isinstance(key, TensorAttr):
            return key
        if not isinstance(key, tuple):
            key = (key,)
        return TensorAttr(*key)

    def __setitem__(self, key, value):
        self.put_tensor(value, self._attr(key))

    def __getitem__(self, key):
        attr = self._attr(key)
        out = self._get_tensor(attr)
        if out is None:
            raise KeyError(f"no tensor for {attr}")
        return out

    def __delitem__(self, key):
        # torch_geometric.data.FeatureStore.__delitem__: `del store[group, attr, index]`
        self.remove_tensor(self._attr(key))

    def put_tensor(self, tensor, *args, **kwargs) -> bool:
        attr = args[0] if args and isinstance(args[0], TensorAttr) else TensorAttr(*args, **kwargs)
        return self._put_tensor(torch.as_tensor(tensor), attr)

    def get_tensor(self, *args, **kwargs):
        attr = args[0] if args and isinstance(args[0], TensorAttr) else TensorAttr(*args, **kwargs)
        return self._get_tensor(attr)

    def multi_get_tensor(self, attrs: List[TensorAttr]):
        return [self._get_tensor(a) for a in attrs]

    def remove_tensor(self, *args, **kwargs) -> bool:
        attr = args[0] if args and isinstance(args[0], TensorAttr) else TensorAttr(*args, **kwargs)
        return self._remove_tensor(attr)

    def _put_tensor(self, tensor, attr) -> bool:
        key = (attr.group_name, attr.attr_name)
        if attr.is_set("index") and attr.index is not None:
            if key not in self.__features:
                self.__features[key] = self.__make_tensor(tensor, ix=torch.as_tensor(attr.index))
            else:
                self.__features[key][attr.index] = tensor
        else:
            self.__features[key] = self.__make_tensor(tensor)
        return True

    def _get_tensor(self, attr) -> Optional[torch.Tensor]:
        key = (attr.group_name, attr.attr_name)
        if key not in self.__features:
            return None
        emb = self.__features[key]
        if attr.is_set("index") and attr.index is not None:
            return emb[attr.index]
        return emb

    def _remove_tensor(self, attr) -> bool:
        return self.__features.pop((attr.group_name, attr.attr_name), None) is not None

    def _get_tensor_size(self, attr) -> Tuple:
        return self.__features[attr.group_name, attr.attr_name].shape

    def get_tensor_size(self, *args, **kwargs):
        attr = args[0] if args and isinstance(args[0], TensorAttr) else TensorAttr(*args, **kwargs)
        return self._get_tensor_size(attr)

    def get_all_tensor_attrs(self) -> List[TensorAttr]:
        return [TensorAttr(group_name=g, attr_name=a) for g, a in self.__features.keys()]
