"""cugraph_pyg_amd — the PyG-facing plugin surface (GraphStore / FeatureStore / NeighborLoader /
sampler) of ``cugraph_pyg`` (/root/reference/python/cugraph-pyg/cugraph_pyg/) on the MI355X-native
hot path of ``wholegraph_amd``.  Node loaders (homogeneous and heterogeneous, uniform and biased) are
implemented, plus homogeneous link loaders with binary / triplet negative sampling; temporal and disjoint
sampling are SURVEY.md §8(f) "next"."""
from . import data, loader, sampler, tensor  # noqa: F401
from .data import FeatureStore, GraphStore  # noqa: F401
from .loader import LinkLoader, LinkNeighborLoader, NeighborLoader, NodeLoader  # noqa: F401
