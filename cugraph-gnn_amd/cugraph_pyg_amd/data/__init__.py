from .feature_store import FeatureStore  # noqa: F401
from .graph_store import CSRGraph, GraphStore  # noqa: F401
