from .link_loader import LinkLoader, LinkNeighborLoader  # noqa: F401
from .neighbor_loader import NeighborLoader  # noqa: F401
from .node_loader import NodeLoader  # noqa: F401
from .call_group import CallGroup, CallGroupIterator, HeteroCallGroup  # noqa: F401
from .per_batch import PerBatchStep, StagedBatch  # noqa: F401
