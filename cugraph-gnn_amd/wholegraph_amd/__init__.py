"""wholegraph_amd — MI355X-native mirror of ``pylibwholegraph.torch`` for the mini-batch hot path.

Public surface (same names as /root/reference/python/pylibwholegraph/pylibwholegraph/torch/__init__.py
for the files SURVEY.md §8 puts on the path): ``GraphStructure``, ``WholeMemoryTensor``,
``create_wholememory_tensor``, ``wholegraph_ops``, ``graph_ops``; plus ``nn`` (SAGEConv / GATConv over the
sampler CSR) and ``fused`` (no-host-sync walk), which have no reference counterpart.

Everything computes through ``lib/libwholegraph_amd.so`` (HIP, gfx950); a missing library raises
``WholeGraphLibraryError`` — there is no CPU fallback.
"""
from . import _lib, comm, dist, embedding, env, fused, graph_ops, nn, wholegraph_ops  # noqa: F401
from ._lib import WholeGraphLibraryError, WholeMemoryError  # noqa: F401
from .graph_structure import GraphStructure  # noqa: F401
from .comm import (WholeMemoryCommunicator, create_group_communicator,  # noqa: F401
                   destroy_communicator, get_global_communicator, get_local_device_communicator,
                   get_local_node_communicator)
from .tensor import (DistributedWholeMemoryTensor, WholeMemoryTensor,  # noqa: F401
                     create_wholememory_tensor, destroy_wholememory_tensor, equal_entry_partition)
from .embedding import (WholeMemoryCachePolicy, WholeMemoryEmbedding, WholeMemoryEmbeddingModule,  # noqa: F401
                        WholeMemoryOptimizer, create_builtin_cache_policy, create_wholememory_cache_policy,
                        destroy_wholememory_cache_policy, create_embedding, create_embedding_from_filelist,
                        create_wholememory_optimizer, destroy_embedding, destroy_wholememory_optimizer)
