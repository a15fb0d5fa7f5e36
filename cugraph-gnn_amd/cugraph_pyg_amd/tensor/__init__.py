from .dist_matrix import DistMatrix  # noqa: F401
from .dist_tensor import DistEmbedding, DistTensor  # noqa: F401
