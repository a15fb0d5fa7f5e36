from .dist_tensor import DistEmbedding, DistTensor  # noqa: F401
