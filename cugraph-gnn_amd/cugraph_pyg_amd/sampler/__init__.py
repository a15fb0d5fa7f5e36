from .sampler import BaseSampler, NeighborSampler, SampleIterator, neighbor_sample  # noqa: F401
