from .sampler import (BaseSampler, HeteroNeighborSampler, NeighborSampler, SampleIterator,  # noqa: F401
                      hetero_neighbor_sample, neighbor_sample)
