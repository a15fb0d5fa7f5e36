"""Grow-only device buffers for tensors whose size changes by a per cent from call group to call group (the rows a group fetches,
the [rows, heads x F] aggregate of a GAT hop): a fresh block from the caching allocator per group is a hipMalloc / hipFree pair in
the middle of the epoch (measured: 11 of the 15 ms a materialised products group took).  A buffer only ever grows and is handed
out again once NOTHING refers to its storage any more — the storage's own use count says so, the caller releases nothing."""
import torch


class GrowOnlyPool:
    """Buffers for the rows a call group fetches (``x = feat[n_id]`` of 191 mini-batches is 4.4 GB on the products workload and
    its size changes by a per cent from group to group: a fresh block from the caching allocator per group is a
    hipMalloc / hipFree pair in the middle of the epoch — 11 of the 15 ms such a group took).  A buffer only ever grows (12 %
    steps) and is handed out again once NOTHING refers to its storage any more — neither the tensor that was returned, nor a
    view of it, nor a tensor autograd saved: the storage's own use count says so, the caller releases nothing."""

    def __init__(self):
        self._bufs = {}

    @staticmethod
    def _idle(buf):
        return torch._C._storage_Use_Count(buf.untyped_storage()._cdata) <= 2     # the pool's tensor + this query's handle

    def take(self, shape, dtype, device):
        nbytes = torch.empty((), dtype=dtype).element_size()
        for d in shape:
            nbytes *= int(d)
        bufs = self._bufs.setdefault(torch.device(device), [])
        idle = [b for b in bufs if self._idle(b)]
        fit = [b for b in idle if b.numel() >= nbytes]
        if fit:
            buf = min(fit, key=lambda b: b.numel())
        else:
            # idle ones are too small for this workload's groups: back to the allocator (by identity: ``list.remove`` would
            # compare tensors element-wise)
            bufs[:] = [b for b in bufs if not any(b is i for i in idle)]
            buf = torch.empty(int(nbytes * 1.12) + (1 << 20), dtype=torch.uint8, device=device)
            bufs.append(buf)
        return buf[:nbytes].view(dtype).view(tuple(int(d) for d in shape))

    def clear(self):
        self._bufs.clear()
