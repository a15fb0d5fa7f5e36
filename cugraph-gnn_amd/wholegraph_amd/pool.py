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

    MAX_BUFFERS = 4            # per device; a loop that needs more at once gets plain allocator blocks for the surplus

    def __init__(self, max_buffers=None):
        self._bufs = {}
        self._baseline = None
        self._max = int(max_buffers or self.MAX_BUFFERS)

    @staticmethod
    def _use_count(buf):
        return torch._C._storage_Use_Count(buf.untyped_storage()._cdata)

    def _idle(self, buf):
        return self._use_count(buf) <= self._baseline

    def _calibrate(self):
        """What the use count of an UNREFERENCED pooled buffer reads in this torch build (the pool's tensor plus whatever the
        query itself holds: 2 in torch 2.10, but storage/PyObject preservation has changed it between releases), and whether a
        view raises it — if it does not, the count cannot tell busy from idle and the pool stands aside (plain allocations)."""
        probe = torch.empty(16, dtype=torch.uint8)
        base = self._use_count(probe)
        view = probe[:8].view(torch.float32)
        held = self._use_count(probe)
        del view
        self._baseline = base if (held > base and self._use_count(probe) == base) else -1

    def take(self, shape, dtype, device):
        nbytes = torch.empty((), dtype=dtype).element_size()
        for d in shape:
            nbytes *= int(d)
        shape = tuple(int(d) for d in shape)
        if self._baseline is None:
            self._calibrate()
        if self._baseline < 0:
            return torch.empty(shape, dtype=dtype, device=device)
        bufs = self._bufs.setdefault(torch.device(device), [])
        idle = [b for b in bufs if self._idle(b)]
        fit = [b for b in idle if b.numel() >= nbytes]
        if fit:
            buf = min(fit, key=lambda b: b.numel())
        else:
            # idle ones are too small for this workload's groups: back to the allocator (by identity: ``list.remove`` would
            # compare tensors element-wise)
            bufs[:] = [b for b in bufs if not any(b is i for i in idle)]
            if len(bufs) >= self._max:
                # every pooled buffer is still referenced by the caller (it keeps several groups alive): do not hoard another
                # multi-GB block for ever — this one goes back to the caching allocator when the caller drops it
                return torch.empty(shape, dtype=dtype, device=device)
            buf = torch.empty(int(nbytes * 1.12) + (1 << 20), dtype=torch.uint8, device=device)
            bufs.append(buf)
        return buf[:nbytes].view(dtype).view(shape)

    def pooled_bytes(self):
        return sum(int(b.numel()) for bufs in self._bufs.values() for b in bufs)

    def clear(self):
        """Drop every pooled buffer (epoch end: the loaders call it when an iterator is exhausted)."""
        self._bufs.clear()
