"""torch-backed allocator callbacks + tensor wrapping for the C ABI.

Same contract as the reference's ``wholegraph_env.py``
(/root/reference/python/pylibwholegraph/pylibwholegraph/torch/wholegraph_env.py:20-211) and its
C++ twin (torch_cpp_ext/torch_env_func_ptrs.cpp:13-56): op outputs whose size is only known
inside the op are allocated by the op through ``wholememory_env_func_t``; the callback creates a
``torch.Tensor`` and parks it in a ``TorchMemoryContext`` the Python caller reads back.
"""
import ctypes
import itertools
import threading

import torch

from . import _lib as L

_TORCH_TO_WM = {
    torch.float32: L.DT_FLOAT, torch.float16: L.DT_HALF, torch.float64: L.DT_DOUBLE,
    torch.bfloat16: L.DT_BF16, torch.int32: L.DT_INT, torch.int64: L.DT_INT64,
    torch.int16: L.DT_INT16, torch.int8: L.DT_INT8,
}
_WM_TO_TORCH = {v: k for k, v in _TORCH_TO_WM.items()}


def torch_dtype_to_wm(dt):
    try:
        return _TORCH_TO_WM[dt]
    except KeyError:
        raise TypeError(f"dtype {dt} has no wholememory_dtype_t") from None


def wm_dtype_to_torch(code):
    return _WM_TO_TORCH[code]


_HAVE_GPU = None


def get_stream():
    """Current HIP stream of torch as the ``void* stream`` argument (wholegraph_env.py:20-27).  (Through torch's raw-stream
    entry point: ``torch.cuda.current_stream()`` builds a Stream object and resolves the device index in Python, 13 us per
    call — every op of a per-mini-batch loop pays it.)"""
    global _HAVE_GPU
    if _HAVE_GPU is None:
        _HAVE_GPU = torch.cuda.is_available()
    if _HAVE_GPU:
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    return ctypes.c_void_p(0)


class TorchMemoryContext:
    """Holds the tensor an op allocated for one output (wholegraph_env.py:35-61)."""

    _ids = itertools.count(1)
    _live = {}
    _lock = threading.Lock()

    def __init__(self):
        self.tensor = None
        self.handle = next(TorchMemoryContext._ids)
        with TorchMemoryContext._lock:
            TorchMemoryContext._live[self.handle] = self

    def get_c_context(self):
        return ctypes.c_void_p(self.handle)

    def get_tensor(self):
        return self.tensor

    def set_tensor(self, t):
        self.tensor = t

    def free(self):
        self.tensor = None

    def release(self):
        with TorchMemoryContext._lock:
            TorchMemoryContext._live.pop(self.handle, None)

    def __del__(self):
        self.release()

    @staticmethod
    def from_handle(h):
        return TorchMemoryContext._live[int(h)]


def _alloc(desc, alloc_type, ctx):
    d = desc.contents
    shape = tuple(d.sizes[i] for i in range(d.dim))
    dtype = wm_dtype_to_torch(d.dtype)
    if alloc_type == L.MA_DEVICE:
        t = torch.empty(shape, dtype=dtype, device="cuda")
    elif alloc_type == L.MA_PINNED:
        t = torch.empty(shape, dtype=dtype, device="cpu", pin_memory=torch.cuda.is_available())
    else:
        t = torch.empty(shape, dtype=dtype, device="cpu")
    ctx.set_tensor(t)
    return t.data_ptr() if t.numel() > 0 else None


_temp_contexts = {}  # handle -> TorchMemoryContext kept alive while the op runs


@L.CREATE_CTX_FN
def _create_ctx(p_ctx, _global):
    c = TorchMemoryContext()
    _temp_contexts[c.handle] = c
    p_ctx[0] = c.handle


@L.DESTROY_CTX_FN
def _destroy_ctx(ctx, _global):
    c = _temp_contexts.pop(int(ctx), None)
    if c is not None:
        c.free()
        c.release()


@L.MALLOC_FN
def _malloc(desc, alloc_type, ctx, _global):
    return _alloc(desc, alloc_type, TorchMemoryContext.from_handle(ctx))


@L.FREE_FN
def _free(ctx, _global):
    TorchMemoryContext.from_handle(ctx).free()


_env = None


def get_wholegraph_env_fns():
    """``wholememory_env_func_t*`` whose callbacks allocate torch tensors (wholegraph_env.py:160-211)."""
    global _env
    if _env is None:
        e = L.EnvFns()
        e.temporary_fns.create_memory_context_fn = _create_ctx
        e.temporary_fns.destroy_memory_context_fn = _destroy_ctx
        e.temporary_fns.malloc_fn = _malloc
        e.temporary_fns.free_fn = _free
        e.temporary_fns.global_context = None
        e.output_fns.malloc_fn = _malloc
        e.output_fns.free_fn = _free
        e.output_fns.global_context = None
        _env = e
    return ctypes.byref(_env)


class WrappedTensor:
    """RAII ``wholememory_tensor_t`` view of a torch tensor (``wrap_torch_tensor``,
    wholegraph_env.py:118-157).  ``None`` wraps to a 0-dim tensor with a NULL pointer, which the
    ops read as "output not requested"."""

    def __init__(self, t):
        self.t = t
        desc = L.TensorDescription()
        L.lib().wholememory_initialize_tensor_desc(ctypes.byref(desc))
        ptr = None
        if t is not None:
            if t.dim() > L.WHOLEMEMORY_MAX_TENSOR_DIM:
                raise ValueError("too many dims")
            desc.dim = t.dim()
            dense = 1  # torch reports arbitrary strides for extent-0/1 dims: normalise them
            for i in reversed(range(t.dim())):
                desc.sizes[i] = t.shape[i]
                desc.strides[i] = t.stride(i) if t.shape[i] > 1 else dense
                dense = desc.strides[i] * max(int(t.shape[i]), 1)
            desc.dtype = torch_dtype_to_wm(t.dtype)
            desc.storage_offset = 0  # data_ptr() already points at the first element of the view
            ptr = t.data_ptr()
        self.c = ctypes.c_void_p()
        L.check(L.lib().wholememory_make_tensor_from_pointer(ctypes.byref(self.c), ctypes.c_void_p(ptr),
                                                             ctypes.byref(desc)),
                "wholememory_make_tensor_from_pointer")

    def __del__(self):
        c = getattr(self, "c", None)
        if c is not None and c.value:
            L.lib().wholememory_destroy_tensor(c)
            self.c = None


def wrap_torch_tensor(t):
    return WrappedTensor(t)
