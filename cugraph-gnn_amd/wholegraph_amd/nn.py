"""GraphSAGE / GAT mini-batch aggregation on the sampler's per-hop CSR (HIP kernels of
``include/wgamd_ext.h``), packaged as the conv layers the reference's models instantiate.

The reference has no aggregation kernel: ``HomoGNNModel`` builds ``torch_geometric.nn.SAGEConv`` /
``GATConv`` and feeds them ``(x, x_target)`` plus the hop's ``[csr_row_ptr, csr_col_ind]`` or COO
(/root/reference/python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:25-59,119-125,178-199).
``SAGEConv`` / ``GATConv`` below keep PyG's parameter names (``lin_l``, ``lin_r``, ``lin``,
``att_src``, ``att_dst``, ``bias``) and maths so they are a drop-in for that call shape; the
segmented reduce / edge softmax run in hand-written gfx950 kernels, the dense ``lin_*`` tail is a
plain ``torch.nn.functional.linear`` (hipBLASLt, MFMA).
"""
import math
import os
from typing import Optional, Tuple, Union

import torch

from . import _lib as L
from .env import get_stream, torch_dtype_to_wm


def _ids_code(x, src_ids) -> int:
    """``src_ids_dtype`` of the layer kernels: the ids' integer type, or WGAMD_IDS_BYTE_OFFSETS when ``x`` is the address space of
    a peer-mapped table (``MappedTable``) and ``src_ids`` holds byte offsets into it."""
    return L.IDS_BYTE_OFFSETS if getattr(x, "byte_offset_ids", False) else torch_dtype_to_wm(src_ids.dtype)


def _check_csr(row_ptr, col):
    assert row_ptr.dtype == torch.int32 and col.dtype == torch.int32, "per-hop CSR is int32 (sampler output)"
    assert row_ptr.is_cuda and col.is_cuda and row_ptr.is_contiguous() and col.is_contiguous()


def spmm_csr_forward(row_ptr, col, x, mean=True, src_ids=None, out=None):
    """out[i] = mean/sum_{e in row i} x[src(e)], src(e) = col[e] or src_ids[col[e]] (fused fetch)."""
    _check_csr(row_ptr, col)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    n_rows = row_ptr.shape[0] - 1
    if out is None:
        out = torch.empty((n_rows, x.shape[1]), dtype=torch.float32, device=x.device)
    ids_ptr, ids_dt = None, 0
    if src_ids is not None:
        assert src_ids.is_contiguous()
        ids_ptr, ids_dt = src_ids.data_ptr(), torch_dtype_to_wm(src_ids.dtype)
    L.check(L.lib().wgamd_spmm_csr_f32(row_ptr.data_ptr(), col.data_ptr(), n_rows, x.data_ptr(), x.stride(0),
                                       x.shape[1], ids_ptr, ids_dt, int(bool(mean)), out.data_ptr(),
                                       out.stride(0), get_stream()), "wgamd_spmm_csr_f32")
    return out


def sage_aggregate_forward(row_ptr, col, x, self_rows, mean=True):
    """-> [n_rows, 2F] = [ mean_{e in row i} x[col[e]] | x[self_rows[i]] ]  (one kernel; feeds ONE GEMM with
    the concatenated weight [W_l | W_r])."""
    _check_csr(row_ptr, col)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    assert self_rows.dtype == torch.int64 and self_rows.is_contiguous()
    n_rows, F_ = row_ptr.shape[0] - 1, x.shape[1]
    assert self_rows.shape[0] == n_rows
    out = torch.empty((n_rows, 2 * F_), dtype=torch.float32, device=x.device)
    L.check(L.lib().wgamd_sage_aggregate_f32(row_ptr.data_ptr(), col.data_ptr(), n_rows, x.data_ptr(), x.stride(0), F_,
                                             self_rows.data_ptr(), int(bool(mean)), out.data_ptr(), out.stride(0),
                                             get_stream()), "wgamd_sage_aggregate_f32")
    return out


def sage_aggregate_fetch_forward(row_ptr, col, table, src_ids, self_rows, mean=True):
    """``sage_aggregate_forward`` with the feature fetch fused in: ``table`` is the global feature table and
    ``src_ids`` the batch's local->global map, so ``x = table[src_ids]`` is never written to HBM."""
    _check_csr(row_ptr, col)
    assert table.dtype == torch.float32 and table.dim() == 2 and table.stride(1) == 1
    assert self_rows.dtype == torch.int64 and self_rows.is_contiguous() and src_ids.is_contiguous()
    n_rows, F_ = row_ptr.shape[0] - 1, table.shape[1]
    out = torch.empty((n_rows, 2 * F_), dtype=torch.float32, device=table.device)
    L.check(L.lib().wgamd_sage_aggregate_fetch_f32(
        row_ptr.data_ptr(), col.data_ptr(), n_rows, table.data_ptr(), table.stride(0), F_, src_ids.data_ptr(),
        torch_dtype_to_wm(src_ids.dtype), self_rows.data_ptr(), int(bool(mean)), out.data_ptr(), out.stride(0),
        get_stream()), "wgamd_sage_aggregate_fetch_f32")
    return out


_FUSED_PRECISION = "bf16x3"      # "bf16x3": 3-way bf16 split of both operands on the bf16 matrix pipe (fp32-class accuracy,
#                                  HBM-bound); "f32": exact fp32 MFMA (v_mfma_f32_16x16x4_f32, bound by the fp32 matrix rate)


def sage_layer_fused_precision() -> str:
    return _FUSED_PRECISION


def set_sage_layer_fused_precision(mode: str) -> None:
    global _FUSED_PRECISION
    assert mode in ("bf16x3", "f32")
    _FUSED_PRECISION = mode


def _pick_precision(F_: int, N: int, precision) -> str:
    mode = precision or _FUSED_PRECISION
    if mode == "bf16x3" and not L.lib().wgamd_sage_layer_bf16x3_supported(F_, _padded_width(N)):
        mode = "f32"
    return mode


def _padded_width(N: int) -> int:
    """Output width the one-kernel layer runs at: its consumer waves own 64 columns each, so N is rounded up to 64, 128 or
    256 with zero weight columns (a 47-class head runs as N = 64; the caller sees the first N columns)."""
    return 64 if N <= 64 else 128 if N <= 128 else 256


def sage_layer_fused_supported(F_: int, N: int) -> bool:
    """Shapes the one-kernel SAGE layer is built for (include/wgamd_ext.h); N is padded to 64 / 128 / 256 on the way in."""
    return F_ % 4 == 0 and 0 < F_ <= 256 and 0 < N <= 256


def sage_layer_fused_preferred(F_: int, N: int) -> bool:
    """Shapes where the one-kernel layer beats aggregate kernel + library GEMM.  bf16x3 kernel: every shape it supports,
    a padded head included (the 47-class layer of the products model runs as N = 64 with one multiplying wave per CU: with
    the round-3 half-tile kernel 0.38 ms against 0.54 ms for aggregate + GEMM at 196 k rows, 256 -> 64; 0.043 against 0.052
    at 16 k rows; the round-2 kernel lost there, 0.46 against 0.20 at 65 k rows).  fp32-MFMA kernel: only when two operand
    tiles fit the 160 KB of LDS and the width needs no padding."""
    if not sage_layer_fused_supported(F_, N):
        return False
    if _FUSED_PRECISION == "bf16x3" and L.lib().wgamd_sage_layer_bf16x3_supported(F_, _padded_width(N)):
        return True
    return N == _padded_width(N) and F_ <= 152


# ---- derived-weight caches and HIP-graph capture ---------------------------------------------------------------------------
# The transposed / padded / bf16-split forms of a layer's weights are cached against the weight tensor's version counter.  A
# captured training step (cugraph_pyg_amd.loader.PerBatchStep) updates the weights INSIDE the graph: Python does not run on a
# replay, so (a) while a stream is capturing the derived forms are rebuilt unconditionally — their kernels become part of the
# graph and run on every replay — and nothing is cached, and (b) every replay bumps `_weights_gen`, which is part of every
# cache key, so eager code that follows a replay never sees a form derived from the weights of an earlier step.
_weights_gen = 0
_capture_epoch = 0       # bumped by begin_capture(): per-capture caches (a hop's transpose) are keyed on it


def begin_capture():
    """Call right before capturing a HIP graph that runs layers over fixed, refilled graph buffers."""
    global _capture_epoch
    _capture_epoch += 1


def bump_weight_generation():
    """Invalidate every cached derived weight (call after the weights changed behind autograd's back, e.g. a graph replay)."""
    global _weights_gen
    _weights_gen += 1


def _capturing() -> bool:
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _padded_head(w_t: torch.Tensor, bias, Np: int):
    """``w_t`` [2F, N] and ``bias`` [N] with zero columns up to Np, cached on the weight tensor (version-checked)."""
    hit = getattr(w_t, "_wgamd_padded", None) if not _capturing() else None
    key = (w_t._version, w_t.data_ptr(), None if bias is None else (bias._version, bias.data_ptr()), Np, _weights_gen)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    wp = torch.zeros((w_t.shape[0], Np), dtype=w_t.dtype, device=w_t.device)
    wp[:, :w_t.shape[1]] = w_t
    bp = None
    if bias is not None:
        bp = torch.zeros(Np, dtype=bias.dtype, device=bias.device)
        bp[:bias.shape[0]] = bias
    if not _capturing():
        try:
            w_t._wgamd_padded = (key, wp, bp)
        except AttributeError:
            pass
    return wp, bp


def sage_weight_planes(w_t: torch.Tensor) -> torch.Tensor:
    """``w_t`` [2F, N] fp32 -> the three bf16 planes ``wgamd_sage_layer_fused_bf16x3`` multiplies with (exact 3-way split:
    hi + mid + lo == w).  Cached ON the weight tensor object together with its version counter, so a layer pays for the
    split once per optimizer step and the cache can never outlive (or be confused with another tensor at) the same address."""
    hit = getattr(w_t, "_wgamd_planes", None) if not _capturing() else None
    if hit is not None and hit[0] == (w_t._version, _weights_gen) and hit[1] == w_t.data_ptr():
        return hit[2]
    K, N = w_t.shape
    planes = torch.empty(L.lib().wgamd_sage_weight_planes_bytes(K, N), dtype=torch.uint8, device=w_t.device)
    L.check(L.lib().wgamd_sage_split_weight_bf16x3(w_t.data_ptr(), w_t.stride(0), K, N, planes.data_ptr(), get_stream()),
            "wgamd_sage_split_weight_bf16x3")
    if not _capturing():
        try:
            w_t._wgamd_planes = ((w_t._version, _weights_gen), w_t.data_ptr(), planes)
        except AttributeError:      # a tensor subclass without a __dict__: split on every call
            pass
    return planes


SAGE_FULL_TILES = 2          # WGAMD_SAGE_FULL_TILES (include/wgamd_ext.h): or-ed into the relu argument of a small launch
_SAGE_SMALL_ROWS = int(os.environ.get("WGAMD_SAGE_SMALL_ROWS", 8192))


def sage_layer_small_launch(F_: int, n_rows: int) -> bool:
    """A launch of one mini-batch's rows at a width whose throughput shape is 64-row half tiles (F > 148): whole 32-row tiles
    give it twice the workgroups, each with half the serial chain of row fetches (47 -> 30 us for 1.2 k rows at F = 256)."""
    return 0 < n_rows <= _SAGE_SMALL_ROWS and bool(L.lib().wgamd_sage_layer_uses_half_tiles(int(F_)))


def sage_layer_planes(w_l: torch.Tensor, w_r: torch.Tensor, bias, Np: int, full_tiles: bool = False):
    """``(planes, padded bias, N)`` of a layer straight from its ``torch.nn.Linear`` parameters in ONE launch
    (``wgamd_sage_layer_weight_planes``) — what ``sage_layer_fused_forward(prepared=...)`` takes instead of deriving the
    transposed / padded / split forms with half a dozen framework launches.  Not cached: made for captured training steps,
    whose weights change on every replay."""
    N, F_ = w_l.shape
    planes = torch.empty(L.lib().wgamd_sage_weight_planes_bytes(2 * F_, Np), dtype=torch.uint8, device=w_l.device)
    bias_p = torch.empty(Np, dtype=torch.float32, device=w_l.device) if (bias is not None or Np != N) else None
    L.check(L.lib().wgamd_sage_layer_weight_planes(w_l.data_ptr(), w_l.stride(0), w_r.data_ptr(), w_r.stride(0),
                                                   None if bias is None else bias.data_ptr(), F_, N, Np, planes.data_ptr(),
                                                   None if bias_p is None else bias_p.data_ptr(), int(bool(full_tiles)), get_stream()),
            "wgamd_sage_layer_weight_planes")
    return planes, bias_p, N, bool(full_tiles)


def sage_layer_fused_forward(row_ptr, col, x, self_rows, w_t, bias=None, relu=False, mean=True, src_ids=None, out=None,
                             precision=None, agg_out=None, prepared=None):
    """A whole SAGEConv layer over a sampled hop in ONE kernel: ``act([mean_j X[col_j] | X[self_i]] @ w_t + bias)`` with
    ``X[r] = x[src_ids[r]]`` when ``src_ids`` is given (``x`` is then the global feature table: the feature fetch is fused
    in too).  ``w_t`` = ``cat([W_l, W_r], 1).t()`` ([2F, N], contiguous).  The ``[n_rows, 2F]`` operand never leaves LDS.
    ``precision``: "bf16x3" (default where the shape allows) or "f32" — see ``_FUSED_PRECISION``.
    ``agg_out`` ([n_rows, F] fp32, training): the launch also keeps the aggregate half of its operand there
    (``wgamd_sage_layer_fused_bf16x3_train``; same bits in ``out``) — needs ``sage_layer_train_supported``."""
    _check_csr(row_ptr, col)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    assert self_rows.dtype == torch.int64 and self_rows.is_contiguous()
    if prepared is not None:      # (planes, padded bias, N) of sage_layer_planes: the bf16x3 kernel's operand, ready made
        n_rows, F_, N = row_ptr.shape[0] - 1, x.shape[1], prepared[2]
        assert sage_layer_fused_supported(F_, _padded_width(N)) and _pick_precision(F_, _padded_width(N), precision) == "bf16x3"
        bias = prepared[1]
    else:
        assert w_t.dtype == torch.float32 and w_t.dim() == 2 and w_t.stride(1) == 1 and w_t.shape[0] == 2 * x.shape[1]
        n_rows, F_, N = row_ptr.shape[0] - 1, x.shape[1], w_t.shape[1]
    Np = _padded_width(N) if sage_layer_fused_supported(F_, N) else N
    user_out = None
    if Np != N and prepared is not None:
        if out is not None and out.stride(0) != Np:
            assert out.shape == (n_rows, N) and out.dtype == torch.float32, "out must be float32 [n_rows, N]"
            user_out, out = out, None
    elif Np != N:
        # zero weight columns / bias entries up to the width the kernel runs at (cached on the weight like its planes); the
        # kernel then WRITES Np columns per row.  A caller's `out` is written in place only when it is the [:, :N] view of
        # an explicit [n_rows, Np] scratch (row stride == Np: columns N..Np-1 are the caller's padding by construction);
        # any other `out` is filled from a temporary, so neither a wider buffer's own columns are overwritten nor a
        # narrower one silently dropped.
        w_t, bias = _padded_head(w_t, bias, Np)
        if out is not None and out.stride(0) != Np:
            assert out.shape == (n_rows, N) and out.dtype == torch.float32, "out must be float32 [n_rows, N]"
            user_out, out = out, None
    if out is None:
        out = torch.empty((n_rows, Np), dtype=torch.float32, device=x.device)[:, :N]
    assert out.shape == (n_rows, N) and out.dtype == torch.float32 and out.stride(1) == 1

    def done(res):
        if user_out is None:
            return res
        user_out.copy_(res)
        return user_out
    N = Np
    ids_ptr, ids_dt = None, 0
    if src_ids is not None:
        assert src_ids.is_contiguous()
        ids_ptr, ids_dt = src_ids.data_ptr(), _ids_code(x, src_ids)
    assert ids_dt != L.IDS_BYTE_OFFSETS or (sage_layer_fused_supported(F_, N) and _pick_precision(F_, N, precision) == "bf16x3"), \
        "a peer-mapped table is read by the bf16x3 layer kernel only"
    if (sage_layer_fused_supported(F_, N) and _pick_precision(F_, N, precision) == "bf16x3"
            and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0):      # its epilogue stores 16 B per lane
        planes = prepared[0] if prepared is not None else sage_weight_planes(w_t)
        flags = int(bool(relu)) | (SAGE_FULL_TILES if prepared is not None and prepared[3] else 0)
        if agg_out is not None:
            assert agg_out.shape == (n_rows, F_) and agg_out.dtype == torch.float32 and agg_out.stride(1) == 1
            L.check(L.lib().wgamd_sage_layer_fused_bf16x3_train(
                row_ptr.data_ptr(), col.data_ptr(), n_rows, x.data_ptr(), x.stride(0), x.shape[0], F_, ids_ptr, ids_dt,
                self_rows.data_ptr(), int(bool(mean)), planes.data_ptr(), N, None if bias is None else bias.data_ptr(),
                flags, out.data_ptr(), out.stride(0), agg_out.data_ptr(), agg_out.stride(0), get_stream()),
                "wgamd_sage_layer_fused_bf16x3_train")
            return done(out)
        L.check(L.lib().wgamd_sage_layer_fused_bf16x3(
            row_ptr.data_ptr(), col.data_ptr(), n_rows, x.data_ptr(), x.stride(0), x.shape[0], F_, ids_ptr, ids_dt,
            self_rows.data_ptr(), int(bool(mean)), planes.data_ptr(), N, None if bias is None else bias.data_ptr(),
            flags, out.data_ptr(), out.stride(0), get_stream()), "wgamd_sage_layer_fused_bf16x3")
        return done(out)
    assert agg_out is None, "agg_out: only the bf16x3 layer kernel keeps the aggregate (sage_layer_train_supported)"
    L.check(L.lib().wgamd_sage_layer_fused_f32(
        row_ptr.data_ptr(), col.data_ptr(), n_rows, x.data_ptr(), x.stride(0), x.shape[0], F_, ids_ptr, ids_dt,
        self_rows.data_ptr(),
        int(bool(mean)), w_t.data_ptr(), w_t.stride(0), N, None if bias is None else bias.data_ptr(), int(bool(relu)),
        out.data_ptr(), out.stride(0), get_stream()), "wgamd_sage_layer_fused_f32")
    return done(out)


def sage_layer_train_supported(F_: int, N: int) -> bool:
    """Shapes whose one-kernel layer has a backward pass on the HIP kernels: the bf16x3 layer kernel (it is the one that keeps
    the aggregate for the backward) and ``wgamd_sage_wgrad_bf16x3``."""
    return (sage_layer_fused_supported(F_, N) and _FUSED_PRECISION == "bf16x3"
            and bool(L.lib().wgamd_sage_layer_bf16x3_supported(F_, _padded_width(N)))
            and L.lib().wgamd_sage_wgrad_workspace_bytes(1, F_, N) > 0)


_WGRAD_WS = {}


def _wgrad_workspace(n_rows: int, F_: int, N: int, device) -> torch.Tensor:
    """Scratch of the weight-gradient launches (the workgroups' partial sums + the composed self rows), grown on demand and
    kept per device: every use is stream-ordered on the caller's stream."""
    need = L.lib().wgamd_sage_wgrad_workspace_bytes(int(n_rows), F_, N)
    if _capturing():       # a captured graph must own its scratch: the shared buffer may be re-allocated by a later eager call
        return torch.empty(int(need) + 4096, dtype=torch.uint8, device=device)
    ws = _WGRAD_WS.get(device)
    if ws is None or ws.numel() < need:
        ws = torch.empty(int(need * 1.15) + 4096, dtype=torch.uint8, device=device)
        _WGRAD_WS[device] = ws
    return ws


def sage_wgrad(agg, x, self_rows, grad_out, grad_w_l, grad_w_r, grad_bias=None, act_out=None, src_ids=None, accumulate=False):
    """Weight gradient of the one-kernel SAGE layer over one hop (``wgamd_sage_wgrad_bf16x3``):
    ``grad_w_l (+)= dZ^T agg``, ``grad_w_r (+)= dZ^T X[self_rows]``, ``grad_bias (+)= sum_i dZ[i]`` with
    ``dZ = grad_out * (act_out > 0)`` (``act_out`` = the layer's ReLU output, None = no activation) and
    ``X[r] = x[src_ids[r]]`` when ``src_ids`` is given.  ``grad_w_*`` are [N, F] contiguous fp32."""
    n, F_ = agg.shape
    N = grad_out.shape[1]
    assert agg.dtype == torch.float32 and agg.stride(1) == 1 and x.dtype == torch.float32 and x.stride(1) == 1 and x.shape[1] == F_
    assert grad_out.dtype == torch.float32 and grad_out.stride(1) == 1 and grad_out.shape[0] == n
    assert self_rows.dtype == torch.int64 and self_rows.is_contiguous() and self_rows.shape[0] == n
    assert grad_w_l.shape == (N, F_) and grad_w_l.is_contiguous() and grad_w_r.shape == (N, F_) and grad_w_r.is_contiguous()
    assert act_out is None or (act_out.shape == grad_out.shape and act_out.stride(1) == 1)
    assert grad_bias is None or (grad_bias.shape == (N,) and grad_bias.is_contiguous())
    ids_ptr, ids_dt = None, 0
    if src_ids is not None:
        assert src_ids.is_contiguous()
        ids_ptr, ids_dt = src_ids.data_ptr(), _ids_code(x, src_ids)
    ws = _wgrad_workspace(n, F_, N, agg.device)
    L.check(L.lib().wgamd_sage_wgrad_bf16x3(
        agg.data_ptr(), agg.stride(0), x.data_ptr(), x.stride(0), F_, ids_ptr, ids_dt, self_rows.data_ptr(), n,
        grad_out.data_ptr(), grad_out.stride(0), None if act_out is None else act_out.data_ptr(),
        0 if act_out is None else act_out.stride(0), N, grad_w_l.data_ptr(), grad_w_r.data_ptr(),
        None if grad_bias is None else grad_bias.data_ptr(), int(bool(accumulate)), ws.data_ptr(), ws.numel(), get_stream()),
        "wgamd_sage_wgrad_bf16x3")


def _csr_transpose(row_ptr, col, n_src, want_perm=False, want_dst=False, want_col_t=False):
    """wgamd_csr_transpose_i32: (row_ptr_t, edge_perm | None, edge_dst | None, col_t | None)."""
    _check_csr(row_ptr, col)
    n_dst, E, dev = row_ptr.shape[0] - 1, col.shape[0], col.device
    i32 = dict(dtype=torch.int32, device=dev)
    row_ptr_t = torch.empty(n_src + 1, **i32)
    perm = torch.empty(E, **i32) if want_perm else None
    dst = torch.empty(E, **i32) if want_dst else None
    col_t = torch.empty(E, **i32) if want_col_t else None
    need = L.lib().wgamd_csr_transpose_workspace_bytes(E, n_src)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    L.check(L.lib().wgamd_csr_transpose_i32(row_ptr.data_ptr(), col.data_ptr(), n_dst, E, n_src, row_ptr_t.data_ptr(),
                                            ptr(perm), ptr(dst), ptr(col_t), ws.data_ptr(), need, get_stream()),
            "wgamd_csr_transpose_i32")
    return row_ptr_t, perm, dst, col_t


def csr_transpose(row_ptr, col, n_src):
    """Destination-major hop CSR -> source-major CSR (rows = sources, entries = destination rows, in edge order: a
    stable radix sort over the bits a source row needs, so the gradient sums below are run-to-run deterministic)."""
    row_ptr_t, _, _, col_t = _csr_transpose(row_ptr, col, n_src, want_col_t=True)
    return row_ptr_t, col_t


def spmm_csr_backward(row_ptr, col, grad_out, n_src, mean=True, atomic=False):
    """grad_x[j] = sum_{edges (i, j)} grad_out[i] / (deg(i) if mean).  Default: transpose the hop CSR once and run the
    forward gather kernel over it (no atomics, deterministic; 7x faster than the scatter-add at products sizes);
    ``atomic=True`` keeps the one-kernel ``wgamd_spmm_csr_bwd_f32`` scatter-add."""
    g = grad_out.contiguous()
    if atomic:
        gx = torch.zeros((n_src, g.shape[1]), dtype=torch.float32, device=g.device)
        L.check(L.lib().wgamd_spmm_csr_bwd_f32(row_ptr.data_ptr(), col.data_ptr(), row_ptr.shape[0] - 1,
                                               g.data_ptr(), g.stride(0), g.shape[1], int(bool(mean)),
                                               gx.data_ptr(), gx.stride(0), get_stream()),
                "wgamd_spmm_csr_bwd_f32")
        return gx
    if mean:
        deg = (row_ptr[1:] - row_ptr[:-1]).clamp_(min=1)
        g = g / deg.unsqueeze(1)
    row_ptr_t, col_t = csr_transpose(row_ptr, col, n_src)
    if not _BWD_SEGMENTS:
        return spmm_csr_forward(row_ptr_t, col_t, g, mean=False)
    # The transposed hop is power-law (a hub is a neighbour of thousands of sampled rows) while the gather kernel walks a
    # row with one lane group: the launch would last as long as its longest row.  wgamd_spmm_csr_segmented_f32 sums rows
    # in pieces of 64 entries and adds the pieces of a row up in order (deterministic, nothing read back).
    E, F_ = col_t.shape[0], g.shape[1]
    out = torch.empty((n_src, F_), dtype=torch.float32, device=g.device)
    need = L.lib().wgamd_spmm_csr_segmented_workspace_bytes(E, F_)
    ws = torch.empty(need, dtype=torch.uint8, device=g.device)
    L.check(L.lib().wgamd_spmm_csr_segmented_f32(row_ptr_t.data_ptr(), col_t.data_ptr(), n_src, E, g.data_ptr(), g.stride(0), F_,
                                                 out.data_ptr(), out.stride(0), ws.data_ptr(), need, get_stream()),
            "wgamd_spmm_csr_segmented_f32")
    return out


_BWD_SEGMENTS = os.environ.get("WGAMD_SPMM_BWD_SEGMENTS", "1") != "0"



class _SpmmCsr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, row_ptr, col, mean):
        ctx.save_for_backward(row_ptr, col)
        ctx.mean, ctx.n_src = mean, x.shape[0]
        return spmm_csr_forward(row_ptr, col, x.contiguous(), mean)

    @staticmethod
    def backward(ctx, grad_out):
        row_ptr, col = ctx.saved_tensors
        return spmm_csr_backward(row_ptr, col, grad_out, ctx.n_src, ctx.mean), None, None, None


class _SoftmaxXent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, row_weight):
        n, C = logits.shape
        lib, dev = L.lib(), logits.device
        lse = torch.empty(n, dtype=torch.float32, device=dev)
        state = torch.zeros((lib.wgamd_softmax_xent_state_bytes(n) + 3) // 4, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        L.check(lib.wgamd_softmax_xent_forward_f32(logits.data_ptr(), logits.stride(0), n, C, target.data_ptr(),
                                                   row_weight.data_ptr() if row_weight is not None else None, lse.data_ptr(),
                                                   state.data_ptr(), 1, loss.data_ptr(), get_stream()), "wgamd_softmax_xent_forward_f32")
        ctx.save_for_backward(logits, target, lse, state)
        ctx.row_weight = row_weight
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        logits, target, lse, state = ctx.saved_tensors
        n, C = logits.shape
        g = grad_loss.to(torch.float32).contiguous()
        gx = torch.empty((n, C), dtype=torch.float32, device=logits.device)
        w = ctx.row_weight
        L.check(L.lib().wgamd_softmax_xent_backward_f32(logits.data_ptr(), logits.stride(0), n, C, target.data_ptr(),
                                                        w.data_ptr() if w is not None else None, lse.data_ptr(), state.data_ptr(),
                                                        g.data_ptr(), gx.data_ptr(), gx.stride(0), get_stream()),
                "wgamd_softmax_xent_backward_f32")
        return gx, None, None


def cross_entropy(logits: torch.Tensor, target: torch.Tensor, row_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``torch.nn.functional.cross_entropy(logits, target)`` (mean over the rows whose target is not negative — torch's
    ``ignore_index``), optionally with a float weight per ROW (``loader.StagedBatch.seed_mask``: the padding of a ragged last
    mini-batch) — as two launches, forward and backward, instead of torch's seven: the loss of the reference's training loops
    (pylibwholegraph/torch/gnn_model.py:119-125) inside a per-mini-batch step whose every launch sits on its latency floor.
    fp32 logits [n, C] with unit column stride, int64 targets; anything else goes to torch."""
    if (logits.dim() != 2 or logits.dtype != torch.float32 or not logits.is_cuda or logits.stride(1) != 1 or logits.shape[0] == 0
            or target.dtype != torch.int64 or target.shape != logits.shape[:1]):
        loss = torch.nn.functional.cross_entropy(logits, target, reduction="none", ignore_index=-100)
        w = (target >= 0).to(loss.dtype) if row_weight is None else row_weight * (target >= 0)
        return (loss * w).sum() / w.sum()
    target = target.contiguous()
    if row_weight is not None:
        row_weight = row_weight.to(torch.float32).contiguous()
    return _SoftmaxXent.apply(logits, target, row_weight)


def spmm_csr(x, row_ptr, col, reduce: str = "mean"):
    """Differentiable segmented mean/sum aggregation over a destination-major CSR."""
    assert reduce in ("mean", "sum", "add")
    return _SpmmCsr.apply(x, row_ptr, col, reduce == "mean")


def gat_forward(row_ptr, col, x, a_src, a_dst, heads, negative_slope=0.2, need_alpha=True):
    """Edge softmax + weighted aggregation.  x [N_src, H*C], a_src [N_src, H], a_dst [n_rows, H]."""
    _check_csr(row_ptr, col)
    n_rows = row_ptr.shape[0] - 1
    HC = x.shape[1]
    C = HC // heads
    out = torch.empty((n_rows, HC), dtype=torch.float32, device=x.device)
    alpha = torch.empty((col.shape[0], heads), dtype=torch.float32, device=x.device) if need_alpha else None
    L.check(L.lib().wgamd_gat_csr_f32(row_ptr.data_ptr(), col.data_ptr(), n_rows, x.data_ptr(), x.stride(0),
                                      a_src.data_ptr(), a_dst.data_ptr(), heads, C, float(negative_slope),
                                      alpha.data_ptr() if need_alpha else None, out.data_ptr(), out.stride(0),
                                      get_stream()), "wgamd_gat_csr_f32")
    return out, alpha


def gat_forward_rows(row_ptr, col, x, a_src, a_dst, heads, out, dst_rows=None, accumulate=False, negative_slope=0.2):
    """``gat_forward`` for the rows of ONE hop and edge type of a heterogeneous call group: row i of the launch reads
    ``a_dst[dst_rows[i]]`` and writes (``accumulate``: adds to) ``out[dst_rows[i]]`` — see wgamd_gat_csr_rows_f32."""
    _check_csr(row_ptr, col)
    n_rows = row_ptr.shape[0] - 1
    C = x.shape[1] // heads
    assert out.dtype == torch.float32 and out.stride(1) == 1 and out.shape[1] == x.shape[1]
    assert dst_rows is None or (dst_rows.dtype == torch.int64 and dst_rows.is_contiguous() and dst_rows.shape[0] >= n_rows)
    L.check(L.lib().wgamd_gat_csr_rows_f32(row_ptr.data_ptr(), col.data_ptr(), n_rows, x.data_ptr(), x.stride(0),
                                           a_src.data_ptr(), a_dst.data_ptr(), heads, C, float(negative_slope),
                                           None if dst_rows is None else dst_rows.data_ptr(), int(bool(accumulate)), None,
                                           out.data_ptr(), out.stride(0), get_stream()), "wgamd_gat_csr_rows_f32")
    return out


def _gat_ids(src_ids, dst_ids, a_src, src_terms_by_id, dst_terms_by_id):
    """(src_ids pointer, dst_ids pointer, terms_by_id) of a fetch-in-the-layer GAT launch: int64 contiguous lists; without
    ``src_terms_by_id`` the attention terms have one row per listed id."""
    if src_ids is None:
        assert not src_terms_by_id and not dst_terms_by_id
        return None, None, 0
    assert src_ids.dtype == torch.int64 and src_ids.is_contiguous() and (src_terms_by_id or src_ids.shape[0] == a_src.shape[0])
    assert not dst_terms_by_id or (dst_ids is not None and dst_ids.dtype == torch.int64 and dst_ids.is_contiguous())
    return src_ids.data_ptr(), dst_ids.data_ptr() if dst_terms_by_id else None, int(bool(src_terms_by_id)) | 2 * int(bool(dst_terms_by_id))


def gat_aggregate_heads(row_ptr, col, x, a_src, a_dst, heads, dst_rows=None, negative_slope=0.2, out=None, src_ids=None, dst_ids=None,
                        src_terms_by_id=False, dst_terms_by_id=False):
    """Aggregate-first GAT (wgamd_gat_aggregate_heads_f32): ``agg[i, h, :] = sum_e alpha_e^h x[col[e], :]`` with x
    untransformed ([N_src, F]); returns ``[n_rows, heads * F]``.  ``gat_transform_heads`` applies the per-head weights.
    ``src_ids`` (int64 [N_src]): the rows are read THROUGH the list, ``x[src_ids[col[e]]]`` — x the feature table, src_ids the
    call group's node list (``LazyRows``); the attention terms stay indexed by ``col`` — unless ``src_terms_by_id``: ``a_src`` then
    holds the terms of the TABLE's rows, read at ``src_ids[col[e]]`` (``dst_terms_by_id``: ``a_dst`` at ``dst_ids[dst row]``)."""
    _check_csr(row_ptr, col)
    ids_p, dids_p, by_id = _gat_ids(src_ids, dst_ids, a_src, src_terms_by_id, dst_terms_by_id)
    n_rows, F_ = row_ptr.shape[0] - 1, x.shape[1]
    if out is None:
        out = torch.empty((n_rows, heads * F_), dtype=torch.float32, device=x.device)
    assert a_src.is_contiguous() and a_dst.is_contiguous() and a_src.shape[1] == heads
    L.check(L.lib().wgamd_gat_aggregate_heads_ids_f32(row_ptr.data_ptr(), col.data_ptr(), n_rows, x.data_ptr(), x.stride(0),
                                                      ids_p, dids_p, by_id, F_, a_src.data_ptr(), a_dst.data_ptr(), heads,
                                                      float(negative_slope), None if dst_rows is None else dst_rows.data_ptr(),
                                                      out.data_ptr(), out.stride(0), get_stream()), "wgamd_gat_aggregate_heads_ids_f32")
    return out


def gat_transform_heads(agg, w, heads, out=None, overwrite=False, fused=None):
    """``out[i, h*C:(h+1)*C] (+)= agg[i, h, :] @ w[:, h*C:(h+1)*C]`` — the H small GEMMs after ``gat_aggregate_heads``.
    ``out`` given: accumulated into (HeteroConv's sum), or written (``overwrite``).  Shapes
    ``wgamd_gat_transform_heads_bf16x3`` is built for (C = 64, F in {64, 128, 256}) run on it (``gat_transform_heads_fused``:
    the bf16 matrix pipe at fp32 accuracy, one pass); ``fused=False`` or any other shape: one strided batched library GEMM
    (fp32 MFMA through hipBLASLt)."""
    n, F_ = agg.shape[0], agg.shape[1] // heads
    C = w.shape[1] // heads
    if fused is None:
        fused = _GAT_TRANSFORM_FUSED
    if fused and n > 0 and agg.is_cuda and agg.stride(1) == 1 and w.stride(1) == 1 and gat_transform_supported(F_, heads, C) \
            and (out is None or out.stride(1) == 1):
        return gat_transform_heads_fused(agg, w, heads, acc_in=None if (out is None or overwrite) else out, out=out)
    a, b = agg.view(n, heads, F_).permute(1, 0, 2), w.view(F_, heads, C).permute(1, 0, 2)
    if out is None:
        return torch.bmm(a, b).permute(1, 0, 2).reshape(n, heads * C)                                    # [H, n, C] -> [n, H C]
    # accumulate in place through the strided view (beta = 1): no [H, n, C] temporary and no separate add pass — same bits,
    # 1.48 -> 1.15 ms at 1 M rows, 4 x 128 -> 4 x 64
    # ``overwrite``: beta = 0 — ``out`` need not be initialised (the first relation of HeteroConv's sum: no zero-fill, no read)
    acc = out.view(n, heads, C).permute(1, 0, 2)
    torch.baddbmm(acc, a, b, beta=0 if overwrite else 1, out=acc)
    return out


_GAT_TRANSFORM_FUSED = os.environ.get("WGAMD_GAT_TRANSFORM", "bf16x3") != "library"
_GAT_LAYER_FUSED = os.environ.get("WGAMD_GAT_LAYER", "fused") != "split"      # one-kernel relation for hops of fan-out <= 10


def gat_transform_supported(F_: int, heads: int, C: int) -> bool:
    return bool(L.lib().wgamd_gat_transform_heads_supported(int(F_), int(heads), int(C)))


def _gat_weight_tiles(w: torch.Tensor, heads: int) -> torch.Tensor:
    """``w`` [F, H C] in the order ``wgamd_gat_transform_heads_bf16x3`` reads it; cached on the weight like ``sage_weight_planes``."""
    hit = getattr(w, "_wgamd_gat_tiles", None)
    if hit is not None and hit[0] == (w._version, _weights_gen) and hit[1] == w.data_ptr():
        return hit[2]
    F_, C = w.shape[0], w.shape[1] // heads
    tiles = torch.empty(L.lib().wgamd_gat_transform_weight_bytes(F_, heads, C), dtype=torch.uint8, device=w.device)
    L.check(L.lib().wgamd_gat_transform_weight_tiles(w.data_ptr(), w.stride(0), F_, heads, C, tiles.data_ptr(), get_stream()),
            "wgamd_gat_transform_weight_tiles")
    try:
        w._wgamd_gat_tiles = ((w._version, _weights_gen), w.data_ptr(), tiles)
    except AttributeError:
        pass
    return tiles


def gat_transform_heads_fused(agg, w, heads, acc_in=None, bias=None, relu=False, out_rows=None, out=None):
    """``out[out_rows[i]] = act(agg[i, h, :] @ w[:, h C:(h+1) C] + acc_in[i] + bias)`` in ONE kernel
    (``wgamd_gat_transform_heads_bf16x3``: the per-head GEMMs on the bf16 matrix pipe at fp32 accuracy, HeteroConv's running sum,
    bias, ReLU and the row placement).  ``acc_in`` may be ``out`` itself when ``out_rows`` is None."""
    n, F_ = agg.shape[0], agg.shape[1] // heads
    C = w.shape[1] // heads
    assert agg.dtype == torch.float32 and agg.stride(1) == 1 and w.dtype == torch.float32 and w.stride(1) == 1 and w.shape[0] == F_
    if out is None:
        assert out_rows is None
        out = torch.empty((n, heads * C), dtype=torch.float32, device=agg.device)
    assert out.stride(1) == 1 and (acc_in is None or acc_in.stride(1) == 1)
    L.check(L.lib().wgamd_gat_transform_heads_bf16x3(
        agg.data_ptr(), agg.stride(0), n, F_, heads, C, _gat_weight_tiles(w, heads).data_ptr(),
        None if acc_in is None else acc_in.data_ptr(), 0 if acc_in is None else acc_in.stride(0),
        None if bias is None else bias.data_ptr(), int(bool(relu)), None if out_rows is None else out_rows.data_ptr(),
        out.data_ptr(), out.stride(0), get_stream()), "wgamd_gat_transform_heads_bf16x3")
    return out


def gat_layer_fused_supported(F_: int, heads: int, C: int) -> bool:
    return bool(L.lib().wgamd_gat_layer_fused_supported(int(F_), int(heads), int(C)))


def gat_layer_fused(row_ptr, col, x, a_src, a_dst, w, heads, dst_rows=None, negative_slope=0.2, acc_in=None, bias=None, relu=False,
                    out_rows=None, out=None, src_ids=None, dst_ids=None, src_terms_by_id=False, dst_terms_by_id=False):
    """``gat_aggregate_heads`` + ``gat_transform_heads_fused`` as ONE kernel (``wgamd_gat_layer_fused_bf16x3``): the
    [n_rows, heads * F] aggregate never leaves the CU.  For hops with a fan-out of at most 10 (longer rows are correct, slow).
    ``src_ids`` / ``dst_ids`` / ``*_terms_by_id``: as in ``gat_aggregate_heads``."""
    _check_csr(row_ptr, col)
    ids_p, dids_p, by_id = _gat_ids(src_ids, dst_ids, a_src, src_terms_by_id, dst_terms_by_id)
    n_rows, F_ = row_ptr.shape[0] - 1, x.shape[1]
    C = w.shape[1] // heads
    assert x.dtype == torch.float32 and x.stride(1) == 1 and a_src.is_contiguous() and a_dst.is_contiguous() and a_src.shape[1] == heads
    if out is None:
        assert out_rows is None
        out = torch.empty((n_rows, heads * C), dtype=torch.float32, device=x.device)
    assert col.numel() > 0 and out.stride(1) == 1 and (acc_in is None or acc_in.stride(1) == 1)
    L.check(L.lib().wgamd_gat_layer_fused_ids_bf16x3(
        row_ptr.data_ptr(), col.data_ptr(), n_rows, x.data_ptr(), x.stride(0), ids_p, dids_p, by_id, F_, a_src.data_ptr(),
        a_dst.data_ptr(), heads, C,
        float(negative_slope), None if dst_rows is None else dst_rows.data_ptr(), _gat_weight_tiles(w, heads).data_ptr(),
        None if acc_in is None else acc_in.data_ptr(), 0 if acc_in is None else acc_in.stride(0),
        None if bias is None else bias.data_ptr(), int(bool(relu)), None if out_rows is None else out_rows.data_ptr(),
        out.data_ptr(), out.stride(0), get_stream()), "wgamd_gat_layer_fused_ids_bf16x3")
    return out


def gather_terms_supported(F_: int, T: int) -> bool:
    return bool(L.lib().wgamd_gather_terms_supported(int(F_), int(T)))


def gather_with_terms(table: torch.Tensor, ids: torch.Tensor, v: torch.Tensor, out: torch.Tensor = None, heads: int = 0):
    """``-> (x, terms)``: ``x = table[ids]`` and ``terms = x @ v`` (``v`` [F, T], T <= 32) in ONE pass over the gathered rows
    (``wgamd_gather_terms_f32``: the row gather feeds an exact-fp32 MFMA).  For GATConv's attention logits, whose folded
    [F, H] matrices of every relation end of a node type are concatenated into ``v``.  ``heads=4``: ``terms`` comes back as
    [T / 4, n, 4] — one contiguous [n, 4] slab per relation end — instead of [n, T]."""
    assert table.dtype == torch.float32 and table.dim() == 2 and table.stride(1) == 1 and v.dtype == torch.float32
    assert ids.dim() == 1 and ids.is_contiguous() and ids.dtype in (torch.int32, torch.int64)
    n, F_, T = int(ids.shape[0]), int(table.shape[1]), int(v.shape[1])
    assert v.shape[0] == F_ and v.is_contiguous()
    if out is None:
        out = torch.empty((n, F_), dtype=torch.float32, device=table.device)
    assert heads in (0, 4) and (heads == 0 or T % 4 == 0)
    terms = torch.empty((T // 4, n, 4) if heads else (n, T), dtype=torch.float32, device=table.device)
    L.check(L.lib().wgamd_gather_terms_f32(table.data_ptr(), table.stride(0), ids.data_ptr(), torch_dtype_to_wm(ids.dtype), n, F_,
                                           v.data_ptr(), T, out.data_ptr(), out.stride(0), terms.data_ptr(), T, heads,
                                           get_stream()), "wgamd_gather_terms_f32")
    return out, terms


def lazy_rows_terms(table: torch.Tensor, ids: torch.Tensor, v: torch.Tensor, heads: int = 0):
    """``terms = table[ids] @ v`` WITHOUT writing the gathered rows (``wgamd_gather_terms_f32`` with no row output): the attention
    logits of a ``LazyRows`` input whose rows the relation kernels then read through ``ids`` themselves."""
    assert table.dtype == torch.float32 and table.dim() == 2 and table.stride(1) == 1 and v.dtype == torch.float32 and v.is_contiguous()
    assert ids.dim() == 1 and ids.is_contiguous() and ids.dtype in (torch.int32, torch.int64)
    n, F_, T = int(ids.shape[0]), int(table.shape[1]), int(v.shape[1])
    assert v.shape[0] == F_ and heads in (0, 4) and (heads == 0 or T % 4 == 0)
    terms = torch.empty((T // 4, n, 4) if heads else (n, T), dtype=torch.float32, device=table.device)
    L.check(L.lib().wgamd_gather_terms_f32(table.data_ptr(), table.stride(0), ids.data_ptr(), torch_dtype_to_wm(ids.dtype), n, F_,
                                           v.data_ptr(), T, None, 0, terms.data_ptr(), T, heads, get_stream()), "wgamd_gather_terms_f32")
    return terms


def gather_term_slabs(slabs: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """``out[k, i] = slabs[k, ids[i]]`` for attention-term slabs [K, n_table, 4] (``wgamd_gather_term_slabs_f32``): the terms of a
    call group's rows from the terms of the table's rows."""
    assert slabs.dtype == torch.float32 and slabs.dim() == 3 and slabs.shape[2] == 4 and slabs.is_contiguous()
    assert ids.dim() == 1 and ids.is_contiguous() and ids.dtype in (torch.int32, torch.int64)
    K, n_in, n = int(slabs.shape[0]), int(slabs.shape[1]), int(ids.shape[0])
    out = torch.empty((K, n, 4), dtype=torch.float32, device=slabs.device)
    L.check(L.lib().wgamd_gather_term_slabs_f32(slabs.data_ptr(), n_in, K, ids.data_ptr(), torch_dtype_to_wm(ids.dtype), n,
                                                out.data_ptr(), get_stream()), "wgamd_gather_term_slabs_f32")
    return out


def rows_terms(x: torch.Tensor, v: torch.Tensor, heads: int = 0):
    """``terms = x @ v`` for a resident [n, F] matrix and a narrow ``v`` [F, T] (T <= 32) in ONE streaming pass over ``x``
    (``wgamd_gather_terms_f32`` without an id list and without a row copy: exact-fp32 MFMA, rows through registers once).
    ``heads=4``: [T / 4, n, 4] slabs — one contiguous [n, 4] block per relation end, what the GAT kernels read — instead of
    [n, T]: no transposing copy after a library GEMM."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and v.dtype == torch.float32 and v.is_contiguous()
    n, F_, T = int(x.shape[0]), int(x.shape[1]), int(v.shape[1])
    assert v.shape[0] == F_ and heads in (0, 4) and (heads == 0 or T % 4 == 0)
    terms = torch.empty((T // 4, n, 4) if heads else (n, T), dtype=torch.float32, device=x.device)
    L.check(L.lib().wgamd_gather_terms_f32(x.data_ptr(), x.stride(0), None, torch_dtype_to_wm(torch.int64), n, F_, v.data_ptr(), T,
                                           None, 0, terms.data_ptr(), T, heads, get_stream()), "wgamd_gather_terms_f32")
    return terms


def bias_act_rows(x, bias=None, relu=True, dst_rows=None, out=None):
    """``out[dst_rows[i]] = act(x[i] + bias)`` in one pass (wgamd_bias_act_rows_f32); ``out`` defaults to a fresh [n, C]."""
    n, C = x.shape
    if out is None:
        assert dst_rows is None
        out = torch.empty((n, C), dtype=torch.float32, device=x.device)
    L.check(L.lib().wgamd_bias_act_rows_f32(x.data_ptr(), x.stride(0), n, C, None if bias is None else bias.data_ptr(), int(bool(relu)),
                                            None if dst_rows is None else dst_rows.data_ptr(), out.data_ptr(), out.stride(0),
                                            get_stream()), "wgamd_bias_act_rows_f32")
    return out


def gat_backward_supported(H: int, C: int) -> bool:
    """Shapes ``wgamd_gat_csr_bwd_f32`` is built for (include/wgamd_ext.h)."""
    return C % 4 == 0 and ((C // 4) & (C // 4 - 1)) == 0 and H * C <= 256


def gat_backward(row_ptr, col, x, a_src, a_dst, alpha, grad_out, heads, negative_slope=0.2):
    """(grad_x, grad_a_src, grad_a_dst) of ``gat_forward`` on the HIP kernels: a destination-major pass (per-head dot
    products, softmax backward, grad_a_dst) and a source-major pass over the transposed hop CSR (grad_x, grad_a_src)."""
    _check_csr(row_ptr, col)
    n_rows, n_src, E = row_ptr.shape[0] - 1, x.shape[0], col.shape[0]
    C = x.shape[1] // heads
    g = grad_out.contiguous()
    row_ptr_t, edge_perm, edge_dst, _ = _csr_transpose(row_ptr, col, n_src, want_perm=True, want_dst=True)
    de = torch.empty((E, heads), dtype=torch.float32, device=x.device)
    gx = torch.empty_like(x)
    ga_src, ga_dst = torch.empty_like(a_src), torch.empty_like(a_dst)
    need = L.lib().wgamd_gat_csr_bwd_workspace_bytes(E, heads, C)      # pieces of long source rows (hubs of the hop)
    ws = torch.empty(need, dtype=torch.uint8, device=x.device)
    L.check(L.lib().wgamd_gat_csr_bwd_f32_v2(
        row_ptr.data_ptr(), col.data_ptr(), n_rows, x.data_ptr(), x.stride(0), a_src.data_ptr(), a_dst.data_ptr(), heads, C,
        float(negative_slope), alpha.data_ptr(), g.data_ptr(), g.stride(0), row_ptr_t.data_ptr(), edge_perm.data_ptr(),
        edge_dst.data_ptr(), n_src, de.data_ptr(), gx.data_ptr(), gx.stride(0), ga_src.data_ptr(), ga_dst.data_ptr(),
        E, ws.data_ptr(), need, get_stream()), "wgamd_gat_csr_bwd_f32_v2")
    return gx, ga_src, ga_dst


class _GatCsr(torch.autograd.Function):
    """Forward: fused HIP kernel.  Backward: edge-wise torch ops on the saved attention (the
    training-time gradient path is not on the north-star hot path)."""

    @staticmethod
    def forward(ctx, x, a_src, a_dst, row_ptr, col, heads, slope):
        x, a_src, a_dst = x.contiguous(), a_src.contiguous(), a_dst.contiguous()
        out, alpha = gat_forward(row_ptr, col, x, a_src, a_dst, heads, slope, need_alpha=True)
        ctx.save_for_backward(x, a_src, a_dst, row_ptr, col, alpha)
        ctx.heads, ctx.slope = heads, slope
        return out

    @staticmethod
    def backward(ctx, g):
        x, a_src, a_dst, row_ptr, col, alpha = ctx.saved_tensors
        H = ctx.heads
        C = x.shape[1] // H
        n_rows = row_ptr.shape[0] - 1
        if gat_backward_supported(H, C) and col.shape[0] > 0:
            return gat_backward(row_ptr, col, x, a_src, a_dst, alpha, g, H, ctx.slope) + (None, None, None, None)
        deg = (row_ptr[1:] - row_ptr[:-1]).long()
        dst = torch.repeat_interleave(torch.arange(n_rows, device=x.device), deg)
        src = col.long()
        g3 = g.reshape(n_rows, H, C)[dst]                      # [E,H,C]
        x3 = x.reshape(-1, H, C)[src]                          # [E,H,C]
        gx = torch.zeros_like(x).reshape(-1, H, C).index_add_(0, src, alpha.unsqueeze(-1) * g3)
        dalpha = (g3 * x3).sum(-1)                             # [E,H]
        dot = torch.zeros((n_rows, H), device=x.device).index_add_(0, dst, alpha * dalpha)
        ds = alpha * (dalpha - dot[dst])
        s = a_src[src] + a_dst[dst]
        ds = torch.where(s > 0, ds, ds * ctx.slope)
        ga_src = torch.zeros_like(a_src).index_add_(0, src, ds)
        ga_dst = torch.zeros_like(a_dst).index_add_(0, dst, ds)
        return gx.reshape(x.shape), ga_src, ga_dst, None, None, None, None


def _to_csr(edge_index, n_dst):
    """edge_index [2,E] (row 0 = source j, row 1 = destination i; PyG) -> destination-major CSR (edge order kept inside a
    destination).  int64 ids on the device go through ``wgamd_coo_to_csr_i64`` (one radix sort over the bits a destination
    id needs); anything else through the torch formulation of the same."""
    ready = getattr(edge_index, "_wgamd_csr", None)      # (version, n_dst, row_ptr, col): made by the loader for its call group
    if ready is not None and ready[0] == edge_index._version and ready[1] == n_dst:
        return ready[2], ready[3]
    src, dst = edge_index[0], edge_index[1]
    # (the flag is the tensor's version counter at the time the loader vouched for the order: an in-place edit of the
    #  edge list afterwards — a permutation, self loops written into the same storage — bumps the counter and the sort runs)
    if getattr(edge_index, "_wgamd_dst_sorted", None) == edge_index._version and edge_index.is_cuda:
        # the loaders' own edge lists are destination-major already (hop after hop, a hop's edges in the CSR order of its
        # frontier, every hop's destinations after the previous hop's): the CSR is a search for the run boundaries and a cast
        # — no sort (0.77 ms per 88 k-edge mini-batch through the radix sort below, most of it launch latency)
        row_ptr = torch.searchsorted(dst.contiguous(), torch.arange(n_dst + 1, device=dst.device, dtype=dst.dtype)).to(torch.int32)
        return row_ptr, src.to(torch.int32).contiguous()
    if edge_index.dtype == torch.int64 and edge_index.is_cuda:
        src, dst = src.contiguous(), dst.contiguous()
        E, dev = dst.shape[0], dst.device
        row_ptr = torch.empty(n_dst + 1, dtype=torch.int32, device=dev)
        col = torch.empty(E, dtype=torch.int32, device=dev)
        need = L.lib().wgamd_coo_to_csr_workspace_bytes(E, n_dst)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        L.check(L.lib().wgamd_coo_to_csr_i64(src.data_ptr(), dst.data_ptr(), E, n_dst, row_ptr.data_ptr(), col.data_ptr(), None,
                                             ws.data_ptr(), need, get_stream()), "wgamd_coo_to_csr_i64")
        return row_ptr, col
    order = torch.sort(dst, stable=True).indices
    col = src[order].to(torch.int32).contiguous()
    counts = torch.bincount(dst, minlength=n_dst)
    row_ptr = torch.zeros(n_dst + 1, dtype=torch.int32, device=dst.device)
    row_ptr[1:] = torch.cumsum(counts, 0)
    return row_ptr, col


_ARANGE = {}


_HEADS_WGRAD_MIN_ROWS = int(os.environ.get("WGAMD_HEADS_WGRAD_MIN_ROWS", 16384))


class _HeadsTransform(torch.autograd.Function):
    """``y[n, h, :] = agg[n, h, :] @ w3[:, h, :]`` — the dense tail of the aggregate-first GAT layer (per-head weights on the
    destination rows).  Forward and the gradient of ``agg`` are library products of ordinary shapes; the WEIGHT gradient
    ``agg_h^T dY_h`` is a [F, n] x [n, C] product with n in the hundreds of thousands and a 128 x 64 result, which the library
    runs at 0.16 TB/s (2.07 ms per mag relation, 6 % of the whole mag run) — it goes to the split-K bf16x3 kernel of the SAGE
    layer's weight gradient instead (``wgamd_sage_wgrad_bf16x3``: ``dZ^T [A | B]`` with A, B = the two halves of agg_h's columns,
    so nothing is read twice), one launch + its partial-sum reduction per head."""

    @staticmethod
    def forward(ctx, agg3, w3, acc3=None):
        # one product per head, written straight into its columns of the [n, H, C] result: the batched form (einsum -> bmm)
        # leaves [H, n, C] and pays a 450 MB permute-copy per mag relation to bring it back (10 % of the mag training step).
        # ``acc3`` (the sum of the relations before this one, HeteroConv's aggregation): the products are added INTO it (beta =
        # 1) instead of a separate 3 x 450 MB addition per relation.
        ctx.save_for_backward(agg3, w3)
        n, H, F_ = agg3.shape
        C = w3.shape[2]
        if acc3 is not None:
            ctx.mark_dirty(acc3)
        if _GAT_TRANSFORM_FUSED and gat_transform_supported(F_, H, C) and agg3.is_contiguous() and (acc3 is None or acc3.is_contiguous()):
            # the inference route's kernel: all heads in ONE pass on the bf16 matrix pipe at fp32 accuracy, the running sum read and
            # written in place (the library's four [n, 128] x [128, 64] products run at 0.9 TB/s: 1.45 ms per mag relation)
            y = acc3 if acc3 is not None else torch.empty((n, H, C), dtype=torch.float32, device=agg3.device)
            gat_transform_heads_fused(agg3.view(n, H * F_), w3.reshape(F_, H * C).contiguous(), H,
                                      acc_in=None if acc3 is None else acc3.view(n, H * C), out=y.view(n, H * C))
            return y
        if acc3 is None:
            y = torch.empty((n, H, C), dtype=torch.float32, device=agg3.device)
            for h in range(H):
                torch.mm(agg3[:, h, :], w3[:, h, :], out=y[:, h, :])
            return y
        for h in range(H):
            acc3[:, h, :].addmm_(agg3[:, h, :], w3[:, h, :])
        return acc3

    @staticmethod
    def backward(ctx, g):
        agg3, w3 = ctx.saved_tensors
        g = g.contiguous()
        d_agg = None
        if ctx.needs_input_grad[0]:
            d_agg = torch.empty_like(agg3)
            for h in range(agg3.shape[1]):
                torch.mm(g[:, h, :], w3[:, h, :].t(), out=d_agg[:, h, :])
        d_w3 = None
        if ctx.needs_input_grad[1]:
            n, H, F_ = agg3.shape
            C = g.shape[2]
            split = F_ % 8 == 0                  # the halves of a row must start 16-byte aligned
            Fk = F_ // 2 if split else F_
            rows = _arange(n, g.device)
            buf = torch.empty((H, 2, C, Fk), dtype=torch.float32, device=g.device)
            for h in range(H):
                a_h = agg3[:, h, :]
                sage_wgrad(a_h[:, :Fk], a_h[:, Fk:] if split else a_h, rows, g[:, h, :], buf[h, 0], buf[h, 1])
            # buf[h, s, c, k] = d w3[s Fk + k, h, c]
            d_w3 = (buf.permute(1, 3, 0, 2).reshape(F_, H, C) if split else buf[:, 0].permute(2, 0, 1)).contiguous()
        return d_agg, d_w3, (g if len(ctx.needs_input_grad) > 2 and ctx.needs_input_grad[2] else None)


def _heads_transform(agg3: torch.Tensor, w3: torch.Tensor, acc3: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``[acc3 +] agg3 @ w3`` per head; ``acc3`` ([n, H, C], the sum so far) may be updated in place and returned."""
    n, H, F_ = agg3.shape
    C = w3.shape[2]
    if (agg3.is_cuda and agg3.dtype == torch.float32 and agg3.is_contiguous() and n >= _HEADS_WGRAD_MIN_ROWS and F_ % 4 == 0
            and (H * F_) % 4 == 0 and (C % 4 == 0 or H == 1) and torch.is_grad_enabled() and w3.requires_grad
            and L.lib().wgamd_sage_wgrad_workspace_bytes(1, F_ // 2 if F_ % 8 == 0 else F_, C) > 0):
        if acc3 is not None and acc3.is_contiguous() and acc3.requires_grad and not acc3.is_leaf:
            return _HeadsTransform.apply(agg3, w3, acc3)
        y = _HeadsTransform.apply(agg3, w3)
        return y if acc3 is None else acc3 + y
    y = torch.einsum("nhf,fhc->nhc", agg3, w3)
    return y if acc3 is None else acc3 + y


def _arange(n: int, device) -> torch.Tensor:
    """``arange(n)`` int64 on ``device`` as a view of one grow-only buffer (the "self rows" of a mini-batch graph whose
    destinations are the first rows of x)."""
    buf = _ARANGE.get(device)
    if buf is None or buf.shape[0] < n:
        buf = torch.arange(max(n * 2, 1 << 16), dtype=torch.int64, device=device)
        _ARANGE[device] = buf
    return buf[:n]


def _split_graph(graph, n_dst):
    if isinstance(graph, (tuple, list)) and len(graph) == 2 and graph[0].dim() == 1:
        return graph[0], graph[1]          # [csr_row_ptr, csr_col_ind] as the sampler emits them
    return _to_csr(graph, n_dst)           # COO edge_index


class LazyRows:
    """``table[ids]`` that has not been gathered: what ``batch.x`` of a loader call group is when the feature table lives
    whole on this device (single GPU, or replicated).  ``nn.SAGEConv`` consumes it as it is — its first-layer kernel reads
    the table through ``ids`` (``src_ids`` of ``wgamd_sage_layer_fused_*``), so the ``[n, F]`` copy of the rows, the
    largest tensor of a mini-batch, is never written to HBM nor read back.  Anything else calls ``materialize()`` (one
    ``wholememory_gather``) or just uses it as a tensor: ``torch`` functions receive the gathered rows."""

    def __init__(self, table: torch.Tensor, ids: torch.Tensor):
        assert table.dim() == 2 and ids.dim() == 1 and ids.dtype in (torch.int32, torch.int64)
        self.table, self.ids, self._rows = table, ids.contiguous(), None

    @property
    def shape(self):
        return torch.Size((self.ids.shape[0], self.table.shape[1]))

    @property
    def dtype(self):
        return self.table.dtype

    @property
    def device(self):
        return self.table.device

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return 2

    def materialize(self) -> torch.Tensor:
        if self._rows is None and getattr(self, "_gather", None) is not None:
            self._rows = self._gather()           # (a peer-mapped table: wholememory_gather over the mapping)
        if self._rows is None:
            from .tensor import local_gather
            self._rows = local_gather(self.table, self.ids, torch.empty(tuple(self.shape), dtype=self.table.dtype,
                                                                        device=self.table.device))
        return self._rows

    def __getitem__(self, index):
        return self.materialize()[index]

    def __len__(self):
        return int(self.ids.shape[0])

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        conv = lambda v: v.materialize() if isinstance(v, LazyRows) else v   # noqa: E731
        return func(*[conv(a) for a in args], **{k: conv(v) for k, v in (kwargs or {}).items()})


class MappedTable:
    """The address space of a PEER-MAPPED feature table (CHUNKED / CONTINUOUS handle whose partitions live on several GPUs of a
    node, each mapped into this process) as the layer kernels take it: a base pointer, the row width, and rows given as BYTE
    offsets from the base (``wgamd_mapped_row_offsets``).  Quacks like the [rows, F] float32 tensor the wrappers expect; it is
    never indexed from Python."""
    byte_offset_ids = True
    dtype, is_cuda, requires_grad = torch.float32, True, False

    def __init__(self, base_ptr: int, width: int, device):
        self._ptr, self.shape, self.device = int(base_ptr), torch.Size((0, int(width))), device

    def data_ptr(self):
        return self._ptr

    def dim(self):
        return 2

    def stride(self, d=None):
        st = (int(self.shape[1]), 1)
        return st if d is None else st[d]


def mapped_lazy_rows(wm_tensor, ids: torch.Tensor) -> LazyRows:
    """``table[ids]`` of a peer-mapped ``DistributedWholeMemoryTensor`` (float32 [rows, F]) as ``LazyRows`` the first SAGE layer
    reads through: one small launch turns the ids into byte offsets over this process's mapping of every rank's partition, the
    layer kernel then loads remote rows over xGMI itself (the reference's mapped gather addresses the partitions the same way,
    gather_scatter_func.cuh:242-505) — the gathered ``[n, F]`` copy never exists.  Anything else that touches it gathers."""
    import ctypes
    assert wm_tensor.dtype == torch.float32 and wm_tensor.dim() == 2 and ids.dim() == 1 and ids.dtype in (torch.int32, torch.int64)
    ids = ids.contiguous()
    offs = torch.empty(ids.shape[0], dtype=torch.int64, device=ids.device)
    base = ctypes.c_void_p()
    L.check(L.lib().wgamd_mapped_row_offsets(wm_tensor.c, ids.data_ptr(), torch_dtype_to_wm(ids.dtype), int(ids.shape[0]),
                                             offs.data_ptr(), ctypes.byref(base), get_stream()), "wgamd_mapped_row_offsets")
    lazy = LazyRows(MappedTable(base.value, wm_tensor.shape[1], ids.device), offs)
    lazy._gather = lambda: wm_tensor.gather(ids)
    return lazy


class HopGraph:
    """One sampled hop as a layer consumes it: CSR over the hop's destination rows (``row_ptr`` int32 [n + 1]), ``col`` int32
    = row of every edge's source IN THE LAYER'S INPUT, ``self_rows`` int64 [n] = input row of every destination itself."""

    def __init__(self, row_ptr, col, self_rows):
        self.row_ptr, self.col, self.self_rows = row_ptr, col, self_rows
        self._t = None
        self.inv_deg = None     # float32 [n] = 1 / max(degree, 1) when whoever built the hop has it (loader.StagedBatch)

    @property
    def n_rows(self):
        return int(self.row_ptr.shape[0]) - 1

    def with_self_loops(self):
        """``(row_ptr, col)`` of the hop with every destination's own input row in front of its sampled neighbours — what
        ``csr_add_self_loop`` (graph_op.h:44-48) does for a square CSR, here with ``self_rows`` as the diagonal; made once."""
        if getattr(self, "_loops", None) is None or self._loops_key != (_capture_epoch if _capturing() else 0):
            self._loops_key = _capture_epoch if _capturing() else 0
            from . import graph_ops
            n = self.n_rows
            rp, col = graph_ops.add_csr_self_loop(self.row_ptr, self.col)     # row i = [i] ++ row i (csr_add_self_loop) ...
            col[rp[:n].long()] = self.self_rows.to(torch.int32)              # ... with the destination's own input row as i
            self._loops = (rp, col)
        return self._loops

    def transposed(self, n_src: int, need_self: bool = True):
        """The hop seen from its ``n_src`` input rows, for the backward pass — computed once per hop and kept, whichever layers
        and however many backward calls use it: ``(row_ptr_t, col_t, self_t)`` with the source-major CSR of
        ``wgamd_csr_transpose_i32`` (entries = destination rows, hop order inside a source: deterministic sums) and
        ``self_t[j]`` = ``n_rows + i`` where input row j is destination i itself (``self_rows`` is injective: every
        destination is a different vertex of its mini-batch), ``2 n_rows`` otherwise — the row indices ``_sage_dx`` reads its
        stacked gradient through."""
        # (under HIP-graph capture the hop's arrays are fixed buffers REFILLED before every replay: the transpose must be part
        #  of the graph, once per capture)
        key = (n_src, _capture_epoch if _capturing() else 0)
        if self._t is None or self._t[0] != key:
            n, dev = self.n_rows, self.row_ptr.device
            if self.col.shape[0] > 0:
                row_ptr_t, _, _, col_t = _csr_transpose(self.row_ptr, self.col, n_src, want_col_t=True)
            else:
                row_ptr_t, col_t = torch.zeros(n_src + 1, dtype=torch.int32, device=dev), self.col
            self._t = [key, row_ptr_t, col_t, None]
        if need_self and self._t[3] is None:     # (only the layer kernel over the transposed hop reads it: three launches)
            n, dev = self.n_rows, self.row_ptr.device
            self_t = torch.full((n_src,), 2 * n, dtype=torch.int64, device=dev)
            self_t[self.self_rows] = torch.arange(n, 2 * n, dtype=torch.int64, device=dev)
            self._t[3] = self_t
        return tuple(self._t[1:])


class LayerGraph:
    """The hops ONE layer of a trimmed GNN runs over (``cugraph_pyg_amd.loader.CallGroup.layer_graph``): the layer's output
    is the hops' destination lists back to back — hop h's rows start at ``sum(n_rows of the hops before it)``."""

    def __init__(self, hops):
        self.hops = list(hops)

    @property
    def n_rows(self):
        return sum(h.n_rows for h in self.hops)


_SAGE_DX_SMALL_ROWS = int(os.environ.get("WGAMD_SAGE_DX_SMALL_ROWS", 16384))


def _sage_dx(hop: HopGraph, gz: torch.Tensor, w_l: torch.Tensor, w_r: torch.Tensor, w_bwd, mean: bool, n_src: int):
    """Gradient of one hop of the SAGE layer w.r.t. its input rows:
    ``dX[j] = sum_{edges (i, j)} dZ[i] W_l / (deg_i if mean) + [j == self(i)] dZ[i] W_r``.
    This is the one-kernel layer itself run over the TRANSPOSED hop — sum aggregation of the destination gradients an input
    row feeds, "self" = its own destination's gradient, weight ``[W_l ; W_r]`` — so it runs on the same kernel
    (``wgamd_sage_layer_fused_*``): the stacked operand ``[dZ / deg ; dZ ; 0]`` is the only tensor built for it.  Shapes
    that kernel does not take go through the segmented transposed SpMM + library GEMMs."""
    n, N = gz.shape
    F_ = w_l.shape[1]
    Nq = (N + 3) // 4 * 4
    # A SMALL hop (one mini-batch: PerBatchStep) goes through the dense products + the segmented transposed SpMM instead: in the
    # transposed hop a popular source is a row of hundreds of entries, which one lane group of the layer kernel walks as a
    # chain of dependent loads — hidden inside a call group's millisecond launch, 250 us of a mini-batch's step when exposed
    # (profiles/r06); the segmented SpMM cuts long rows into pieces.
    if w_bwd is not None and sage_layer_fused_supported(Nq, F_) and n > _SAGE_DX_SMALL_ROWS:
        row_ptr_t, col_t, self_t = hop.transposed(n_src)
        xs = torch.zeros((2 * n + 1, Nq), dtype=torch.float32, device=gz.device)
        if mean:
            deg = (hop.row_ptr[1:] - hop.row_ptr[:-1]).clamp_(min=1).unsqueeze(1)
            torch.div(gz, deg, out=xs[:n, :N])
        else:
            xs[:n, :N] = gz
        xs[n:2 * n, :N] = gz
        return sage_layer_fused_forward(row_ptr_t, col_t, xs, self_t, w_bwd, None, relu=False, mean=False)
    if n <= _SAGE_DX_SMALL_ROWS and n > 0 and hop.col.shape[0] > 0 and _BWD_SEGMENTS:
        # the hop's own (kept) transpose + the segmented SpMM; self_rows is injective, so the W_r term is a plain indexed add
        row_ptr_t, col_t, _ = hop.transposed(n_src, need_self=False)
        g = gz
        if mean:
            g = gz * hop.inv_deg.unsqueeze(1) if hop.inv_deg is not None \
                else gz / (hop.row_ptr[1:] - hop.row_ptr[:-1]).clamp_(min=1).unsqueeze(1)
        gl = g @ w_l
        E = col_t.shape[0]
        gx = torch.empty((n_src, F_), dtype=torch.float32, device=gz.device)
        need = L.lib().wgamd_spmm_csr_segmented_workspace_bytes(E, F_)
        ws = torch.empty(need, dtype=torch.uint8, device=gz.device)
        L.check(L.lib().wgamd_spmm_csr_segmented_f32(row_ptr_t.data_ptr(), col_t.data_ptr(), n_src, E, gl.data_ptr(), gl.stride(0), F_,
                                                     gx.data_ptr(), gx.stride(0), ws.data_ptr(), need, get_stream()),
                "wgamd_spmm_csr_segmented_f32")
        return gx.index_add_(0, hop.self_rows, gz @ w_r)
    gx = spmm_csr_backward(hop.row_ptr, hop.col, gz @ w_l, n_src, mean)
    return gx.index_add_(0, hop.self_rows, gz @ w_r)


def _sage_layer_launch(ctx, src, w_l, w_r, bias, conv, graph, ids, relu, mean):
    """The layer's launches (one per hop of ``graph``); ``ctx`` = the autograd context of ``_SageLayer`` (then the aggregate is
    kept wherever a gradient is needed) or None."""
    N, F_ = w_l.shape
    Np = _padded_width(N)
    keep = ctx is not None and any(ctx.needs_input_grad[:4])
    buf = torch.empty((graph.n_rows, Np), dtype=torch.float32, device=src.device)
    # under HIP-graph capture (a per-mini-batch training step: the weights change on every replay) the kernel's operand is made
    # from the parameters by ONE launch; otherwise the cached transposed / padded / split forms
    prepared = None
    if _capturing() and sage_layer_fused_supported(F_, Np) and _pick_precision(F_, Np, None) == "bf16x3" \
            and w_l.stride(1) == 1 and w_r.stride(1) == 1:
        # (a mini-batch's few thousand rows at a width of half tiles: whole 32-row tiles — sage_layer_small_launch)
        prepared = sage_layer_planes(w_l, w_r, bias, Np, full_tiles=all(sage_layer_small_launch(F_, h.n_rows) for h in graph.hops))
    w_t, aggs, at = (None if prepared is not None else conv._weight_t()), [], 0
    for h in graph.hops:
        n = h.n_rows
        agg = torch.empty((n, F_), dtype=torch.float32, device=src.device) if keep and n > 0 else None
        if n > 0:
            sage_layer_fused_forward(h.row_ptr, h.col, src, h.self_rows, w_t, bias, relu=relu, mean=mean, src_ids=ids,
                                     out=buf[at:at + n, :N], agg_out=agg, prepared=prepared)
        aggs.append(agg)
        at += n
    out = buf[:, :N]
    if keep:
        if ctx.needs_input_grad[0] and ids is not None:
            raise NotImplementedError("gradient w.r.t. a feature table read through ids (LazyRows): trainable node "
                                      "embeddings go through wholegraph_amd.embedding")
        if isinstance(src, torch.Tensor):
            ctx.save_for_backward(src, w_l, w_r, out)
        else:                                     # (a MappedTable: an address space, not a tensor)
            ctx.save_for_backward(w_l, w_r, out)
            ctx.src_obj = src
        ctx.conv, ctx.graph, ctx.ids, ctx.relu, ctx.mean, ctx.aggs, ctx.has_bias = conv, graph, ids, relu, mean, aggs, bias is not None
    return out


class _SageLayer(torch.autograd.Function):
    """The one-kernel SAGE layer over a ``LayerGraph`` with its backward pass on the HIP kernels.  Forward: the inference
    launch (``wgamd_sage_layer_fused_bf16x3``), which under autograd also keeps the aggregate half of its operand (same
    bits in the output).  Backward: ``sage_wgrad`` per hop (ReLU mask folded in when the input needs no gradient) and, when
    the input rows need a gradient (every layer but the one that reads the features), ``_sage_dx`` over the hop's
    transpose — computed once per hop (``HopGraph.transposed``)."""

    @staticmethod
    def forward(ctx, src, w_l, w_r, bias, conv, graph, ids, relu, mean):
        return _sage_layer_launch(ctx, src, w_l, w_r, bias, conv, graph, ids, relu, mean)

    @staticmethod
    def backward(ctx, g):
        if ctx.aggs is None:
            # the kept aggregates (the largest tensors of the layer) are released by the first backward pass whatever
            # retain_graph says — they are not autograd-saved tensors, so autograd's own message would not appear
            raise RuntimeError("wholegraph_amd.nn.SAGEConv: backward through this layer a second time — its kept aggregates were "
                               "released by the first backward pass (retain_graph=True is not supported by the one-kernel "
                               "layer; run the forward again, or sum the losses before calling backward)")
        src, w_l, w_r, out = ctx.saved_tensors if len(ctx.saved_tensors) == 4 else (ctx.src_obj,) + tuple(ctx.saved_tensors)
        graph, ids, relu, mean = ctx.graph, ctx.ids, ctx.relu, ctx.mean
        N, F_ = w_l.shape
        need_x = ctx.needs_input_grad[0]
        if g.stride(1) != 1 or g.dtype != torch.float32:
            g = g.contiguous().float()
        act = out if relu else None
        if relu and need_x:
            g, act = torch.ops.aten.threshold_backward(g, out, 0), None      # dZ once, read by both gradients
        gwl, gwr = torch.empty_like(w_l, memory_format=torch.contiguous_format), torch.empty_like(w_r, memory_format=torch.contiguous_format)
        gb = torch.empty(N, dtype=torch.float32, device=g.device) if ctx.has_bias else None
        w_bwd = ctx.conv._weight_bwd() if need_x and any(h.n_rows > _SAGE_DX_SMALL_ROWS for h in graph.hops) else None
        gx, at, first = None, 0, True
        for h, agg in zip(graph.hops, ctx.aggs):
            n = h.n_rows
            if n > 0:
                sage_wgrad(agg, src, h.self_rows, g[at:at + n], gwl, gwr, gb, None if act is None else act[at:at + n],
                           src_ids=ids, accumulate=not first)
                first = False
                if need_x:
                    gh = _sage_dx(h, g[at:at + n], w_l, w_r, w_bwd, mean, src.shape[0])
                    gx = gh if gx is None else gx.add_(gh)
            at += n
        if first:
            gwl.zero_(), gwr.zero_()
            if gb is not None:
                gb.zero_()
        if need_x and gx is None:
            gx = torch.zeros_like(src)
        ctx.aggs = None
        return gx, gwl, gwr, gb, None, None, None, None, None


class SAGEConv(torch.nn.Module):
    """``out = lin_l(mean_{j in N(i)} x_j) + lin_r(x_i)`` (PyG ``SAGEConv``, aggr mean|sum).

    ``forward(x, graph)``: ``graph`` = the hop's ``[csr_row_ptr, csr_col_ind]`` or a COO ``edge_index`` (the reference's call
    shapes, gnn_model.py:178-199), or a ``LayerGraph`` of a loader call group — then the whole layer (feature fetch when
    ``x`` is a ``LazyRows``, aggregation, both linear maps, bias and the optional ``act="relu"``) is ONE kernel per hop
    (``wgamd_sage_layer_fused_*``) where the shape allows it, aggregation kernel + library GEMM otherwise."""

    def __init__(self, in_channels: Union[int, Tuple[int, int]], out_channels: int, aggr: str = "mean",
                 root_weight: bool = True, bias: bool = True):
        super().__init__()
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        self.in_channels, self.out_channels, self.aggr, self.root_weight = in_channels, out_channels, aggr, root_weight
        self.lin_l = torch.nn.Linear(in_channels[0], out_channels, bias=bias)
        self.lin_r = torch.nn.Linear(in_channels[1], out_channels, bias=False) if root_weight else None
        self._w_t = self._w_bwd = None

    def _weight_bwd(self):
        """``[W_l ; W_r]`` ([2 Nq, F], Nq = N rounded up to 4, zero rows between) — the weight of the layer kernel when it runs
        the input gradient over the transposed hop (``_sage_dx``); None when that kernel does not take the shape."""
        wl, wr = self.lin_l.weight, self.lin_r.weight
        N, F_ = wl.shape
        Nq = (N + 3) // 4 * 4
        if not sage_layer_fused_supported(Nq, F_):
            return None
        key = (wl._version, wl.data_ptr(), wr._version, wr.data_ptr(), _weights_gen)
        if _capturing() or self._w_bwd is None or self._w_bwd[0] != key:
            w = torch.zeros((2 * Nq, F_), dtype=torch.float32, device=wl.device)
            w[:N], w[Nq:Nq + N] = wl.detach(), wr.detach()
            if _capturing():
                return w
            self._w_bwd = (key, w)
        return self._w_bwd[1]

    def _weight_t(self):
        """``cat([W_l, W_r], 1).t()`` ([2F, N]) for the one-kernel layer, rebuilt when a weight changed."""
        wl, wr = self.lin_l.weight, self.lin_r.weight
        key = (wl._version, wl.data_ptr(), wr._version, wr.data_ptr(), _weights_gen)
        if _capturing():
            return torch.cat([wl.detach(), wr.detach()], dim=1).t().contiguous()
        if self._w_t is None or self._w_t[0] != key:
            self._w_t = (key, torch.cat([wl.detach(), wr.detach()], dim=1).t().contiguous())
        return self._w_t[1]

    def _forward_layer(self, x, graph: LayerGraph, act=None):
        lazy = isinstance(x, LazyRows)
        src = x.table if lazy else x
        F_, N = src.shape[1], self.out_channels
        relu = act == "relu"
        assert act in (None, "relu"), "act: None or 'relu'"
        one_kernel = (self.lin_r is not None and self.aggr in ("mean", "sum") and src.dtype == torch.float32 and src.is_cuda
                      and src.stride(1) == 1 and sage_layer_fused_preferred(F_, N))
        if one_kernel and not sage_layer_train_supported(F_, N):
            # (a shape only the fp32-MFMA layer kernel takes has no backward kernels: under autograd it runs as aggregation
            #  kernel + library GEMM, whose autograd pieces exist for every shape)
            one_kernel = not (torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters())
                                                           or (not lazy and x.requires_grad)))
        if one_kernel:
            args = (src, self.lin_l.weight, self.lin_r.weight, self.lin_l.bias, self, graph, x.ids if lazy else None, relu,
                    self.aggr == "mean")
            if not (torch.is_grad_enabled() and (self.lin_l.weight.requires_grad or self.lin_r.weight.requires_grad
                                                 or (not lazy and x.requires_grad))):
                return _sage_layer_launch(None, *args)       # same launches; no autograd node to build (~10 us per call)
            return _SageLayer.apply(*args)
        # aggregation kernel(s) + library GEMM: every hop's [mean | self] rows, then lin_l / lin_r as torch modules
        xd = x.materialize() if lazy else x
        outs = []
        for h in graph.hops:
            agg = spmm_csr(xd, h.row_ptr, h.col, self.aggr)
            o = self.lin_l(agg)
            if self.lin_r is not None:
                o = o + self.lin_r(xd[h.self_rows])
            outs.append(o)
        out = outs[0] if len(outs) == 1 else torch.cat(outs)
        return torch.relu_(out) if relu else out

    def forward(self, x, graph, act=None):
        if isinstance(graph, LayerGraph):
            return self._forward_layer(x, graph, act)
        if isinstance(x, LazyRows):
            x = x.materialize()
        x_src, x_dst = (x, x) if isinstance(x, torch.Tensor) else x
        row_ptr, col = _split_graph(graph, x_dst.shape[0])
        # ((x, x_target) with x_target = x[:n], the reference's call shape gnn_model.py:178-199: the same rows)
        same = x_src is x_dst or (x_dst.data_ptr() == x_src.data_ptr() and x_dst.stride() == x_src.stride())
        if (same and x_src.is_cuda and x_src.dtype == torch.float32
                and self.lin_r is not None and self.aggr in ("mean", "sum") and act in (None, "relu")
                and x_src.stride(1) == 1 and sage_layer_fused_preferred(x_src.shape[1], self.out_channels)):
            # one sampled (sub)graph whose destinations are the first rows of x: the whole layer as ONE kernel, forward and
            # backward (no library GEMM — whose shape heuristics alone cost ~1 ms for every new row count a mini-batch brings)
            n = row_ptr.shape[0] - 1
            # the hop (and with it its transpose, for the backward pass) belongs to the GRAPH, not to the layer: every layer a
            # model runs over one edge_index shares it
            keep = getattr(graph, "_wgamd_hop", None) if isinstance(graph, torch.Tensor) else None
            if keep is not None and keep[0] == graph._version and keep[1] == n and keep[2].row_ptr is row_ptr:
                lg = keep[3]
            else:
                lg = LayerGraph([HopGraph(row_ptr, col, _arange(n, x_src.device))])
                if isinstance(graph, torch.Tensor):
                    try:
                        graph._wgamd_hop = (graph._version, n, lg.hops[0], lg)
                    except AttributeError:
                        pass
            return self._forward_layer(x_src, lg, act)
        out = self.lin_l(spmm_csr(x_src, row_ptr, col, self.aggr))
        if self.lin_r is not None:
            out = out + self.lin_r(x_dst[: out.shape[0]])
        return torch.relu_(out) if act == "relu" else out


class GATConv(torch.nn.Module):
    """PyG ``GATConv`` (shared ``lin``, ``att_src``/``att_dst``, LeakyReLU 0.2, per-destination
    softmax, concat or mean over heads, optional self loops)."""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, concat: bool = True,
                 negative_slope: float = 0.2, add_self_loops: bool = True, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.negative_slope, self.add_self_loops = concat, negative_slope, add_self_loops
        self.lin = torch.nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_src = torch.nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = torch.nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = torch.nn.Parameter(torch.zeros(heads * out_channels if concat else out_channels)) if bias else None
        bound = math.sqrt(6.0 / (heads + out_channels))
        torch.nn.init.uniform_(self.att_src, -bound, bound)
        torch.nn.init.uniform_(self.att_dst, -bound, bound)

    def _forward_layer(self, x, lg: LayerGraph, act=None):
        """A trimmed layer of a loader call group (``CallGroup.layer_graph(j)``), AGGREGATE-FIRST: per hop one
        ``_GatAggregateHeads`` launch over the untransformed input rows (a ``LazyRows`` input is read through its node list, its
        attention terms are those of the table's rows when the table is the shorter side), self loops as an extra leading
        neighbour, then the per-head weights, bias and activation on the hop's destination rows.  Under autograd the same
        code trains (``wgamd_gat_aggregate_heads_bwd_f32``)."""
        assert act in (None, "relu")
        H, C, F_ = self.heads, self.out_channels, self.in_channels
        lazy = isinstance(x, LazyRows) and isinstance(x.table, torch.Tensor) and x.ids.dtype == torch.int64 \
            and x.table.dtype == torch.float32 and x.table.stride(1) == 1 and x.table.stride(0) % 4 == 0 \
            and x.table.data_ptr() % 16 == 0 and x._rows is None
        if lazy:
            X, ids = x.table, x.ids
            by_id = 2 * X.shape[0] <= len(x)
            rows = X if by_id else x.materialize()
        else:
            X = x.materialize() if isinstance(x, LazyRows) else x
            ids, by_id, rows = None, False, X
        w3 = self.lin.weight.t().reshape(F_, H, C)
        folds = torch.cat([(w3 * self.att_src.view(1, H, C)).sum(-1), (w3 * self.att_dst.view(1, H, C)).sum(-1)], 1)
        terms = _NarrowTerms.apply(rows, folds)
        a_src, a_dst = terms[:, :H], terms[:, H:]
        outs = []
        for hop in lg.hops:
            n = hop.n_rows
            if n == 0:
                continue
            rp, col = hop.with_self_loops() if self.add_self_loops else (hop.row_ptr, hop.col)
            if col.shape[0] == 0:         # (no self loops and nothing sampled: the aggregate of an empty neighbourhood)
                outs.append(torch.zeros((n, H, C), dtype=torch.float32, device=X.device))
                continue
            agg = _GatAggregateHeads.apply(X, a_src, a_dst, rp, col, H, hop.self_rows, ids, ids if by_id else None, by_id, by_id,
                                           self.negative_slope)
            outs.append(_heads_transform(agg.view(n, H, F_), w3))
        out = torch.cat(outs) if outs else torch.zeros((0, H, C), dtype=torch.float32, device=X.device)
        out = out.reshape(out.shape[0], H * C) if self.concat else out.mean(1)
        if self.bias is not None:
            out = out + self.bias
        return torch.relu(out) if act == "relu" else out

    def forward(self, x, graph, act=None):
        if _capturing():
            # (the derived forms of this layer's parameters — folded attention vectors, weight tiles, pooled buffers — are cached
            #  in Python against the parameters' versions: a replayed graph would keep using the ones of the capture)
            raise RuntimeError("wholegraph_amd.nn.%s is not supported under HIP-graph capture (loader.PerBatchStep): its "
                               "derived-weight caches are not capture-safe; SAGEConv layers are" % type(self).__name__)
        from . import graph_ops
        if isinstance(graph, LayerGraph):
            if self.in_channels % 4 == 0 and self.in_channels <= 256 and self.heads in (1, 2, 4, 8):
                return self._forward_layer(x, graph, act)
            raise NotImplementedError("GATConv over a call group's LayerGraph: in_channels % 4 == 0, <= 256; heads 1, 2, 4 or 8")
        assert act is None, "act: only with a call group's LayerGraph"
        x_src, x_dst = (x, x) if isinstance(x, torch.Tensor) else x
        H, C = self.heads, self.out_channels
        n_dst = x_dst.shape[0]
        row_ptr, col = _split_graph(graph, n_dst)
        if self.add_self_loops:
            # destinations are the first n_dst sources (sampler layout), so "self" = own row index
            row_ptr, col = graph_ops.add_csr_self_loop(row_ptr, col)
        h_src = self.lin(x_src)
        h_dst = h_src[:n_dst] if x_dst is x_src or x_dst.data_ptr() == x_src.data_ptr() else self.lin(x_dst)
        a_src = (h_src.view(-1, H, C) * self.att_src).sum(-1)
        a_dst = (h_dst.view(-1, H, C) * self.att_dst).sum(-1)
        out = _GatCsr.apply(h_src, a_src, a_dst, row_ptr, col, H, self.negative_slope)
        if not self.concat:
            out = out.view(-1, H, C).mean(1)
        if self.bias is not None:
            out = out + self.bias
        return out


# ---------------------------------------------------------------------------------------------------------------------
# heterogeneous layers over a call group (BASELINE configs[4]: ogbn-mag-like 2-hop walk + HeteroConv{GATConv})
# ---------------------------------------------------------------------------------------------------------------------
_stage_hook = None


def set_stage_hook(hook):
    """``hook(name, fn) -> fn()`` wraps every device stage of the call-group layers (bench_mag.py times them with HIP events);
    ``None`` = plain calls."""
    global _stage_hook
    _stage_hook = hook


def _stage(name, fn):
    return fn() if _stage_hook is None else _stage_hook(name, fn)


class RelationHop:
    """One (hop, edge type) of a heterogeneous call group as a layer consumes it: CSR over the hop's frontier entries of the
    destination type (``row_ptr`` int32 [n + 1]); ``col`` int32 = row of every edge's source in the layer's INPUT of the source
    type; ``dst_rows`` int64 [n] = row of every frontier entry in the layer's input of the destination type (its attention
    term); ``out_rows`` int64 [n] = its row in the layer's OUTPUT of the destination type (None: entry j is output row j)."""

    def __init__(self, edge_type, hop, row_ptr, col, dst_rows, out_rows, n_edges, fanout):
        self.edge_type, self.hop, self.row_ptr, self.col = edge_type, hop, row_ptr, col
        self.dst_rows, self.out_rows, self.n_edges, self.fanout = dst_rows, out_rows, int(n_edges), int(fanout)

    @property
    def n_rows(self):
        return int(self.row_ptr.shape[0]) - 1


class HeteroLayerGraph:
    """What ONE layer of a trimmed heterogeneous GNN runs over (``cugraph_pyg_amd.loader.HeteroCallGroup.layer_graph``): the
    relation hops, and per node type the number of output rows (every output row of a type is a frontier entry of exactly one
    hop of that type)."""

    def __init__(self, relations, n_out, node_types):
        self.relations, self.n_out, self.node_types = list(relations), dict(n_out), list(node_types)

    @property
    def num_edges(self):
        return sum(r.n_edges for r in self.relations)


from .pool import GrowOnlyPool  # noqa: E402

_agg_pool = GrowOnlyPool()


class _NarrowTerms(torch.autograd.Function):
    """``terms = x @ v`` for a LONG ``x`` ([n, F], a feature table or a hidden state) and a narrow ``v`` ([F, T], T <= 32: the
    folded attention vectors of a node type).  Forward: ``rows_terms`` (one streaming pass, exact-fp32 MFMA) where the shape
    allows, a library product otherwise; backward: ``dv = x^T @ dterms`` by ``wgamd_rows_terms_bwd_f32`` (x streamed once; a
    library GEMM sees a 128 x 12 output and a reduction over a million rows), ``dx = dterms @ v^T`` only where x asks for it."""

    @staticmethod
    def forward(ctx, x, v):
        v = v.contiguous()
        F_, T = int(v.shape[0]), int(v.shape[1])
        fast = x.dtype == torch.float32 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 \
            and gather_terms_supported(F_, T)
        terms = rows_terms(x, v) if fast and x.shape[0] > 0 else x @ v
        ctx.save_for_backward(x, v)
        return terms

    @staticmethod
    def backward(ctx, g):
        x, v = ctx.saved_tensors
        g = g.contiguous()
        F_, T = int(v.shape[0]), int(v.shape[1])
        dv = None
        if ctx.needs_input_grad[1]:
            if x.stride(1) == 1 and F_ <= 256 and T <= 32 and x.shape[0] > 0:
                dv = torch.zeros((F_, T), dtype=torch.float32, device=g.device)
                L.check(L.lib().wgamd_rows_terms_bwd_f32(x.data_ptr(), x.stride(0), int(x.shape[0]), F_, g.data_ptr(), T, dv.data_ptr(),
                                                         get_stream()), "wgamd_rows_terms_bwd_f32")
            else:
                dv = x.t() @ g
        dx = g @ v.t() if ctx.needs_input_grad[0] else None
        return dx, dv


class _GatAggregateHeads(torch.autograd.Function):
    """``gat_aggregate_heads`` under autograd: the aggregate-first GAT aggregation of a sampled hop with gradients for the
    attention terms (at the rows the forward read them from: per listed row, or per TABLE row) and, when ``x`` requires it, for
    the source rows (``wgamd_gat_aggregate_heads_bwd_f32``; float atomics: reproducible to fp32 rounding).  The dense tail —
    per-head weights, the sum over relations, bias — stays in torch on the few destination rows."""

    @staticmethod
    def forward(ctx, x, a_src, a_dst, row_ptr, col, heads, dst_rows, src_ids, dst_ids, src_by_id, dst_by_id, slope):
        a_src, a_dst = a_src.contiguous(), a_dst.contiguous()
        # (the [rows, heads x F] aggregate is gigabytes per hop of a call group and changes size by a per cent from group to
        #  group: a grow-only buffer instead of a hipMalloc / hipFree pair per hop — recycled once nothing holds it any more)
        out = _agg_pool.take((int(row_ptr.shape[0]) - 1, heads * int(x.shape[1])), torch.float32, x.device)
        agg = gat_aggregate_heads(row_ptr, col, x, a_src, a_dst, heads, dst_rows=dst_rows, negative_slope=slope, out=out, src_ids=src_ids,
                                  dst_ids=dst_ids, src_terms_by_id=src_by_id, dst_terms_by_id=dst_by_id)
        ctx.save_for_backward(x, a_src, a_dst, row_ptr, col)
        ctx.extra = (heads, dst_rows, src_ids, dst_ids, src_by_id, dst_by_id, slope)
        return agg

    @staticmethod
    def backward(ctx, g):
        x, a_src, a_dst, row_ptr, col = ctx.saved_tensors
        heads, dst_rows, src_ids, dst_ids, src_by_id, dst_by_id, slope = ctx.extra
        g = g.contiguous()
        n_rows, F_ = row_ptr.shape[0] - 1, x.shape[1]
        ga_src, ga_dst = torch.zeros_like(a_src), torch.zeros_like(a_dst)
        de = torch.empty((max(int(col.shape[0]), 1), heads), dtype=torch.float32, device=g.device)
        ids_p, dids_p, by_id = _gat_ids(src_ids, dst_ids, a_src, src_by_id, dst_by_id)
        want_gx = ctx.needs_input_grad[0]
        # the source rows' gradient: source-major over the transposed hop (every row written once) where the addressing is plain —
        # a hidden-state input —, float atomics (F per edge) otherwise
        transposed = want_gx and src_ids is None and int(col.shape[0]) > 0
        gx = torch.zeros_like(x) if (want_gx and not transposed) else None
        stats = torch.empty((2, n_rows, heads), dtype=torch.float32, device=g.device) if transposed else None
        L.check(L.lib().wgamd_gat_aggregate_heads_bwd_f32(
            row_ptr.data_ptr(), col.data_ptr(), n_rows, x.data_ptr(), x.stride(0), ids_p, dids_p, by_id, F_, a_src.data_ptr(),
            a_dst.data_ptr(), heads, float(slope), None if dst_rows is None else dst_rows.data_ptr(), g.data_ptr(), g.stride(0),
            de.data_ptr(), ga_src.data_ptr(), ga_dst.data_ptr(), None if gx is None else gx.data_ptr(),
            0 if gx is None else gx.stride(0), None if stats is None else stats.data_ptr(), get_stream()),
            "wgamd_gat_aggregate_heads_bwd_f32")
        if transposed:
            n_src = int(x.shape[0])
            hit = getattr(row_ptr, "_wgamd_gat_t", None)          # (one transpose per hop CSR, kept on the tensor the layer graph holds)
            if hit is None or hit[0] != n_src or hit[1] != col.data_ptr():
                row_ptr_t, _, _, col_t = _csr_transpose(row_ptr, col, n_src, want_col_t=True)
                hit = (n_src, col.data_ptr(), row_ptr_t, col_t)
                try:
                    row_ptr._wgamd_gat_t = hit
                except AttributeError:
                    pass
            gx = torch.empty_like(x)
            L.check(L.lib().wgamd_gat_aggregate_heads_bwd_gx_f32(
                hit[2].data_ptr(), hit[3].data_ptr(), n_src, n_rows, F_, a_src.data_ptr(), a_dst.data_ptr(), heads, float(slope),
                None if dst_rows is None else dst_rows.data_ptr(), stats.data_ptr(), g.data_ptr(), g.stride(0), gx.data_ptr(),
                gx.stride(0), get_stream()), "wgamd_gat_aggregate_heads_bwd_gx_f32")
        return gx, ga_src, ga_dst, None, None, None, None, None, None, None, None, None


class HeteroConv(torch.nn.Module):
    """``torch_geometric.nn.HeteroConv({edge_type: conv}, aggr="sum")`` for ``GATConv`` relations: the output of a node type
    is the sum over the relations ending in it (examples/mag_lp_mnmg.py:141 builds this stack; GATConv as
    pylibwholegraph/torch/gnn_model.py:45-59).

    ``forward(x_dict, graph, act=None)``
      * ``graph`` a ``HeteroLayerGraph`` of a loader call group (no autograd): every (hop, edge type) is ONE launch —
        AGGREGATE-FIRST (the attention-weighted sum is linear, so it runs over the UNTRANSFORMED source rows and the per-head
        weights are applied to the few destination rows afterwards: the ``lin`` GEMM over every source row, 10-20x more rows,
        never runs), attention logits ``x @ fold(W, att)`` made by the feature gather itself when ``x_dict[t]`` is a
        ``LazyRows`` (``wgamd_gather_terms_f32``), HeteroConv's running sum, bias, ReLU and the row placement folded into the
        last relation's launch (``wgamd_gat_layer_fused_bf16x3`` / ``wgamd_gat_transform_heads_bf16x3``).
      * ``graph`` a dict ``{edge_type: edge_index | [csr_row_ptr, csr_col_ind]}`` (a mini-batch ``HeteroData``; autograd):
        ``convs[edge_type]((x_src, x_dst), graph[edge_type])`` summed per destination type — PyG's own formulation."""

    def __init__(self, convs, aggr: str = "sum"):
        super().__init__()
        assert aggr in ("sum", "add"), "aggr: sum"
        self.edge_types = sorted(convs)
        self.convs = torch.nn.ModuleDict({"__".join(et): convs[et] for et in self.edge_types})
        self._folded = {}
        self.stage_tag = ""        # suffix of this layer's stage names under set_stage_hook ("1": gat1:..., transform1, ...)
        # the one-kernel relation keeps 10 neighbours of a row in registers and continues longer rows one neighbour at a time:
        # hops with a larger fan-out take the two-kernel path (the fan-out-25 hop of the mag workload through the one-kernel
        # relation: 0.77 ms instead of 0.39 + 0.16 per call group)
        self.fused_max_fanout = int(os.environ.get("WGAMD_GAT_FUSED_MAX_FANOUT", "10"))
        # a LazyRows input (table + node list) stays lazy: its attention terms come from one read-only pass over the listed rows
        # and every relation kernel reads the table through the list — the [n, F] copy of the rows is never written
        self.fetch_in_layer = os.environ.get("WGAMD_GAT_FETCH_IN_LAYER", "1") != "0"
        # under autograd a call-group layer is aggregate-first too (``_forward_layer_train``); 0: PyG's own relation-by-relation,
        # transform-first formulation on ``GATConv`` (``_forward_relations``: the lin GEMM over every source row)
        self.train_aggregate_first = os.environ.get("WGAMD_GAT_TRAIN_AGGREGATE_FIRST", "1") != "0"

    def conv(self, edge_type):
        return self.convs["__".join(edge_type)]

    # ---- parameters in the form the kernels read -----------------------------------------------------------------------
    def _rel(self, et):
        """(w [in, H C] contiguous, fold(w, att_src) [in, H], fold(w, att_dst) [in, H]) of a relation, rebuilt when a parameter
        changed: ``alpha_src = ((x W).view(H, C) * att).sum(-1) = x (W . att)``."""
        c = self.conv(et)
        key = tuple((p._version, p.data_ptr()) for p in (c.lin.weight, c.att_src, c.att_dst) + ((c.bias,) if c.bias is not None else ())) \
            + (_weights_gen,)
        hit = self._folded.get(et)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                w = c.lin.weight.t().contiguous()
                w3 = w.view(w.shape[0], c.heads, c.out_channels)
                hit = (key, w, (w3 * c.att_src.view(1, c.heads, c.out_channels)).sum(-1).contiguous(),
                       (w3 * c.att_dst.view(1, c.heads, c.out_channels)).sum(-1).contiguous())
            self._folded[et] = hit
            self._folded.pop(("terms", et[0]), None), self._folded.pop(("terms", et[2]), None)
            self._folded.pop(("bias", et[2]), None)
        return hit[1:]

    def _term_keys(self, t):
        keys = []
        for et in self.edge_types:
            if et[0] == t:
                keys.append(("src", et))
            if et[2] == t:
                keys.append(("dst", et))
        return keys

    def _terms_matrix(self, t):
        """[in, H x relation ends of node type t]: the folded attention vectors of every relation end reading type t."""
        mats = [self._rel(et)[1 if end == "src" else 2] for end, et in self._term_keys(t)]      # (refreshes stale folds first)
        hit = self._folded.get(("terms", t))
        if hit is None:
            hit = torch.cat(mats, 1).contiguous() if mats else None
            self._folded[("terms", t)] = hit
        return hit

    def _bias(self, dt):
        """Sum of the biases of the relations ending in ``dt`` (HeteroConv adds the relations' outputs, bias included)."""
        rels = [et for et in self.edge_types if et[2] == dt]
        for et in rels:
            self._rel(et)
        hit = self._folded.get(("bias", dt))
        if hit is None:
            bs = [self.conv(et).bias.detach() for et in rels if self.conv(et).bias is not None]
            hit = (torch.stack(bs).sum(0).contiguous() if bs else None,)
            self._folded[("bias", dt)] = hit
        return hit[0]

    # ---- call-group layer ----------------------------------------------------------------------------------------------
    def _attention_terms(self, xs, graph):
        """-> (x tensors, a_src{et}, a_dst{et}): ``x_t @ [fold(W_r, att_src) | fold(W_r, att_dst) ...]`` in ONE pass over the
        rows of every node type — inside the row gather for a ``LazyRows`` input, a streaming pass over a resident one."""
        from .tensor import local_gather
        x, a_src, a_dst = {}, {}, {}
        self._by_id = set()          # node types whose terms are those of the TABLE's rows (read through the node list)
        for t in graph.node_types:
            v = xs.get(t)
            if v is None:
                continue
            keys = [(a_src if end == "src" else a_dst, et) for end, et in self._term_keys(t)]
            vt = self._terms_matrix(t)
            H = self.conv(keys[0][1]).heads if keys else 0
            slabs = None
            if isinstance(v, LazyRows):
                n, F_ = len(v), v.table.shape[1]
                terms_ok = vt is not None and n > 0 and H == 4 and v.table.dtype == torch.float32 and gather_terms_supported(F_, vt.shape[1])
                if self.fetch_in_layer and terms_ok and v.ids.dtype == torch.int64 and isinstance(v.table, torch.Tensor) \
                        and v._rows is None and v.table.stride(1) == 1 and v.table.stride(0) % 4 == 0 and v.table.data_ptr() % 16 == 0:
                    x[t] = v
                    if 2 * v.table.shape[0] <= n:
                        # the group lists a table row once per mini-batch that sampled it: terms of the TABLE's rows (made in
                        # every call: nothing is kept between calls), which the relation kernels read through the node lists
                        slabs = _stage("attn_terms(table)" + self.stage_tag, lambda: rows_terms(v.table, vt, heads=4))
                        self._by_id.add(t)
                    else:
                        slabs = _stage("attn_terms(lazy)" + self.stage_tag, lambda: lazy_rows_terms(v.table, v.ids, vt, heads=4))
                    for k, (dst, et) in enumerate(keys):
                        dst[et] = slabs[k]
                    continue
                buf = torch.empty((n, F_), dtype=torch.float32, device=v.table.device)
                if terms_ok:
                    x[t], slabs = _stage("gather+attn_terms" + self.stage_tag, lambda: gather_with_terms(v.table, v.ids, vt, out=buf, heads=4))
                else:
                    x[t] = _stage("gather", lambda: local_gather(v.table, v.ids, buf))
            else:
                x[t] = v
            if not keys or x[t].shape[0] == 0:
                continue
            if slabs is None:
                xt = x[t]
                if H == 4 and xt.stride(1) == 1 and xt.stride(0) % 4 == 0 and xt.data_ptr() % 16 == 0 \
                        and gather_terms_supported(int(xt.shape[1]), int(vt.shape[1])):
                    slabs = _stage("attn_terms" + self.stage_tag, lambda: rows_terms(xt, vt, heads=4))
                else:
                    both = _stage("attn_terms" + self.stage_tag, lambda: xt @ vt)
                    slabs = both.view(both.shape[0], len(keys), H).permute(1, 0, 2).contiguous()
            for k, (dst, et) in enumerate(keys):
                dst[et] = slabs[k]
        return x, a_src, a_dst

    def _forward_layer(self, xs, graph: HeteroLayerGraph, act=None):
        assert act in (None, "relu")
        relu = act == "relu"
        x, a_src, a_dst = self._attention_terms(xs, graph)
        dev = next(iter(x.values())).device
        out = {}
        groups = {}
        listed = set()
        for r in graph.relations:
            groups.setdefault((r.hop, r.edge_type[2]), []).append(r)
        for t, n in graph.n_out.items():
            if n > 0 and any(dt == t for _, dt in groups):
                out[t] = torch.empty((n, self._width(t)), dtype=torch.float32, device=dev)
        for (hop, dt), mine in sorted(groups.items(), key=lambda kv: (kv[0][0], kv[0][1])):
            n_f, HC = mine[0].n_rows, self._width(dt)
            if n_f == 0:
                continue
            bias, place = self._bias(dt), mine[0].out_rows
            live = [r for r in mine if r.n_edges > 0]      # (a relation that sampled nothing adds nothing to the sum)
            acc = torch.empty((n_f, HC), dtype=torch.float32, device=dev)
            c0 = self.conv(mine[0].edge_type)
            H, C = c0.heads, c0.out_channels
            one_pass = bool(live) and _GAT_TRANSFORM_FUSED and all(
                gat_transform_supported(x[r.edge_type[0]].shape[1], H, C) for r in live)
            target = out[dt] if place is not None else None
            if one_pass and place is None:
                out[dt] = target = torch.empty((n_f, HC), dtype=torch.float32, device=dev)
            for j, r in enumerate(live):
                et = r.edge_type
                w = self._rel(et)[0]
                xsrc, last = x[et[0]], j == len(live) - 1
                ids, through = None, {}
                if isinstance(xsrc, LazyRows):      # fetch in the layer: the kernels read the table through the node list
                    xsrc, ids = xsrc.table, xsrc.ids
                    through = dict(src_ids=ids, src_terms_by_id=et[0] in self._by_id)
                    if et[2] in self._by_id:
                        through.update(dst_ids=x[et[2]].ids, dst_terms_by_id=True)
                elif et[2] in self._by_id and et not in listed:
                    # (terms of the destination TABLE's rows next to a resident source: per-list terms, made once per edge type —
                    #  the relation runs in several hops)
                    a_dst[et] = gather_term_slabs(a_dst[et].unsqueeze(0), x[et[2]].ids)[0]
                    listed.add(et)
                tail = dict(acc_in=acc if j > 0 else None, bias=bias if (last and one_pass) else None, relu=last and one_pass and relu,
                            out_rows=place if (last and one_pass) else None, out=target if (last and one_pass) else acc)
                name = "%s hop %d (%d rows, %d edges)" % (et[1], hop + 1, n_f, r.n_edges)
                if one_pass and _GAT_LAYER_FUSED and r.fanout <= self.fused_max_fanout and gat_layer_fused_supported(xsrc.shape[1], H, C):
                    # deep hop (fan-out <= 10): aggregation + dense tail as ONE kernel, the aggregate stays in LDS
                    _stage("gat%s+transform:" % self.stage_tag + name, lambda: gat_layer_fused(r.row_ptr, r.col, xsrc, a_src[et], a_dst[et], w, H,
                                                                            dst_rows=r.dst_rows, **through, **tail))
                    continue
                agg = _stage("gat%s:" % self.stage_tag + name, lambda: gat_aggregate_heads(r.row_ptr, r.col, xsrc, a_src[et], a_dst[et], H,
                                                                       dst_rows=r.dst_rows, **through))
                if one_pass:
                    _stage("transform" + self.stage_tag, lambda: gat_transform_heads_fused(agg, w, H, **tail))
                else:
                    _stage("transform" + self.stage_tag, lambda: gat_transform_heads(agg, w, H, out=acc, overwrite=j == 0))
            if one_pass:
                continue
            if not live:
                acc.zero_()        # no relation of this type sampled an edge in this hop: act(bias) rows
            if place is not None:
                _stage("bias_act", lambda: bias_act_rows(acc, bias, relu, place, out[dt]))
            else:
                out[dt] = _stage("bias_act", lambda: bias_act_rows(acc, bias, relu))
        return out

    def _width(self, dt):
        c = next(self.conv(et) for et in self.edge_types if et[2] == dt)
        return c.heads * c.out_channels if c.concat else c.out_channels

    def forward(self, x_dict, graph, act=None):
        if _capturing():
            # (the derived forms of this layer's parameters — folded attention vectors, weight tiles, pooled buffers — are cached
            #  in Python against the parameters' versions: a replayed graph would keep using the ones of the capture)
            raise RuntimeError("wholegraph_amd.nn.%s is not supported under HIP-graph capture (loader.PerBatchStep): its "
                               "derived-weight caches are not capture-safe; SAGEConv layers are" % type(self).__name__)
        if isinstance(graph, HeteroLayerGraph):
            needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
            plain = all(self.conv(et).concat and not self.conv(et).add_self_loops for et in self.edge_types)
            if not needs_grad and plain:
                return self._forward_layer(x_dict, graph, act)
            if plain and self.train_aggregate_first and all(
                    self.conv(et).heads in (1, 2, 4, 8) and self.conv(et).lin.weight.shape[1] % 4 == 0
                    and self.conv(et).lin.weight.shape[1] <= 256 for et in self.edge_types):
                return self._forward_layer_train(x_dict, graph, act)
            return self._forward_relations(x_dict, graph, act)
        out = {}
        for et in self.edge_types:
            if et not in graph or x_dict.get(et[0]) is None or x_dict.get(et[2]) is None:
                continue
            y = self.conv(et)((x_dict[et[0]], x_dict[et[2]]), graph[et])
            out[et[2]] = y if et[2] not in out else out[et[2]] + y
        return {t: torch.relu(v) for t, v in out.items()} if act == "relu" else out

    def _forward_layer_train(self, xs, graph: HeteroLayerGraph, act=None):
        """The call-group layer under autograd, AGGREGATE-FIRST like the inference route: per relation the attention-weighted
        sum of the UNTRANSFORMED source rows (``_GatAggregateHeads``: one kernel forward, one backward; a ``LazyRows`` input is
        read through its node list, its attention terms are those of the table's rows when the table is the shorter side), then
        the per-head weights on the few destination rows, HeteroConv's sum, bias, activation and row placement in torch — every
        parameter (lin weight, att_src, att_dst, bias) gets its gradient through ordinary autograd on top of the two kernels."""
        assert act in (None, "relu")
        X, ids, by_id, terms = {}, {}, {}, {}
        for t in graph.node_types:
            v = xs.get(t)
            if v is None:
                continue
            ends = self._term_keys(t)
            if isinstance(v, LazyRows) and isinstance(v.table, torch.Tensor) and v.ids.dtype == torch.int64 \
                    and v.table.dtype == torch.float32 and v.table.stride(1) == 1 and v.table.stride(0) % 4 == 0 \
                    and v.table.data_ptr() % 16 == 0 and v._rows is None:
                X[t], ids[t] = v.table, v.ids
                by_id[t] = 2 * v.table.shape[0] <= len(v)
                rows = v.table if by_id[t] else None
            else:
                X[t] = v.materialize() if isinstance(v, LazyRows) else v
                ids[t], by_id[t], rows = None, False, X[t]
            if not ends or len(v) == 0:
                continue
            folds = []
            for end, et in ends:          # differentiable folds: alpha = x (W . att)
                c = self.conv(et)
                w3 = c.lin.weight.t().reshape(c.lin.weight.shape[1], c.heads, c.out_channels)
                folds.append((w3 * (c.att_src if end == "src" else c.att_dst).view(1, c.heads, c.out_channels)).sum(-1))
            if rows is None:              # lazy, table longer than the list: terms of the listed rows
                rows = v.materialize()
            both = _NarrowTerms.apply(rows, torch.cat(folds, 1))
            H = self.conv(ends[0][1]).heads
            for k, (end, et) in enumerate(ends):
                terms[(end, et)] = both[:, k * H:(k + 1) * H]
        dev = next(iter(X.values())).device
        groups = {}
        for r in graph.relations:
            groups.setdefault((r.hop, r.edge_type[2]), []).append(r)
        out = {t: torch.zeros((n, self._width(t)), dtype=torch.float32, device=dev) for t, n in graph.n_out.items()
               if n > 0 and any(dt == t for _, dt in groups)}
        for (hop, dt), mine in sorted(groups.items(), key=lambda kv: (kv[0][0], kv[0][1])):
            n_f = mine[0].n_rows
            if n_f == 0:
                continue
            acc = None
            for r in mine:
                if r.n_edges == 0:
                    continue
                et = r.edge_type
                c = self.conv(et)
                H, C = c.heads, c.out_channels
                st, dt_ = et[0], et[2]
                lazy_src = ids[st] is not None
                # (the kernel's id-list mode needs a lazy source; a resident source next to by-id destination terms reads them per row)
                a_dst = terms[("dst", et)]
                dst_by_id = bool(by_id[dt_]) and lazy_src
                if by_id[dt_] and not lazy_src:
                    a_dst = a_dst[ids[dt_]]
                agg = _GatAggregateHeads.apply(X[st], terms[("src", et)], a_dst, r.row_ptr, r.col, H, r.dst_rows,
                                               ids[st], ids[dt_] if dst_by_id else None, bool(by_id[st]) and lazy_src, dst_by_id,
                                               c.negative_slope)
                F_ = X[st].shape[1]
                w3 = c.lin.weight.t().reshape(F_, H, C)
                if acc is not None and acc.shape[1:] != (H, C):      # (relations of one destination type with different head shapes)
                    acc = (acc.reshape(n_f, H * C) + _heads_transform(agg.view(n_f, H, F_), w3).reshape(n_f, H * C)).view(n_f, H, C)
                else:
                    acc = _heads_transform(agg.view(n_f, H, F_), w3, acc)
            if acc is not None:
                acc = acc.reshape(n_f, -1)
            if acc is None:
                acc = torch.zeros((n_f, self._width(dt)), dtype=torch.float32, device=dev)
            bs = [self.conv(r.edge_type).bias for r in mine if self.conv(r.edge_type).bias is not None]
            if bs:            # the relations' biases summed first: ONE pass over the rows (HeteroConv adds outputs, bias included)
                acc = acc + (bs[0] if len(bs) == 1 else torch.stack(bs).sum(0))
            if act == "relu":
                acc = torch.relu(acc)
            place = mine[0].out_rows
            if place is None:
                out[dt] = acc
            else:
                out[dt].index_copy_(0, place, acc)      # (in place: the hops of a type write disjoint rows of one buffer)
        return out

    def _forward_relations(self, xs, graph: HeteroLayerGraph, act=None):
        """The same layer relation by relation through ``GATConv`` (autograd: the training route of a call group)."""
        x = {t: (v.materialize() if isinstance(v, LazyRows) else v) for t, v in xs.items()}
        dev = next(iter(x.values())).device
        out = {t: torch.zeros((n, self._width(t)), dtype=torch.float32, device=dev) for t, n in graph.n_out.items() if n > 0}
        for r in graph.relations:
            if r.n_rows == 0:
                continue
            et = r.edge_type
            y = self.conv(et)((x[et[0]], x[et[2]][r.dst_rows]), [r.row_ptr, r.col])
            rows = r.out_rows if r.out_rows is not None else torch.arange(r.n_rows, device=dev)
            out[et[2]] = out[et[2]].index_add(0, rows, y)
        return {t: torch.relu(v) for t, v in out.items()} if act == "relu" else out
