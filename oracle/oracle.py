"""numpy/ctypes front-end of ``wg_oracle.c`` (TEST INFRASTRUCTURE ONLY — see package docstring).

Reference code each entry follows (paths relative to /root/reference):
  sample_offsets / unweighted_sample   cpp/tests/wholegraph_ops/graph_sampling_test_utils.cu:226-401
  weighted_sample                      cpp/tests/wholegraph_ops/graph_sampling_test_utils.cu:530-656
  append_unique                        cpp/tests/graph_ops/append_unique_test_utils.cu:52-157
  csr_add_self_loop                    cpp/src/graph_ops/csr_add_self_loop_func.cuh:13-32
  multilayer_sample                    python/pylibwholegraph/pylibwholegraph/torch/graph_structure.py:136-196
  gather / scatter                     cpp/src/wholememory_ops/functions/gather_scatter_func.cuh:242-305,508-587
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libwg_oracle.so")
_lib = None

__all__ = [
    "build", "lib", "pcg_raw_u32", "generate_random_positive_int",
    "generate_exponential_distribution_negative_float", "weighted_keys", "sample_offsets", "unweighted_sample_with_replacement",
    "unweighted_sample", "weighted_sample", "append_unique", "csr_add_self_loop",
    "multilayer_sample", "gather", "gather_rows", "scatter", "spmm_csr", "gat_csr", "gat_aggregate_heads", "gat_aggregate_heads_f64",
    "num_threads",
    "set_num_threads", "py_pcg_u32_stream", "unique_bounded",
]


def build(force=False):
    """Compile wg_oracle.c with gcc (idempotent)."""
    src = os.path.join(_HERE, "wg_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.wgo_sample_offsets.restype = ctypes.c_int
        _lib.wgo_append_unique.restype = ctypes.c_int
        _lib.wgo_num_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _is64(a):
    a = np.asarray(a)
    if a.dtype == np.int64:
        return 1
    if a.dtype == np.int32:
        return 0
    raise TypeError(f"index array must be int32/int64, got {a.dtype}")


def _c(a, dtype=None):
    return np.ascontiguousarray(a, dtype=dtype)


i64 = ctypes.c_int64
u64 = ctypes.c_uint64
cint = ctypes.c_int


# ---------------------------------------------------------------------------- RNG
def pcg_raw_u32(seed, subsequence, offset, n):
    out = np.empty(n, dtype=np.uint32)
    lib().wgo_pcg_raw_u32(u64(seed), u64(subsequence), u64(offset), _p(out), i64(n))
    return out


def generate_random_positive_int(seed, subsequence, n, dtype=np.int32):
    out = np.empty(n, dtype=dtype)
    lib().wgo_generate_random_positive_int(i64(seed), i64(subsequence), _p(out), i64(n), cint(_is64(out)))
    return out


def generate_exponential_distribution_negative_float(seed, subsequence, n):
    out = np.empty(n, dtype=np.float32)
    lib().wgo_generate_exponential_distribution_negative_float(i64(seed), i64(subsequence), _p(out), i64(n))
    return out


def weighted_keys(seed, subsequence, weights):
    w = _c(weights, np.float32)
    out = np.empty_like(w)
    lib().wgo_weighted_keys(i64(seed), i64(subsequence), _p(w), _p(out), i64(w.size))
    return out


def py_pcg_u32_stream(seed, subsequence, offset, n):
    """Independent pure-Python PCG32 (python ints), used to cross-check the C restatement."""
    mask = (1 << 64) - 1
    mult = 6364136223846793005
    inc = ((subsequence << 1) | 1) & mask
    state = 0

    def step():
        nonlocal state
        old = state
        state = (old * mult + inc) & mask
        x = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((x >> rot) | (x << ((-rot) & 31))) & 0xFFFFFFFF

    step()
    state = (state + seed) & mask
    step()
    for _ in range(offset):  # naive skip: one step at a time
        step()
    return [step() for _ in range(n)]


# ---------------------------------------------------------------------------- sampling
def sample_offsets(row_ptr, seeds, max_sample_count):
    row_ptr = _c(row_ptr, np.int64)
    seeds = _c(seeds)
    off = np.empty(seeds.size + 1, dtype=np.int32)
    lib().wgo_sample_offsets(_p(row_ptr), _p(seeds), cint(_is64(seeds)), i64(seeds.size),
                             cint(max_sample_count), _p(off))
    return off


def unweighted_sample(row_ptr, col, seeds, max_sample_count, random_seed):
    """-> (offsets int32[n+1], dst (col dtype), src_lid int32, edge_gid int64)."""
    row_ptr = _c(row_ptr, np.int64)
    col = _c(col)
    seeds = _c(seeds)
    off = sample_offsets(row_ptr, seeds, max_sample_count)
    total = int(off[-1])
    dst = np.empty(total, dtype=col.dtype)
    lid = np.empty(total, dtype=np.int32)
    gid = np.empty(total, dtype=np.int64)
    lib().wgo_unweighted_sample(_p(row_ptr), _p(col), cint(_is64(col)), _p(seeds), cint(_is64(seeds)),
                                i64(seeds.size), cint(max_sample_count), u64(random_seed & (2**64 - 1)),
                                _p(off), _p(dst), _p(lid), _p(gid))
    return off, dst, lid, gid


def unweighted_sample_with_replacement(row_ptr, col, seeds, sample_count, random_seed):
    """Uniform sampling WITH replacement (wgo_uniform_sample_with_replacement; no reference counterpart in tree).
    -> (offsets int32[n+1], dst (col dtype), src_lid int32, edge_gid int64)."""
    row_ptr = _c(row_ptr, np.int64)
    col = _c(col)
    seeds = _c(seeds)
    off = np.empty(seeds.size + 1, dtype=np.int32)
    args = (_p(row_ptr), _p(col), cint(_is64(col)), _p(seeds), cint(_is64(seeds)), i64(seeds.size), cint(sample_count),
            u64(random_seed & (2**64 - 1)), _p(off))
    lib().wgo_uniform_sample_with_replacement(*args, None, None, None)
    total = int(off[-1])
    dst = np.empty(total, dtype=col.dtype)
    lid = np.empty(total, dtype=np.int32)
    gid = np.empty(total, dtype=np.int64)
    lib().wgo_uniform_sample_with_replacement(*args, _p(dst), _p(lid), _p(gid))
    return off, dst, lid, gid


def weighted_sample(row_ptr, col, weights, seeds, max_sample_count, random_seed, return_keys=False):
    row_ptr = _c(row_ptr, np.int64)
    col = _c(col)
    weights = _c(weights)
    assert weights.dtype in (np.float32, np.float64)
    seeds = _c(seeds)
    off = sample_offsets(row_ptr, seeds, max_sample_count)
    total = int(off[-1])
    dst = np.empty(total, dtype=col.dtype)
    lid = np.empty(total, dtype=np.int32)
    gid = np.empty(total, dtype=np.int64)
    keys = np.empty(total, dtype=np.float32)
    lib().wgo_weighted_sample(_p(row_ptr), _p(col), cint(_is64(col)), _p(weights),
                              cint(1 if weights.dtype == np.float64 else 0), _p(seeds),
                              cint(_is64(seeds)), i64(seeds.size), cint(max_sample_count),
                              u64(random_seed & (2**64 - 1)), _p(off), _p(dst), _p(lid), _p(gid), _p(keys))
    if return_keys:
        return off, dst, lid, gid, keys
    return off, dst, lid, gid


# ---------------------------------------------------------------------------- renumber
def append_unique(targets, neighbors):
    """-> (unique [T+U], raw_to_unique int32[E]); new nodes in first-appearance order."""
    targets = _c(targets)
    neighbors = _c(neighbors, targets.dtype)
    T, E = targets.size, neighbors.size
    uniq = np.empty(T + E, dtype=targets.dtype)
    mp = np.empty(E, dtype=np.int32)
    U = lib().wgo_append_unique(_p(targets), i64(T), _p(neighbors), i64(E), cint(_is64(targets)),
                                _p(uniq), _p(mp))
    return uniq[: T + U].copy(), mp


def csr_add_self_loop(row_ptr, col):
    row_ptr = _c(row_ptr, np.int32)
    col = _c(col, np.int32)
    n = row_ptr.size - 1
    orp = np.empty(n + 1, dtype=np.int32)
    oc = np.empty(col.size + n, dtype=np.int32)
    lib().wgo_csr_add_self_loop(_p(row_ptr), _p(col), i64(n), _p(orp), _p(oc))
    return orp, oc


def multilayer_sample(row_ptr, col, seeds, max_neighbors, random_seeds, weights=None):
    """GraphStructure.multilayer_sample_without_replacement (graph_structure.py:136-196).

    ``random_seeds[k]`` is the seed of the k-th sampling call in execution order (seeds hop
    first).  Returns (target_gids[0..L], edge_indice[0..L-1], csr_row_ptr[...], csr_col_ind[...]).
    """
    hops = len(max_neighbors)
    tg = [None] * (hops + 1)
    ei, rp, ci = [None] * hops, [None] * hops, [None] * hops
    tg[hops] = _c(seeds)
    for i in range(hops - 1, -1, -1):
        k = hops - i - 1
        if weights is None:
            off, dst, lid, _ = unweighted_sample(row_ptr, col, tg[i + 1], max_neighbors[k], random_seeds[k])
        else:
            off, dst, lid, _ = weighted_sample(row_ptr, col, weights, tg[i + 1], max_neighbors[k], random_seeds[k])
        uniq, mp = append_unique(tg[i + 1], dst.astype(tg[i + 1].dtype))
        rp[i], ci[i] = off, mp
        ei[i] = np.stack([mp, lid])
        tg[i] = uniq
    return tg, ei, rp, ci


# ---------------------------------------------------------------------------- gather/scatter
def gather(table, idx, out=None, out_dtype=None):
    """out[i,:] = convert(table[idx[i],:]); idx<0 rows untouched (need ``out`` then)."""
    table = np.asarray(table)
    idx = np.asarray(idx)
    two_d = table.ndim == 2
    t2 = table if two_d else table[:, None]
    od = out_dtype or table.dtype
    if out is None:
        out = np.zeros((idx.size, t2.shape[1]) if two_d else (idx.size,), dtype=od)
    o2 = out if two_d else out[:, None]
    ok = idx >= 0
    o2[ok] = t2[idx[ok]].astype(o2.dtype)
    return out


def gather_rows(table, idx):
    """Same-dtype row gather through the C kernel (OpenMP over rows) — used by the CPU baseline."""
    table = np.ascontiguousarray(table)
    idx = _c(idx)
    out = np.empty((idx.size,) + table.shape[1:], dtype=table.dtype)
    row_bytes = table.dtype.itemsize * int(np.prod(table.shape[1:], dtype=np.int64))
    lib().wgo_gather_rows(_p(table), i64(row_bytes), _p(idx), cint(_is64(idx)), i64(idx.size), i64(row_bytes),
                          _p(out), i64(row_bytes))
    return out


def scatter(inp, idx, table):
    inp = np.asarray(inp)
    idx = np.asarray(idx)
    ok = idx >= 0
    # duplicates: last writer wins in index order (sequential semantics)
    for i in np.nonzero(ok)[0]:
        table[idx[i]] = inp[i].astype(table.dtype)
    return table


# ---------------------------------------------------------------------------- aggregation
def spmm_csr(row_ptr, col, x, mean=True, acc_double=False):
    row_ptr = _c(row_ptr, np.int32)
    col = _c(col, np.int32)
    x = _c(x, np.float32)
    n, F = row_ptr.size - 1, x.shape[1]
    out = np.empty((n, F), dtype=np.float32)
    lib().wgo_spmm_csr(_p(row_ptr), _p(col), i64(n), _p(x), i64(F), i64(F), cint(int(mean)),
                       cint(int(acc_double)), _p(out), i64(F))
    return out


def gat_csr(row_ptr, col, x, a_src, a_dst, slope=0.2):
    """x [N_src,H,C], a_src [N_src,H], a_dst [n_rows,H] -> (out [n_rows,H,C], alpha [E,H])."""
    row_ptr = _c(row_ptr, np.int32)
    col = _c(col, np.int32)
    x = _c(x, np.float32)
    a_src = _c(a_src, np.float32)
    a_dst = _c(a_dst, np.float32)
    n = row_ptr.size - 1
    H, C = x.shape[1], x.shape[2]
    out = np.empty((n, H, C), dtype=np.float32)
    alpha = np.empty((col.size, H), dtype=np.float32)
    lib().wgo_gat_csr(_p(row_ptr), _p(col), i64(n), _p(x), _p(a_src), _p(a_dst), i64(H), i64(C),
                      ctypes.c_float(slope), _p(alpha), _p(out))
    return out, alpha


def gat_aggregate_heads(row_ptr, col, x, a_src, a_dst, dst_rows=None, slope=0.2):
    """x [N_src, F] untransformed, a_src [N_src, H], a_dst [*, H] (row ``dst_rows[i]`` or i) -> agg [n_rows, H, F]."""
    row_ptr, col = _c(row_ptr, np.int32), _c(col, np.int32)
    x, a_src, a_dst = _c(x, np.float32), _c(a_src, np.float32), _c(a_dst, np.float32)
    dr = None if dst_rows is None else _c(dst_rows, np.int64)
    n, H, F = row_ptr.size - 1, a_src.shape[1], x.shape[1]
    out = np.empty((n, H, F), dtype=np.float32)
    lib().wgo_gat_aggregate_heads(_p(row_ptr), _p(col), i64(n), _p(x), i64(F), _p(a_src), _p(a_dst),
                                  None if dr is None else _p(dr), i64(H), ctypes.c_float(slope), _p(out))
    return out


def gat_aggregate_heads_f64(row_ptr, col, x, a_src, a_dst, dst_rows=None, slope=0.2):
    """``gat_aggregate_heads`` with EVERY step in float64 (numpy, no C): scores, LeakyReLU, per-destination softmax, the
    attention-weighted sum of the untransformed source rows — the restatement of PyG's GATConv formulas
    that the 1e-5 parity of the BASELINE config-5 pipeline is held against.  Inputs may be float32 or float64."""
    row_ptr = np.asarray(row_ptr).astype(np.int64)
    col = np.asarray(col).astype(np.int64)
    x, a_src, a_dst = (np.asarray(v, dtype=np.float64) for v in (x, a_src, a_dst))
    n, H, F = row_ptr.size - 1, a_src.shape[1], x.shape[1]
    out = np.zeros((n, H, F), dtype=np.float64)
    deg = np.diff(row_ptr)
    E = int(row_ptr[-1])
    if E == 0:
        return out
    dst = np.repeat(np.arange(n), deg)
    ad = a_dst[np.asarray(dst_rows, dtype=np.int64)[dst]] if dst_rows is not None else a_dst[dst]
    sc = a_src[col[:E]] + ad
    sc = np.where(sc > 0, sc, sc * slope)                                   # [E, H]
    live = deg > 0
    starts = row_ptr[:-1][live]
    mx = np.maximum.reduceat(sc, starts, axis=0)                            # per destination with at least one edge
    row_of_live = np.cumsum(live) - 1
    p = np.exp(sc - mx[row_of_live[dst]])
    den = np.add.reduceat(p, starts, axis=0)
    alpha = p / den[row_of_live[dst]]
    contrib = alpha[:, :, None] * x[col[:E]][:, None, :]                    # [E, H, F]
    out[live] = np.add.reduceat(contrib, starts, axis=0)
    return out


def num_threads():
    return int(lib().wgo_num_threads())


def set_num_threads(n):
    lib().wgo_set_num_threads(cint(n))


def unique_bounded(ids, id_bound):
    """CPU statement of wgamd_unique_bounded (include/wgamd_ext.h; this library's own op — the reference's NCCL gather
    exchanges every id, gather_op_impl_nccl.cu:23-171): -> (distinct non-negative ids ascending, int32 position of every id
    in that list, -1 for negative ids).  Test infrastructure only."""
    import numpy as np
    ids = np.asarray(ids).astype(np.int64)
    assert ids.size == 0 or ids.max() < id_bound
    distinct = np.unique(ids[ids >= 0])
    inverse = np.full(ids.shape, -1, np.int32)
    inverse[ids >= 0] = np.searchsorted(distinct, ids[ids >= 0]).astype(np.int32)
    return distinct, inverse
