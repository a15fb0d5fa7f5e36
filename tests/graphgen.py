"""Seeded synthetic CSR graphs shared by tests and bench (numpy only)."""
import numpy as np


def random_csr(n_nodes, n_edges, seed, col_dtype=np.int64, zero_deg_frac=0.05):
    """Uniform random multigraph in CSR (like the reference gtests' gen_csr_graph:
    /root/reference/cpp/tests/wholegraph_ops/graph_sampling_test_utils.cu:30-120)."""
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n_nodes, n_edges)
    if zero_deg_frac > 0:
        dead = rng.random(n_nodes) < zero_deg_frac
        src = src[~dead[src]]
    deg = np.bincount(src, minlength=n_nodes)
    row_ptr = np.zeros(n_nodes + 1, np.int64)
    row_ptr[1:] = np.cumsum(deg)
    col = rng.integers(0, n_nodes, row_ptr[-1]).astype(col_dtype)
    return row_ptr, col


def powerlaw_csr(n_nodes, avg_deg, seed, col_dtype=np.int64, max_deg=None, alpha=1.8):
    """Power-law out-degrees (Zipf), endpoints drawn degree-biased: hubs are both long rows and
    popular neighbours — the skew the sampler has to live with on products/papers-like graphs."""
    rng = np.random.default_rng(seed)
    deg = rng.zipf(alpha, n_nodes).astype(np.int64)
    cap = max_deg or max(16, n_nodes // 8)
    deg = np.minimum(deg, cap)
    scale = avg_deg / max(deg.mean(), 1e-9)
    deg = np.maximum((deg * scale).astype(np.int64), (rng.random(n_nodes) < 0.97).astype(np.int64))
    deg = np.minimum(deg, cap)
    row_ptr = np.zeros(n_nodes + 1, np.int64)
    row_ptr[1:] = np.cumsum(deg)
    E = int(row_ptr[-1])
    # degree-biased endpoints: pick a random edge slot and take its source
    owner = np.repeat(np.arange(n_nodes), deg)
    half = E // 2
    col = np.empty(E, dtype=np.int64)
    col[:half] = owner[rng.integers(0, E, half)]
    col[half:] = rng.integers(0, n_nodes, E - half)
    rng.shuffle(col)
    return row_ptr, col.astype(col_dtype)


def sage_layer_case(seed=2026, G=8, F=100, N=256, fanout=10):
    """A SAGE layer-1 input shaped like the products call group of bench.py, scaled to G mini-batches: a block-diagonal hop
    (batch b's destination rows only reference batch b's segment of x; the targets are the first rows of the segment, so
    self_rows[i] = segment start + i), degrees min(power-law, fan-out) with empty rows, x ~ U(-1, 1), weights ~ U(-.05, .05)
    as bench.py initialises them.  Shared by tests/golden/make_golden.py (freezes fp64 expectations) and the GPU test."""
    rng = np.random.default_rng(seed)
    n_dst_b = rng.integers(2000, 2800, G)
    n_src_b = n_dst_b + rng.integers(5000, 8000, G)
    src_off = np.concatenate([[0], np.cumsum(n_src_b)])
    deg, cols, self_rows = [], [], []
    for b in range(G):
        d = np.minimum(rng.zipf(1.6, n_dst_b[b]), fanout)
        d[rng.random(n_dst_b[b]) < 0.03] = 0
        deg.append(d)
        hub = rng.integers(0, n_src_b[b], 32)                        # a few hub sources are everyone's neighbour
        c = np.where(rng.random(d.sum()) < 0.3, hub[rng.integers(0, 32, d.sum())], rng.integers(0, n_src_b[b], d.sum()))
        cols.append(c + src_off[b])
        self_rows.append(np.arange(n_dst_b[b]) + src_off[b])
    deg = np.concatenate(deg)
    rp = np.zeros(deg.size + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    col = np.concatenate(cols).astype(np.int32)
    self_rows = np.concatenate(self_rows).astype(np.int64)
    x = (rng.random((int(src_off[-1]), F), dtype=np.float32) * 2 - 1)
    w_t = ((rng.random((2 * F, N), dtype=np.float32) - 0.5) * 0.1)
    bias = ((rng.random(N, dtype=np.float32) - 0.5) * 0.1)
    return rp, col, self_rows, x, w_t, bias


def gat_layer_case(seed=2027, G=6, F=128, H=4, C=64, fanout=10):
    """A GATConv layer shaped like one (hop, edge type) launch of the ogbn-mag-like call group of bench_mag.py, scaled to G
    mini-batches: block-diagonal hop with hub sources and empty rows, the launch's rows a SUBSET of a larger destination
    list (``dst_rows``), x ~ U(-1, 1), lin weight ~ U(-a, a) with a = 1 / sqrt(F), attention vectors ~ U(-.25, .25) folded into
    the [F, H] matrices the pipeline multiplies x by (bench_mag.make_params).  Shared by tests/golden/make_golden.py (freezes
    float64 expectations) and the GPU test."""
    rng = np.random.default_rng(seed)
    n_dst_b = rng.integers(700, 1100, G)
    n_src_b = n_dst_b + rng.integers(2500, 4000, G)
    src_off = np.concatenate([[0], np.cumsum(n_src_b)])
    deg, cols = [], []
    for b in range(G):
        d = np.minimum(rng.zipf(1.5, n_dst_b[b]), fanout)
        d[rng.random(n_dst_b[b]) < 0.05] = 0
        deg.append(d)
        hub = rng.integers(0, n_src_b[b], 16)
        c = np.where(rng.random(d.sum()) < 0.3, hub[rng.integers(0, 16, d.sum())], rng.integers(0, n_src_b[b], d.sum()))
        cols.append(c + src_off[b])
    deg = np.concatenate(deg)
    rp = np.zeros(deg.size + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    col = np.concatenate(cols).astype(np.int32)
    n_rows, n_src = deg.size, int(src_off[-1])
    n_all = n_rows + 1500
    dst_rows = np.sort(rng.permutation(n_all)[:n_rows]).astype(np.int64)
    x = rng.random((n_src, F), dtype=np.float32) * 2 - 1
    x_dst = rng.random((n_all, F), dtype=np.float32) * 2 - 1
    w = ((rng.random((F, H * C), dtype=np.float32) - 0.5) * (2.0 / np.sqrt(F))).astype(np.float32)
    att_s = ((rng.random((H, C), dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
    att_d = ((rng.random((H, C), dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
    bias = ((rng.random(H * C, dtype=np.float32) - 0.5) * 0.1).astype(np.float32)
    return rp, col, dst_rows, x, x_dst, w, att_s, att_d, bias
