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
