#!/usr/bin/env python
"""Regenerates the golden vectors under tests/golden/ (run from the repo root).

Neither the reference's compiled path (nvcc + raft + NCCL) nor its Python packages can run in the
build container, and the reference stores NO golden vectors for this path — its
tests recompute a host reference every run.  The fixtures are therefore produced by this repo's CPU
restatement of that host reference (oracle/wg_oracle.c) and FROZEN here, so that later edits to the
oracle cannot drift silently; what pins the restatement itself is listed in tests/test_oracle_*.py
(PCG32 published demo vector, pure-Python twins, the reference's docstring / tiny-graph examples).

karate.csv is the reference's own test data file (/root/reference/datasets/karate.csv, 156 directed
lines `src dst weight`), copied as data.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
from graphgen import gat_layer_case, powerlaw_csr, random_csr, sage_layer_case  # noqa: E402


def karate_csr():
    e = np.loadtxt(os.path.join(HERE, "karate.csv"), dtype=np.int64, usecols=(0, 1))
    order = np.lexsort((e[:, 1], e[:, 0]))
    e = e[order]
    V = int(e.max()) + 1
    row_ptr = np.zeros(V + 1, np.int64)
    row_ptr[1:] = np.cumsum(np.bincount(e[:, 0], minlength=V))
    return row_ptr, e[:, 1].copy()


def main():
    out = {}
    # RNG stream (assumption A1: DeviceState ctor skips ahead by `subsequence`)
    out["rng_i31_seed42_sub0"] = oracle.generate_random_positive_int(42, 0, 8)
    out["rng_i31_seed42_sub5"] = oracle.generate_random_positive_int(42, 5, 8)
    out["rng_i63_seed7_sub3"] = oracle.generate_random_positive_int(7, 3, 4, np.int64)
    out["rng_expneg_seed9_sub2"] = oracle.generate_exponential_distribution_negative_float(9, 2, 8)
    # S0: karate, fan-out [5,5], seeds 0..33 in batches of 16 (BASELINE config 0)
    rp, col = karate_csr()
    out["karate_row_ptr"], out["karate_col"] = rp, col
    for b, seeds in enumerate(np.array_split(np.arange(34, dtype=np.int64), [16, 32])):
        tg, ei, orp, oci = oracle.multilayer_sample(rp, col, seeds, [5, 5], [62 + 2 * b, 63 + 2 * b])
        for i, t in enumerate(tg):
            out[f"karate_b{b}_target_gids_{i}"] = t
        for i in range(2):
            out[f"karate_b{b}_csr_row_ptr_{i}"] = orp[i]
            out[f"karate_b{b}_csr_col_ind_{i}"] = oci[i]
            out[f"karate_b{b}_edge_indice_{i}"] = ei[i]
    # the reference pytest's graph size: 103 nodes / 1043 edges / 13 seeds, M in {11, -1}
    rp, col = random_csr(103, 1043, seed=2024, col_dtype=np.int32, zero_deg_frac=0.0)
    seeds = np.random.default_rng(1).integers(0, 103, 13).astype(np.int32)
    out["g103_row_ptr"], out["g103_col"], out["g103_seeds"] = rp, col, seeds
    for M in (11, -1, 40, 70):
        off, dst, lid, gid = oracle.unweighted_sample(rp, col, seeds, M, 1234)
        out[f"g103_M{M}_offset"], out[f"g103_M{M}_dst"] = off, dst
        out[f"g103_M{M}_lid"], out[f"g103_M{M}_gid"] = lid, gid
    w = np.random.default_rng(3).random(col.size).astype(np.float32) + 0.01
    out["g103_weight"] = w
    off, dst, lid, gid = oracle.weighted_sample(rp, col, w, seeds, 5, 99)
    out["g103_w5_offset"], out["g103_w5_gid"] = off, gid
    # power-law, BASELINE fan-outs
    rp, col = powerlaw_csr(3000, 25, seed=5, max_deg=900)
    seeds = np.random.default_rng(2).permutation(3000)[:64].astype(np.int64)
    out["pl_seeds"] = seeds
    tg, ei, orp, oci = oracle.multilayer_sample(rp, col, seeds, [25, 10], [62, 63])
    out["pl_n_id"] = tg[0]
    out["pl_csr_row_ptr_0"], out["pl_csr_col_ind_0"] = orp[0], oci[0]
    out["pl_csr_row_ptr_1"], out["pl_csr_col_ind_1"] = orp[1], oci[1]
    np.savez_compressed(os.path.join(HERE, "hotpath_golden.npz"), **out)
    print("wrote", len(out), "arrays")
    sage_layer_golden()
    gat_layer_golden()


def sage_layer_golden():
    """Freezes a SAGE layer (F=100 -> 256, ReLU, mean) at the products shape: fp64 expectations for 512 sampled rows of a
    block-diagonal 8-batch hop, so that fp32 results stay put across kernel rewrites (tests/test_gpu_aggregate.py).  The
    inputs are regenerated from the seed by tests/graphgen.sage_layer_case; their checksum is stored with the fixture."""
    import hashlib
    rp, col, self_rows, x, w_t, bias = sage_layer_case()
    n_dst = rp.size - 1
    rng = np.random.default_rng(7)
    rows = np.unique(np.concatenate([rng.integers(0, n_dst, 500), [0, n_dst - 1], np.nonzero(np.diff(rp) == 0)[0][:10]]))
    cat = np.zeros((rows.size, 2 * x.shape[1]), np.float64)
    for j, r in enumerate(rows):
        nb = col[rp[r]:rp[r + 1]]
        if nb.size:
            cat[j, :x.shape[1]] = x[nb].astype(np.float64).sum(0) / nb.size
        cat[j, x.shape[1]:] = x[self_rows[r]]
    pre = cat @ w_t.astype(np.float64) + bias
    scale = np.abs(cat) @ np.abs(w_t).astype(np.float64) + np.abs(bias)
    h = hashlib.sha256()
    for a in (rp, col, self_rows, x, w_t, bias):
        h.update(np.ascontiguousarray(a).tobytes())
    np.savez_compressed(os.path.join(HERE, "sage_layer_golden.npz"), rows=rows, pre_activation=pre, scale=scale,
                        inputs_sha256=np.array(h.hexdigest()))
    print("sage layer golden:", rows.size, "rows of", n_dst, "; inputs sha256", h.hexdigest()[:16])


def gat_layer_golden():
    """Freezes a GATConv layer (F = 128 -> 4 heads x 64, LeakyReLU 0.2, per-destination softmax, bias, ReLU) at the shape of
    one launch of the ogbn-mag-like pipeline (BASELINE configs[4]): float64 expectations, computed in GATConv's OWN order
    (transform every source row, scores from the transformed rows and the attention vectors, softmax, weighted sum) for 400
    sampled rows.  tests/test_gpu_mag_pipeline.py holds the device's aggregate-first formulation to it at 1e-5."""
    import hashlib
    rp, col, dst_rows, x, x_dst, w, att_s, att_d, bias = gat_layer_case()
    F, (H, C) = x.shape[1], att_s.shape
    n_rows = rp.size - 1
    rng = np.random.default_rng(11)
    rows = np.unique(np.concatenate([rng.integers(0, n_rows, 400), [0, n_rows - 1], np.nonzero(np.diff(rp) == 0)[0][:8]]))
    w64 = w.astype(np.float64).reshape(F, H, C)
    pre = np.zeros((rows.size, H, C))
    scale = np.zeros((rows.size, H, C))
    for j, r in enumerate(rows):
        nb = col[rp[r]:rp[r + 1]]
        if nb.size == 0:
            continue
        hs = np.einsum("ef,fhc->ehc", x[nb].astype(np.float64), w64)                       # lin(x_j)          [e, H, C]
        hd = np.einsum("f,fhc->hc", x_dst[dst_rows[r]].astype(np.float64), w64)            # lin(x_i)          [H, C]
        sc = (hs * att_s.astype(np.float64)).sum(-1) + (hd * att_d.astype(np.float64)).sum(-1)   # [e, H]
        sc = np.where(sc > 0, sc, 0.2 * sc)
        p = np.exp(sc - sc.max(0))
        alpha = p / p.sum(0)
        pre[j] = np.einsum("eh,ehc->hc", alpha, hs)
        scale[j] = np.einsum("eh,ehc->hc", alpha, np.einsum("ef,fhc->ehc", np.abs(x[nb]).astype(np.float64), np.abs(w64)))
    pre = pre.reshape(rows.size, H * C) + bias
    scale = scale.reshape(rows.size, H * C) + np.abs(bias)
    h = hashlib.sha256()
    for a in (rp, col, dst_rows, x, x_dst, w, att_s, att_d, bias):
        h.update(np.ascontiguousarray(a).tobytes())
    np.savez_compressed(os.path.join(HERE, "gat_layer_golden.npz"), rows=rows, pre_activation=pre, scale=scale,
                        inputs_sha256=np.array(h.hexdigest()))
    print("gat layer golden:", rows.size, "rows of", n_rows, "; inputs sha256", h.hexdigest()[:16])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "gat":
        gat_layer_golden()          # only the GAT layer fixture (the others stay byte-identical)
    else:
        main()
