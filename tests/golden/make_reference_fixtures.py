#!/usr/bin/env python
"""Generates tests/golden/reference_py_*.npz by IMPORTING the reference's own Python host samplers.

Runs in the build container only (it reads /root/reference; nothing of the reference travels to the
GPU box — only the .npz data this script writes).  Run from the repo root:

    python tests/golden/make_reference_fixtures.py

What is reference code here and what is not
-------------------------------------------
The reference keeps pure-Python restatements of its device ops next to its pytest cases:

* ``host_unweighted_sample_without_replacement`` (+ ``…_func``, ``unweighte_sample_without_replacement_base``)
  tests/wholegraph_torch/ops/test_wholegraph_unweighted_sample_without_replacement.py:22-211
* ``host_weighted_sample_without_replacement`` (+ ``…_func``)
  tests/wholegraph_torch/ops/test_wholegraph_weighted_sample_without_replacement.py:22-166
* ``host_neighbor_raw_to_unique``            tests/wholegraph_torch/ops/test_graph_append_unique.py:8-19
* ``host_add_csr_self_loop``                 tests/wholegraph_torch/ops/test_graph_add_csr_self_loop.py:9-28
* ``gen_csr_graph``, ``host_get_sample_offset_tensor``, ``host_sample_all_neighbors``
  pylibwholegraph/test_utils/test_comm.py:44-142

Those modules import the compiled binding (nvcc + raft + NCCL: not buildable here) at module import, so this
script installs STUB modules for exactly the imports that cannot be satisfied —
``pylibwholegraph.binding.wholememory_binding`` (only the ``WholeMemoryDataType`` names are touched by the host
functions), ``pylibwholegraph.utils.multiprocess``, ``pylibwholegraph.torch.initialize``,
``pylibwholegraph.torch.dlpack_utils``, ``pylibwholegraph.torch.graph_ops`` — and a
``pylibwholegraph.torch.wholegraph_ops`` whose two host RNG helpers (``generate_random_positive_int_cpu``,
``generate_exponential_distribution_negative_float_cpu``: in the reference thin wrappers over the C symbols of
``wholegraph_op.h:82-94``) call the SAME-NAMED C symbols exported by THIS repo's libwholegraph_amd.so.  Everything else
— the launch tables, the RNG→index mapping, the Fisher–Yates table, the 128/256-lane neighbour ownership of the
weighted sampler, its key composition `(1/w)·r` and top-M selection, offsets, sample-all — is the reference's code,
imported from where it lies and executed unmodified.

What this pins: oracle/wg_oracle.c (and the HIP ops) == reference Python host sampler ∘ {this library's two RNG
helpers}.  What stays assumption A1: that those two helpers produce raft's PCGenerator stream (the
DeviceState ctor's skip by `subsequence`, and how raft composes float / 64-bit draws) — raft is not in the image.
"""
import importlib.util
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_PKG = "/root/reference/python/pylibwholegraph/pylibwholegraph"
REF_OPS = os.path.join(REF_PKG, "tests", "wholegraph_torch", "ops")

sys.path.insert(0, os.path.join(ROOT, "cugraph-gnn_amd"))
sys.path.insert(0, ROOT)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    """Stand-ins for what the reference test modules import but the host functions never execute."""
    from wholegraph_amd import wholegraph_ops as amd_ops  # ctypes view of libwholegraph_amd.so (host symbols only)

    class WholeMemoryDataType:  # names compared by host_weighted_sample_without_replacement (…weighted…py:140-142)
        DtFloat, DtHalf, DtDouble, DtBF16, DtInt, DtInt64, DtInt16, DtInt8 = range(1, 9)

    def _never(*a, **k):
        raise RuntimeError("stubbed reference entry point called: only host functions may run here")

    pkg = _stub("pylibwholegraph")
    pkg.__path__ = []  # a package with no real sub-modules: every import below resolves to a stub or an explicit load
    _stub("pylibwholegraph.binding").__path__ = []
    wmb = _stub("pylibwholegraph.binding.wholememory_binding", WholeMemoryDataType=WholeMemoryDataType,
                finalize=_never, create_wholememory_array=_never, destroy_wholememory_tensor=_never)
    sys.modules["pylibwholegraph.binding"].wholememory_binding = wmb
    _stub("pylibwholegraph.utils").__path__ = []
    _stub("pylibwholegraph.utils.multiprocess", multiprocess_run=_never)
    _stub("pylibwholegraph.torch").__path__ = []
    _stub("pylibwholegraph.torch.initialize", init_torch_env_and_create_wm_comm=_never)
    _stub("pylibwholegraph.torch.dlpack_utils", torch_import_from_dlpack=_never)
    _stub("pylibwholegraph.torch.graph_ops", append_unique=_never, add_csr_self_loop=_never)
    _stub("pylibwholegraph.torch.wholegraph_ops",
          # reference: torch/wholegraph_ops.py:158-175 → C symbols of wholegraph_op.h:82-94; here: the same symbols
          # of libwholegraph_amd.so (include/wgamd_ops.h:76-81)
          generate_random_positive_int_cpu=amd_ops.generate_random_positive_int_cpu,
          generate_exponential_distribution_negative_float_cpu=(
              amd_ops.generate_exponential_distribution_negative_float_cpu),
          unweighted_sample_without_replacement=_never, weighted_sample_without_replacement=_never)
    _stub("pylibwholegraph.test_utils").__path__ = []
    return wmb


def load_reference(modname, path):
    """Import a reference source file from where it lies (never copied)."""
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def npy(t):
    return t.numpy().copy() if isinstance(t, torch.Tensor) else np.asarray(t)


def main():
    assert os.path.isdir(REF_PKG), "build-container only: /root/reference is not present"
    wmb = install_stubs()
    test_comm = load_reference("pylibwholegraph.test_utils.test_comm", os.path.join(REF_PKG, "test_utils", "test_comm.py"))
    ref_u = load_reference("ref_test_unweighted",
                           os.path.join(REF_OPS, "test_wholegraph_unweighted_sample_without_replacement.py"))
    ref_w = load_reference("ref_test_weighted",
                           os.path.join(REF_OPS, "test_wholegraph_weighted_sample_without_replacement.py"))
    ref_au = load_reference("ref_test_append_unique", os.path.join(REF_OPS, "test_graph_append_unique.py"))
    ref_sl = load_reference("ref_test_self_loop", os.path.join(REF_OPS, "test_graph_add_csr_self_loop.py"))
    import oracle  # only for the ORDER of append_unique's new nodes, which the reference leaves unpinned (see below)

    torch.manual_seed(20260929)
    random.seed(20260929)
    DT = wmb.WholeMemoryDataType
    tdt = {"int32": torch.int32, "int64": torch.int64}

    # ------------------------------------------------------------------ graphs (reference generator, test_comm.py:44-77)
    # g103: the reference pytest's own size (…unweighted…py:355-358: 103 nodes / 1043 edges / 13 centres);
    # gwide: rows long enough for every launch-table class the walk uses and for M = 200 / 300 (…_func.cuh tables).
    graphs = {}
    # g113: the weighted pytest's size (…weighted…py:353-356: 113 nodes / 1043 edges / 13 centres).
    for name, (V, E, NB) in {"g103": (103, 1043, None), "g113": (113, 1043, None), "gwide": (24, 9000, 700),
                             "ghub": (6, 6900, 1500)}.items():  # ghub: rows of ~1150 (> 1024 candidates; M up to 1000)
        rp, col, w32 = test_comm.gen_csr_graph(V, E, neighbor_node_count=NB, csr_row_dtype=torch.int64,
                                               csr_col_dtype=torch.int64, weight_dtype=torch.float32)
        graphs[name] = (V, rp, col, w32)

    out_u, out_w = {}, {}
    cases_u, cases_w = [], []

    def centres(V, n, dtype):
        return torch.randint(0, V, (n,), dtype=dtype)

    # ------------------------------------------------------------------ uniform sampling
    # reference parameter set (M in {11, -1}; both id widths for centres and columns) + the BASELINE fan-outs and the
    # launch-table boundaries (B·items: 32·1, 32·2, 32·3, 64·2, 64·3, 128·2 …; …unweighted…py:47-117).
    plan_u = [("g103", M, cd, kd, 13) for M in (11, -1) for cd in ("int32", "int64") for kd in ("int32", "int64")]
    plan_u += [("g103", M, "int64", "int64", 13) for M in (25, 10, 15, 5, 40, 70)]
    plan_u += [("gwide", M, "int64", "int32", 6) for M in (25, 10, 15, 5, 32, 33, 40, 64, 65, 70, 96, 97, 128, 200, 300)]
    plan_u += [("ghub", M, "int32", "int64", 6) for M in (25, 385, 512, 1000, 1024)]
    for gname, M, cd, kd, n in plan_u:
        V, rp, col, _ = graphs[gname]
        c = centres(V, n, tdt[cd])
        colk = col.to(tdt[kd])
        seed = random.randint(1, 10000)  # …unweighted…py:272
        off, dst, lid, gid = ref_u.host_unweighted_sample_without_replacement(rp, colk, c, M, tdt[kd], seed)
        tag = f"u{len(cases_u)}"
        cases_u.append((tag, gname, M, seed, kd))
        out_u[f"{tag}_centres"] = npy(c)
        out_u[f"{tag}_offset"], out_u[f"{tag}_dst"], out_u[f"{tag}_lid"], out_u[f"{tag}_gid"] = map(npy, (off, dst, lid, gid))
    for gname, (V, rp, col, w32) in graphs.items():
        out_u[f"{gname}_row_ptr"], out_u[f"{gname}_col"] = npy(rp), npy(col)  # columns int64; a case casts to its width
    out_u["cases"] = np.array([f"{t}|{g}|{M}|{s}|{kd}" for t, g, M, s, kd in cases_u])

    # ------------------------------------------------------------------ weighted sampling
    # reference parameter set (…weighted…py:353-360: M = 11, float and double weights, both id widths) + BASELINE
    # fan-outs and rows past one 128-lane round (lane j owns neighbours j, j+128, …) and the 256-lane layout (M > 256).
    # (M <= 0 is not in the reference's weighted set: its host function hands the binding's dtype enum to torch.empty
    # there, …weighted…py:126-134, and raises.)
    plan_w = [("g113", 11, cd, kd, wd, 13) for cd in ("int32", "int64") for kd in ("int32", "int64") for wd in ("f32", "f64")]
    plan_w += [("g103", M, "int32", "int64", "f32", 13) for M in (25, 10, 15, 5, 40, 70)]
    plan_w += [("gwide", M, "int64", "int32", "f32", 5) for M in (25, 10, 129, 200, 300)]
    plan_w += [("ghub", M, "int64", "int64", wd, 6) for M, wd in ((25, "f32"), (10, "f64"), (300, "f32"))]
    for gname, M, cd, kd, wd, n in plan_w:
        V, rp, col, w32 = graphs[gname]
        c = centres(V, n, tdt[cd])
        colk = col.to(tdt[kd])
        w = w32 if wd == "f32" else w32.double()
        seed = random.randint(1, 10000)
        off, dst, lid, gid = ref_w.host_weighted_sample_without_replacement(
            rp, colk, w, c, M, DT.DtInt if kd == "int32" else DT.DtInt64, seed)
        tag = f"w{len(cases_w)}"
        cases_w.append((tag, gname, M, seed, kd, wd))
        out_w[f"{tag}_centres"] = npy(c)
        out_w[f"{tag}_offset"], out_w[f"{tag}_dst"], out_w[f"{tag}_lid"], out_w[f"{tag}_gid"] = map(npy, (off, dst, lid, gid))
    for gname, (V, rp, col, w32) in graphs.items():
        out_w[f"{gname}_row_ptr"], out_w[f"{gname}_col"], out_w[f"{gname}_weight_f32"] = npy(rp), npy(col), npy(w32)
    out_w["cases"] = np.array([f"{t}|{g}|{M}|{s}|{kd}|{wd}" for t, g, M, s, kd, wd in cases_w])  # f64 = the f32 values widened

    # ------------------------------------------------------------------ append_unique + self loops
    # reference set-up (test_graph_append_unique.py:26-31): targets = randperm(n)[:T], neighbours = randint(0, n, n).
    # The reference leaves the ORDER of the new nodes unpinned (hash-slot order); this repo fixes it to first-appearance
    # order, so the unique list handed to the reference's host_neighbor_raw_to_unique is the oracle's; the mapping and
    # the sorted set are then reference-computed.
    out_g = {}
    k = 0
    for T, n in ((10, 100), (113, 1987), (0, 50), (40, 40)):
        for dt in (torch.int32, torch.int64):
            tgt = torch.randperm(n, dtype=dt)[:T]
            nbr = torch.randint(0, n, (n,), dtype=dt)
            uniq, _ = oracle.append_unique(npy(tgt), npy(nbr))
            mapping = ref_au.host_neighbor_raw_to_unique(torch.from_numpy(uniq), nbr)
            sorted_set = torch.unique(torch.cat((tgt, nbr), 0), sorted=True)  # test_graph_append_unique.py:55-57
            out_g[f"au{k}_targets"], out_g[f"au{k}_neighbors"] = npy(tgt), npy(nbr)
            out_g[f"au{k}_unique_first_appearance"] = uniq
            out_g[f"au{k}_raw_to_unique"], out_g[f"au{k}_sorted_set"] = npy(mapping), npy(sorted_set)
            k += 1
    out_g["n_append_unique"] = np.int64(k)
    k = 0
    for T, NB, E in ((101, 157, 1001), (113, 193, 1001), (113, 1987, 2305), (7, 7, 0)):  # test_graph_add_csr_self_loop.py:60-62 (+ an empty graph)
        rp, col, _ = test_comm.gen_csr_graph(T, E, neighbor_node_count=NB, csr_row_dtype=torch.int32,
                                             csr_col_dtype=torch.int32, weight_dtype=torch.float32)
        orp, ocol = ref_sl.host_add_csr_self_loop(rp, col)
        out_g[f"sl{k}_row_ptr"], out_g[f"sl{k}_col"] = npy(rp), npy(col)
        out_g[f"sl{k}_out_row_ptr"], out_g[f"sl{k}_out_col"] = npy(orp), npy(ocol)
        k += 1
    out_g["n_self_loop"] = np.int64(k)

    np.savez_compressed(os.path.join(HERE, "reference_py_unweighted.npz"), **out_u)
    np.savez_compressed(os.path.join(HERE, "reference_py_weighted.npz"), **out_w)
    np.savez_compressed(os.path.join(HERE, "reference_py_graph_ops.npz"), **out_g)
    print("uniform cases:", len(cases_u), " weighted cases:", len(cases_w), " graph-op arrays:", len(out_g))


if __name__ == "__main__":
    main()
