"""Training with the reference's optimizer semantics over call groups: SAMPLE per call group, STEP per mini-batch.

Every training loop of the reference steps its optimizer once per mini-batch of ``batch_size`` seeds
(/root/reference/python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:119-125,
python/cugraph-pyg/cugraph_pyg/examples/gcn_dist_mnmg.py: ``loss.backward(); optimizer.step()`` inside ``for batch in loader``).
On an MI355X one such step over 1024 seeds at fan-out [25, 10] is a few tens of microseconds of kernel time behind ~30
launches, so the loop is bound by the host unless the launches are taken off it.  ``PerBatchStep`` does that without
changing what is computed:

* the walk, the renumbering and (lazily) the feature rows are still produced once per CALL GROUP (``loader.call_groups()``);
* one launch (``wgamd_call_group_stage_batch``) copies mini-batch b of the group — its trimmed per-layer graphs in
  batch-local ids, its vertex list — into buffers of FIXED size and address, padded to capacities (the slack edges belong to
  slack rows whose outputs nobody reads and whose gradients are zero);
* the user's whole step — forward over ``batch.layer_graph(j)``, loss, ``backward()``, ``optimizer.step()`` — is captured
  ONCE as a HIP graph over those buffers (``torch.cuda.graph``) and replayed for every mini-batch.

Per mini-batch the host issues two launches.  The numbers the step computes are those of the eager per-batch loop
(tests/test_gpu_per_batch_step.py: gradients against the float64 formula of the same mini-batch at 1e-5).
"""
import ctypes
from typing import Callable, List, Optional

import torch

from wholegraph_amd import _lib as L
from wholegraph_amd import nn as wnn
from wholegraph_amd.env import get_stream, torch_dtype_to_wm
from wholegraph_amd.nn import HopGraph, LayerGraph, LazyRows


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


class StagedBatch:
    """One mini-batch in fixed-size buffers — what ``step_fn`` receives.

    ``x`` (``LazyRows`` over the feature table, or the gathered rows), ``n_id`` [node_cap] global ids (the first
    ``batch_size`` are the seeds; rows past the live count repeat the first id), ``seeds`` = ``n_id[:batch_size]``,
    ``layer_graph(j)`` = the trimmed graph of layer j as ``wholegraph_amd.nn.LayerGraph`` (layer j's output rows: hop 0's
    ``row_cap[0]`` rows — the seeds first — then hop 1's ...; the LAST layer's output is ``[row_cap[0], N]`` and its first
    ``batch_size`` rows are the seeds').  ``seed_mask`` [row_cap[0]] float: 1 for a live seed row, 0 for padding (a ragged last
    mini-batch) — multiply per-seed losses by it and divide by ``n_live_seeds``."""

    def __init__(self, hops: int, batch_size: int, row_cap: List[int], edge_cap: List[int], node_cap: int, id_dtype, device,
                 table: Optional[torch.Tensor]):
        self.hops, self.batch_size, self.row_cap, self.edge_cap, self.node_cap = hops, batch_size, list(row_cap), list(edge_cap), node_cap
        i32 = dict(dtype=torch.int32, device=device)
        R, E = sum(row_cap), sum(edge_cap)
        # The hops' arrays lie back to back, and `row_ptr_all` is their CSRs as ONE CSR (hop k's rows from sum(row_cap[:k]), its
        # edges from sum(edge_cap[:k])): a layer over hops 0..j is ONE launch over a prefix instead of j + 1 launches — every
        # launch of a mini-batch's step sits on its latency floor (30-47 us for 1-10 k rows), so launches are what a step costs.
        self.row_ptr_all = torch.zeros(R + 1, **i32)
        self.inv_deg_all = torch.ones(R, dtype=torch.float32, device=device)    # 1 / max(degree, 1): the mean's backward
        self._seed_mask = torch.zeros(row_cap[0], dtype=torch.float32, device=device)
        self.self0_all = torch.zeros(R, dtype=torch.int64, device=device)
        self.col_all = torch.zeros(E, **i32)
        # (the LAST hop's sources include the vertices it discovered itself, which no layer's output holds: no col_seg for it)
        self.col_seg_all = torch.zeros(max(E - edge_cap[-1], 1), **i32)
        self.self_seg_all = torch.arange(R, dtype=torch.int64, device=device)   # input row of an entry itself for layers >= 1
        self.row_ptr = [torch.zeros(r + 1, **i32) for r in row_cap]
        self.self0, self.col, self.col_seg, self.self_seg = [], [], [], []
        rb = eb = 0
        for k, (r, e) in enumerate(zip(row_cap, edge_cap)):
            self.self0.append(self.self0_all[rb:rb + r])
            self.self_seg.append(self.self_seg_all[rb:rb + r])
            self.col.append(self.col_all[eb:eb + e])
            self.col_seg.append(self.col_seg_all[eb:eb + e] if k + 1 < hops else None)
            rb, eb = rb + r, eb + e
        self.n_id = torch.zeros(node_cap, dtype=id_dtype, device=device)
        self.sizes = torch.zeros(2 * hops + 2, **i32)
        self.table = table
        self.x = LazyRows(table, self.n_id) if table is not None else None
        self.seeds = self.n_id[:batch_size]
        self._layers = {}
        self._ptrs = (_ptr_array(self.row_ptr), _ptr_array(self.self0), _ptr_array(self.col), _ptr_array(self.col_seg))
        self._caps = ((ctypes.c_int * hops)(*row_cap), (ctypes.c_int * hops)(*edge_cap))

    @property
    def seed_mask(self):
        """float32 [row_cap[0]]: 1 for the live seed rows of the staged mini-batch, 0 for the padding — written by the staging
        launch.  The last layer's output has row_cap[0] rows: ``nn.cross_entropy(out, labels[batch.n_id[:out.shape[0]]],
        batch.seed_mask)`` is the mean loss over the live seeds without slicing ``out`` (a slice costs a zero-fill and a copy in
        the backward pass of every step)."""
        return self._seed_mask

    @property
    def n_live_seeds(self):
        return self.sizes[0].to(torch.float32)

    def layer_graph(self, layer: int, per_hop: bool = False) -> LayerGraph:
        """The trimmed graph of layer ``layer``: ONE ``HopGraph`` over hops 0..H-1-layer back to back (``per_hop=True``: one
        ``HopGraph`` per hop, the form ``CallGroup.layer_graph`` has — same output rows either way)."""
        H = self.hops
        if not 0 <= layer < H:
            raise IndexError(f"layer {layer} of a {H}-hop mini-batch")
        key = (layer, bool(per_hop))
        if key not in self._layers:
            if per_hop:
                hops = [HopGraph(self.row_ptr[k], self.col[k] if layer == 0 else self.col_seg[k],
                                 self.self0[k] if layer == 0 else self.self_seg[k]) for k in range(H - layer)]
            else:
                R, E = sum(self.row_cap[:H - layer]), sum(self.edge_cap[:H - layer])
                hops = [HopGraph(self.row_ptr_all[:R + 1], (self.col_all if layer == 0 else self.col_seg_all)[:E],
                                 (self.self0_all if layer == 0 else self.self_seg_all)[:R])]
                hops[0].inv_deg = self.inv_deg_all[:R]
            self._layers[key] = LayerGraph(hops)
        return self._layers[key]

    def refilled(self):
        """The buffers hold another mini-batch: drop what eager code cached on the hop objects (transposes, self-loop forms).
        A captured graph recomputes them itself on every replay (wholegraph_amd.nn.begin_capture)."""
        for lg in self._layers.values():
            for h in lg.hops:
                h._t = None
                h._loops = None

    def fits(self, rows, edges, nodes) -> bool:
        return (all(r < c for r, c in zip(rows, self.row_cap)) and all(e <= c for e, c in zip(edges, self.edge_cap))
                and nodes <= self.node_cap)


class PerBatchStep:
    """``step_fn(batch: StagedBatch) -> loss`` captured once, replayed per mini-batch of every call group.

    ``step_fn`` is the whole training step as the user would write it inside ``for batch in loader`` — zero_grad, forward,
    loss, ``loss.backward()``, ``optimizer.step()`` — over ``batch.x`` / ``batch.layer_graph(j)`` / ``batch.seeds``; it must not
    synchronise or read values back (it runs under HIP-graph capture) and every tensor it reads from outside must keep its
    address (labels tables, the model's parameters: true of ``torch.optim`` in-place updates).  Optimizers with host-side step
    counters need their ``capturable=True`` form.  Layers: ``wholegraph_amd.nn.SAGEConv`` (its derived weights are rebuilt inside
    the graph) and anything made of plain torch ops; ``nn.GATConv`` / ``nn.HeteroConv`` refuse capture (their derived-weight
    caches live in Python).

    ``table``: the float32 feature table held whole on this device (then ``batch.x`` is a ``LazyRows``: the first layer reads
    it through ``batch.n_id``).  ``margin``: capacities = the largest mini-batch of the first call group x margin; a later
    mini-batch that does not fit triggers a re-capture with larger buffers.  ``restore``: parameters (and optimizer state)
    to snapshot around the warm-up passes capture needs, so that warm-up does not train."""

    def __init__(self, step_fn: Callable, table: Optional[torch.Tensor] = None, margin: float = 1.2, warmup: int = 2,
                 optimizer: Optional[torch.optim.Optimizer] = None):
        self.step_fn, self.table, self.margin, self.warmup, self.optimizer = step_fn, table, float(margin), int(warmup), optimizer
        self.batch: Optional[StagedBatch] = None
        self._graph = None
        self._loss = None
        self.captures = 0
        self._group_key = None

    # ---- sizes of a group's mini-batches: one small read-back per call group --------------------------------------------------
    def _group_sizes(self, grp):
        key = id(grp)
        if self._group_key is not None and self._group_key[0] == key:
            return self._group_key[1]
        if not hasattr(grp, "_res") or not hasattr(grp._res, "frontier_local0"):
            raise NotImplementedError("PerBatchStep stages the mini-batches of a homogeneous CallGroup (NeighborLoader.call_groups() "
                                      "over a homogeneous GraphStore); got %s" % type(grp).__name__)
        grp._wait()
        res, H = grp._res, grp.hops
        rows, edges = [], []
        for k in range(H):
            seg = res.frontier_seg[k].long()
            rows.append((seg[1:] - seg[:-1]).max())
            off = res.offsets[k][seg].long()
            edges.append((off[1:] - off[:-1]).max())
        nseg = res.node_seg.long()
        v = torch.stack(rows + edges + [(nseg[1:] - nseg[:-1]).max()]).tolist()
        out = (v[:H], v[H:2 * H], v[2 * H])
        self._group_key = (key, out)
        return out

    def _make_buffers(self, grp, rows, edges, nodes):
        def cap(v, floor):
            return max(int(v * self.margin) + 16, floor) // 16 * 16 + 16
        B = int(grp._res.batch_size)
        # (hop 0's frontier = the seeds, <= B; like every hop it gets slack ROWS in proportion to its slack edges, so that a
        #  slack row is no longer than a live one — a few slack rows owning all slack edges are a latency tail in every kernel)
        row_cap = [cap(max(r, B if k == 0 else r), 16) for k, r in enumerate(rows)]
        self.batch = StagedBatch(grp.hops, B, row_cap, [cap(e, 64) for e in edges], cap(nodes, 64), grp._res.nodes.dtype,
                                 grp._res.nodes.device, self.table)
        self._graph = None

    def stage(self, grp, b: int):
        """Copy mini-batch ``b`` of ``grp`` into the fixed buffers (one launch)."""
        res, sb = grp._res, self.batch
        H = grp.hops
        L.check(L.lib().wgamd_call_group_stage_batch(
            H, _ptr_array(res.offsets[:H]), _ptr_array(res.row_local[:H]), _ptr_array(res.frontier_seg[:H]),
            _ptr_array(res.frontier_local0[:H]), res.nodes.data_ptr(), torch_dtype_to_wm(res.nodes.dtype), res.node_seg.data_ptr(),
            int(b), sb._caps[0], sb._caps[1], sb.node_cap, sb._ptrs[0], sb._ptrs[1], sb._ptrs[2], sb._ptrs[3], sb.n_id.data_ptr(),
            sb.sizes.data_ptr(), sb.row_ptr_all.data_ptr(), sb.inv_deg_all.data_ptr(), sb._seed_mask.data_ptr(), get_stream()), "wgamd_call_group_stage_batch")
        sb.refilled()

    def _snapshot(self):
        if self.optimizer is None:
            return None
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        state = {p: {k: v.detach().clone() for k, v in st.items() if torch.is_tensor(v)} for p, st in self.optimizer.state.items()}
        return params, [p.detach().clone() for p in params], state

    def _restore(self, snap):
        """Parameters back to their values before the warm-up passes; optimizer state tensors back to theirs — and the ones
        the warm-up CREATED (momentum buffers, Adam moments and step counts) zeroed in place rather than dropped: the captured
        step must take the optimizer's steady-state code path, and a zero buffer is what its first-step path starts from
        (torch.optim.SGD: buf = grad when dampening is 0; Adam: zero moments, step 0)."""
        if snap is None:
            return
        params, values, state = snap
        with torch.no_grad():
            for p, v in zip(params, values):
                p.copy_(v)
            for p, st in self.optimizer.state.items():
                for k, v in st.items():
                    if torch.is_tensor(v):
                        old = state.get(p, {}).get(k)
                        v.copy_(old) if old is not None else v.zero_()

    def _capture(self):
        """Warm-up passes on a side stream (allocator, library handles, optimizer state), parameters restored afterwards, then
        the capture itself (which executes nothing)."""
        snap = self._snapshot()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self.step_fn(self.batch)
        cur.wait_stream(side)
        self._restore(snap)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        wnn.begin_capture()
        with torch.cuda.graph(g):
            self._loss = self.step_fn(self.batch)
        self._graph = g
        self.captures += 1
        wnn.bump_weight_generation()

    def __call__(self, grp, b: int):
        """Stage mini-batch ``b`` of ``grp`` and run the captured step; returns the step's (static) loss tensor."""
        rows, edges, nodes = self._group_sizes(grp)
        if self.batch is None or not self.batch.fits(rows, edges, nodes):
            self._make_buffers(grp, rows, edges, nodes)
        self.stage(grp, b)
        if self._graph is None:
            self._capture()
        self._graph.replay()
        wnn.bump_weight_generation()
        return self._loss

    def run_group(self, grp):
        """Every mini-batch of the call group in order: ``n_batches`` optimizer steps."""
        loss = None
        for b in range(grp.n_batches):
            loss = self(grp, b)
        return loss
