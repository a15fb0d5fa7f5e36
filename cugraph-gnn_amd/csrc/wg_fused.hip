// No-host-sync hop: uniform sampling + renumbering with device-resident sizes (include/wgamd_ext.h),
// for ONE mini-batch or for a CALL GROUP of G mini-batches processed by the same launches.
//
// The reference's walk (GraphStructure.multilayer_sample_without_replacement,
// /root/reference/python/pylibwholegraph/pylibwholegraph/torch/graph_structure.py:136-196) pays at
// least five stream synchronisations per hop because every op returns an
// exact-size tensor, and its kernels see 1024 seeds at a time — a few hundred workgroups on a
// 256-CU chip.  Here the SAME kernels run with capacity-sized grids and read the live sizes from
// device memory, and a call group (the idea of cugraph_pyg's `local_seeds_per_call`,
// python/cugraph-pyg/cugraph_pyg/sampler/distributed_sampler.py:279-343) gives every launch G x the
// work while each mini-batch keeps exactly the result a single-batch call would produce.
#include <atomic>

#include "wg_common.hpp"
#include "wgamd_ext.h"

namespace wgamd {
namespace {

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct hop_workspace {
  size_t cnt, scan_tmp, nbr, keys, minpos, slot_of, rank, big_list, slab, row_start, row_deg, loc_hist, total;
  int64_t slots, slab_len;
};

// max_row_len > 0: biased hop — also the long-row list and the key slabs of the persistent workgroups
hop_workspace plan(int64_t target_cap, int64_t edge_cap, size_t id_bytes, bool batched, int64_t max_row_len = 0)
{
  hop_workspace w{};
  size_t off = 0;
  auto take  = [&](size_t bytes) {
    size_t at = off;
    off += align_up(bytes);
    return at;
  };
  w.slots        = append_unique_slots(target_cap + edge_cap);
  int64_t scan_n = std::max(target_cap, edge_cap) + 1;
  w.cnt          = take(sizeof(int) * (size_t)(target_cap + 1));
  w.scan_tmp     = take(sizeof(int) * (size_t)scan_tmp_ints(scan_n));
  w.nbr          = take(id_bytes * (size_t)edge_cap);
  w.keys         = take((batched ? 8 : id_bytes) * (size_t)w.slots);
  w.minpos       = take(sizeof(int) * (size_t)w.slots);
  w.slot_of      = take(sizeof(int) * (size_t)(target_cap + edge_cap));
  w.rank         = take(sizeof(int) * (size_t)(edge_cap + 1));
  w.row_start    = take(sizeof(int64_t) * (size_t)target_cap);   // first CSR slot / degree of every sampled row (count -> sample)
  w.row_deg      = take(sizeof(int) * (size_t)target_cap);
  w.loc_hist     = take(sizeof(int) * (size_t)sample_locality_hist_ints());   // vertex-grouped sampling order (2 MB, any capacity)
  if (max_row_len > 0) {
    w.slab_len = max_row_len > kWeightedLdsKeys ? max_row_len : 1;
    w.big_list = take(sizeof(int) * (size_t)weighted_list_ints(target_cap));
    w.slab     = take(sizeof(uint32_t) * (size_t)kWeightedBlocks * (size_t)w.slab_len);
  }
  w.total        = off;
  return w;
}

// smallest frontier capacity whose hop walks the frontier grouped by vertex range (WGAMD_SAMPLE_LOCALITY=<n>: from capacity
// n; wgamd_set_sample_locality_min at run time).  OFF by default: measured on the products hop 2 the grouped order fetches
// 2.7x fewer lines but the sampling kernel is bound by VALU issue, so its duration is unchanged and the two launches that
// build the order (78 us) are a net loss (profiles/r06/README.md)
inline std::atomic<int64_t>& sample_locality_min()
{
  static std::atomic<int64_t> v{[] {
    const char* e = getenv("WGAMD_SAMPLE_LOCALITY");
    return e && atoll(e) > 0 ? (int64_t)atoll(e) : INT64_MAX;
  }()};
  return v;
}

struct hop_args {
  // biased hop: per-CSR-slot weights (FLOAT | DOUBLE) and the graph's maximum degree (sizes the key slabs); null = uniform
  const void* csr_weight  = nullptr;
  bool weight64           = false;
  int64_t max_row_len     = 0;
  // PyG-style walk: the sampled list (frontier) differs from the renumber target list; null = same list
  const void* sample_targets = nullptr;
  const int* n_sample_dev    = nullptr;
  int64_t sample_cap         = 0;
  const int64_t* csr_row_ptr;
  const void* csr_col;
  wholememory_dtype_t id_dtype;
  bool col32 = false;   // WGAMD_HOP_COL_INT32: csr_col holds INT entries although id_dtype is INT64
  const void* targets;
  const int* n_targets_dev;
  int64_t target_cap;
  int M;
  rng_plan rng;
  batch_view bv;
  int* offsets;
  int* neighbor_row;
  int* center_row;
  int64_t* edge_gid;
  int64_t edge_cap;
  void* unique;
  int* counts_dev;
  void* workspace;
  size_t workspace_bytes;
  hipStream_t stream;
};

void run_hop(hop_args a)
{
  WG_REQUIRE_INPUT(a.id_dtype == WHOLEMEMORY_DT_INT || a.id_dtype == WHOLEMEMORY_DT_INT64, "id dtype must be INT|INT64");
  WG_REQUIRE_INPUT(a.csr_row_ptr && a.csr_col && a.targets && a.n_targets_dev && a.offsets && a.neighbor_row &&
                     a.unique && a.workspace,
                   "null pointer");
  WG_REQUIRE_INPUT(a.M > 0, "the no-sync walk needs a positive fan-out (capacity = targets * M)");
  const void* s_targets = a.sample_targets ? a.sample_targets : a.targets;
  const int* s_n_dev    = a.sample_targets ? a.n_sample_dev : a.n_targets_dev;
  const int64_t s_cap   = a.sample_targets ? a.sample_cap : a.target_cap;
  WG_REQUIRE_INPUT(a.target_cap > 0 && s_cap > 0 && a.edge_cap >= s_cap * (int64_t)a.M, "edge_cap < sampled_cap * M");
  WG_REQUIRE_INPUT(a.target_cap + a.edge_cap < ((int64_t)1 << 30), "capacities too large for one call");
  const bool i64     = a.id_dtype == WHOLEMEMORY_DT_INT64;
  const bool col64   = i64 && !a.col32;   // width of the column array = width of the sampled neighbour list
  const bool batched = a.bv.target_batch != nullptr;
  WG_REQUIRE_INPUT(a.csr_weight == nullptr || a.max_row_len > 0, "a biased hop needs max_row_len (the graph's maximum degree)");
  // (the plan is sized for the id width whatever the column width: a caller need not know which one a hop will take)
  hop_workspace w    = plan(std::max(a.target_cap, s_cap), a.edge_cap, i64 ? 8 : 4, batched, a.csr_weight ? a.max_row_len : 0);
  WG_REQUIRE_INPUT(a.workspace_bytes >= w.total, "workspace too small: need %zu bytes", w.total);
  WG_REQUIRE_INPUT((reinterpret_cast<uintptr_t>(a.workspace) & 255) == 0, "workspace must be 256-byte aligned");
  char* base    = static_cast<char*>(a.workspace);
  int* cnt      = reinterpret_cast<int*>(base + w.cnt);
  int* scan_tmp = reinterpret_cast<int*>(base + w.scan_tmp);
  void* nbr     = base + w.nbr;
  void* keys    = base + w.keys;
  int* minpos   = reinterpret_cast<int*>(base + w.minpos);
  int* slot_of  = reinterpret_cast<int*>(base + w.slot_of);
  int* rank     = reinterpret_cast<int*>(base + w.rank);
  hipStream_t st = a.stream;

  dev_count T{(int)a.target_cap, a.n_targets_dev};
  dev_count S{(int)s_cap, s_n_dev};
  if (a.csr_weight != nullptr) {
    int* big_list = reinterpret_cast<int*>(base + w.big_list);
    weighted_count_enqueue(a.csr_row_ptr, s_targets, i64, S, a.M, cnt, big_list, st);
    exclusive_scan_i32(cnt, a.offsets, s_cap, scan_tmp, st, s_n_dev);  // offsets[cap] = #edges
    weighted_sample_enqueue(a.csr_row_ptr, a.csr_col, col64, a.csr_weight, a.weight64, s_targets, i64, S, a.M, a.rng, a.offsets,
                            big_list, reinterpret_cast<uint32_t*>(base + w.slab), w.slab_len, nbr, a.center_row, a.edge_gid,
                            st);
  } else {
    int64_t* row_start = reinterpret_cast<int64_t*>(base + w.row_start);
    int* row_deg       = reinterpret_cast<int*>(base + w.row_deg);
    sample_count_enqueue(a.csr_row_ptr, s_targets, i64, S, a.M, cnt, nullptr, st, row_start, row_deg);
    exclusive_scan_i32(cnt, a.offsets, s_cap, scan_tmp, st, s_n_dev);  // offsets[cap] = #edges
    // Long hops of a call group walk the frontier grouped by vertex range (wg_sample.hip, loc_rec): same bits, a third of
    // the line fetches where the frontier repeats its hubs batch after batch.  The records
    // live in the renumber scratch, which nothing uses before the sampled neighbours exist (keys: 16 B per position of
    // capacity); the histogram matrix has its own 2 MB.  WGAMD_SAMPLE_LOCALITY=0 keeps list order; =<n> sets the smallest capacity it applies to.
    const int64_t loc_min = sample_locality_min().load(std::memory_order_relaxed);
    sample_locality loc{reinterpret_cast<int*>(base + w.loc_hist), keys, sample_locality_shift(a.bv.id_bound)};
    const bool use_loc = a.M <= 32 && s_cap >= loc_min && a.bv.id_bound > 0 && batched && 8 * w.slots >= 32 * s_cap;
    uniform_sample_enqueue(a.csr_row_ptr, a.csr_col, col64, s_targets, i64, S, a.M, a.rng, a.offsets, nbr, a.center_row,
                           a.edge_gid, st, row_start, row_deg, use_loc ? &loc : nullptr);
  }
  dev_count E{(int)a.edge_cap, a.offsets + s_cap};
  batch_view bv   = a.bv;
  bv.edge_row     = a.center_row;
  bv.edge_offsets = a.offsets;
  append_unique_prepare_enqueue(a.targets, T, i64, nbr, E, col64, bv, keys, minpos, w.slots, slot_of, rank, scan_tmp, st);
  append_unique_emit_enqueue(a.targets, T, i64, nbr, E, col64, bv, minpos, slot_of, rank, w.slots, a.unique, a.neighbor_row,
                             a.counts_dev, st);
}

__global__ void __launch_bounds__(256)
target_rows_kernel(const int* __restrict__ unique_seg, const int* __restrict__ target_seg, const int* __restrict__ target_batch,
                   int n, int64_t* __restrict__ rows)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = target_batch[i];
  rows[i]     = (int64_t)i + unique_seg[b] - target_seg[b];
}

// One (hop, edge type) of a HETEROGENEOUS call group, as the layers consume it: where the hop's frontier entries sit in the
// destination type's node list (batch-major lists: row = segment start of the batch + local id) and the source rows of its
// edges in the source type's list — in the FULL numbering (all vertices of the walk) and in the COMPACT one (the vertices
// discovered by hops 0-1 only: what layer 1 has to produce and layer 2 reads).  One 16-lane group per frontier entry walks
// the entry's edges; replaces a dozen torch index ops per hop and edge type.
__global__ void __launch_bounds__(256)
hetero_hop_rows_kernel(const int* __restrict__ offsets, const int* __restrict__ f_batch, const int* __restrict__ f_seg,
                       const int* __restrict__ f_local0, const int* __restrict__ row_local, int n_f,
                       const int* __restrict__ seg_dst, const int64_t* __restrict__ cseg_dst, const int* __restrict__ seg_src,
                       const int64_t* __restrict__ cseg_src, int64_t* __restrict__ dst_full, int64_t* __restrict__ dst_compact,
                       int* __restrict__ col_full, int* __restrict__ col_compact)
{
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t j   = tid >> 4;
  const int sub     = (int)(tid & 15);
  if (j >= n_f) return;
  const int b     = f_batch[j];
  const int local = f_local0[b] + ((int)j - f_seg[b]);
  if (sub == 0) {
    dst_full[j] = (int64_t)seg_dst[b] + local;
    if (dst_compact) dst_compact[j] = cseg_dst[b] + local;
  }
  const int s = offsets[j], e = offsets[j + 1];
  const int add_full = seg_src[b];
  const int add_c    = col_compact ? (int)cseg_src[b] : 0;
  for (int i = s + sub; i < e; i += 16) {
    const int r = row_local[i];
    col_full[i] = r + add_full;
    if (col_compact) col_compact[i] = r + add_c;
  }
}

// The same as hetero_hop_rows_kernel with the number of batches known: block (batch, chunk) reads the batch's four segment
// starts once and streams its stretch of frontier entries and edges with every lane (see layer_cols_kernel below).
__global__ void __launch_bounds__(256)
hetero_hop_rows_batched_kernel(const int* __restrict__ offsets, const int* __restrict__ f_seg, const int* __restrict__ f_local0,
                               const int* __restrict__ row_local, int chunks, const int* __restrict__ seg_dst,
                               const int64_t* __restrict__ cseg_dst, const int* __restrict__ seg_src,
                               const int64_t* __restrict__ cseg_src, int64_t* __restrict__ dst_full,
                               int64_t* __restrict__ dst_compact, int* __restrict__ col_full, int* __restrict__ col_compact)
{
  const int b = blockIdx.x / chunks, c = blockIdx.x % chunks, t = threadIdx.x;
  const int f0 = f_seg[b], f1 = f_seg[b + 1];
  const int j0 = f0 + (int)((int64_t)(f1 - f0) * c / chunks), j1 = f0 + (int)((int64_t)(f1 - f0) * (c + 1) / chunks);
  if (j0 >= j1) return;
  const int e0 = offsets[j0], e1 = offsets[j1];
  const int64_t d_full = (int64_t)seg_dst[b] + f_local0[b] - f0;
  const int64_t d_c    = dst_compact ? cseg_dst[b] + f_local0[b] - f0 : 0;
  const int add_full   = seg_src[b];
  const int add_c      = col_compact ? (int)cseg_src[b] : 0;
  for (int j = j0 + t; j < j1; j += 256) {
    dst_full[j] = d_full + j;
    if (dst_compact) dst_compact[j] = d_c + j;
  }
  for (int i = e0 + t; i < e1; i += 256) {
    const int r = row_local[i];
    col_full[i] = r + add_full;
    if (col_compact) col_compact[i] = r + add_c;
  }
}

// One hop of a homogeneous (or one edge type of a) PyG-style call group, renumbered for the LAYER that consumes it.  The
// layer's input rows are laid out as `n_seg` segments per batch: local ids [local0[s][b], local0[s+1][b]) of batch b sit at
// rows base[s] + start[s][b] + (local id - local0[s][b]) — one segment (start = the node-list offsets) is the batch-major
// list of ALL vertices (x = feat[n_id]); the output of a trimmed layer is one segment per hop it ran (the hop's frontier
// list, batch-major).  For frontier entry j of batch b the kernel writes the input row of the entry itself (self_rows[j]) and
// of every one of its sampled neighbours (col[e]).  seg_tab: int32 [2 * n_seg, G + 1], row 2 s = local0[s], row 2 s + 1 =
// start[s].  Block (batch, chunk): the frontier entries of a batch are consecutive and so are their edges, so a block reads
// what depends on the batch ONCE (segment table -> LDS as (first local id, row shift) pairs) and then streams its stretch of
// `row_local` -> `col` with every lane busy — one 16-lane group per frontier entry chased five dependent loads per edge with
// 10 of 16 lanes (196 us for the 15 M edges of a products call group's second hop; the bytes are worth 17 us).
constexpr int kLayerSegMax = 16;
__global__ void __launch_bounds__(256)
layer_cols_kernel(const int* __restrict__ offsets, const int* __restrict__ f_seg, const int* __restrict__ f_local0,
                  const int* __restrict__ row_local, int G, int chunks, int n_seg, const int* __restrict__ seg_tab,
                  const int64_t* __restrict__ seg_base, int64_t* __restrict__ self_rows, int* __restrict__ col)
{
  __shared__ int s_l0[kLayerSegMax];
  __shared__ int64_t s_add[kLayerSegMax];
  const int b = blockIdx.x / chunks, c = blockIdx.x % chunks;
  const int t = threadIdx.x;
  if (t < n_seg) {
    const int l0 = seg_tab[(int64_t)(2 * t) * (G + 1) + b];
    s_l0[t]      = l0;
    s_add[t]     = seg_base[t] + seg_tab[(int64_t)(2 * t + 1) * (G + 1) + b] - l0;
  }
  const int f0 = f_seg[b], f1 = f_seg[b + 1];
  const int j0 = f0 + (int)((int64_t)(f1 - f0) * c / chunks), j1 = f0 + (int)((int64_t)(f1 - f0) * (c + 1) / chunks);
  if (j0 >= j1) return;   // (uniform over the block, before the barrier)
  const int e0 = offsets[j0], e1 = offsets[j1];
  const int first_local = f_local0[b] - f0;
  __syncthreads();
  auto row_of = [&](int local) -> int64_t {
    int s = 0;
    for (int k = 1; k < n_seg; k++) s = s_l0[k] <= local ? k : s;
    return (int64_t)local + s_add[s];
  };
  if (self_rows)
    for (int j = j0 + t; j < j1; j += 256) self_rows[j] = row_of(first_local + j);
  int i = e0 + t;
  for (; i + 768 < e1; i += 1024) {   // four independent loads in flight per lane
    const int r0 = row_local[i], r1 = row_local[i + 256], r2 = row_local[i + 512], r3 = row_local[i + 768];
    col[i]       = (int)row_of(r0);
    col[i + 256] = (int)row_of(r1);
    col[i + 512] = (int)row_of(r2);
    col[i + 768] = (int)row_of(r3);
  }
  for (; i < e1; i += 256) col[i] = (int)row_of(row_local[i]);
}

// Frontier of a node type between two hops of the heterogeneous call-group walk (HeteroPygWalk._frontier): batch b gained the
// vertices [begin[b], seg[b+1] - seg[b]) of its list since the previous hop; they are laid out batch-major in `ids` with their
// batch and the per-batch offsets f_seg.  One launch instead of thirteen framework ops per node type and hop; every workgroup
// rebuilds the (short) offset table in LDS and binary-searches it.  Slots past the live total are padding with the same
// (clamped) values the framework formulation produced.
constexpr int kFrontierMaxG = 4095;
__global__ void __launch_bounds__(256)
frontier_list_kernel(const int64_t* __restrict__ nodes, int64_t n_nodes, const int* __restrict__ seg, const int* __restrict__ begin,
                     int G, int64_t cap, int64_t* __restrict__ ids, int* __restrict__ batch, int* __restrict__ f_seg)
{
  __shared__ int s_seg[kFrontierMaxG + 1];
  if (threadIdx.x < 64) {   // one wave scans the G counts, 64 at a time
    int carry = 0;
    for (int b0 = 0; b0 < G; b0 += 64) {
      const int b = b0 + (int)threadIdx.x;
      int v       = b < G ? (seg[b + 1] - seg[b]) - begin[b] : 0;
      int incl    = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if ((int)threadIdx.x >= d) incl += o;
      }
      if (b < G) s_seg[b] = carry + incl - v;
      carry += __shfl(incl, 63, 64);
    }
    if (threadIdx.x == 0) s_seg[G] = carry;
  }
  __syncthreads();
  if (blockIdx.x == 0)
    for (int b = threadIdx.x; b <= G; b += blockDim.x) f_seg[b] = s_seg[b];
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < cap; p += (int64_t)gridDim.x * blockDim.x) {
    int lo = 0, hi = G - 1;   // largest b < G with s_seg[b] <= p (s_seg[0] = 0)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if ((int64_t)s_seg[mid] <= p) lo = mid; else hi = mid - 1;
    }
    int64_t src = (int64_t)seg[lo] + begin[lo] + (p - s_seg[lo]);
    src         = src < 0 ? 0 : (src >= n_nodes ? n_nodes - 1 : src);
    ids[p]      = nodes[src];
    batch[p]    = lo;
  }
}

// ---- one mini-batch of a PyG-style call group, staged into FIXED-SIZE buffers ------------------------------------------------
// The reference's training loops step the optimizer once per mini-batch (pylibwholegraph/torch/gnn_model.py:119-125,
// cugraph_pyg/examples/gcn_dist_mnmg.py).  A call group is walked once; its mini-batches are then consecutive slices of every
// array — frontier entries [frontier_seg[b], frontier_seg[b + 1]) of each hop and their edges, vertices [node_seg[b],
// node_seg[b + 1]) — in BATCH-LOCAL ids.  This kernel copies one mini-batch into buffers whose sizes and addresses never
// change, padded to their capacities (padded frontier rows have no edges, padded vertices repeat a valid id), so the whole
// per-batch training step — forward, loss, backward, optimizer — can be captured in ONE hipGraph and replayed per mini-batch:
// one staging launch and one graph launch instead of ~30 kernel launches and the host work behind them.
// Layout of a trimmed layer's OUTPUT rows (and so of the next layer's input): hop 0's row_cap[0] rows, then hop 1's
// row_cap[1], ...: `col_seg` holds the sources as rows of that layout, `col` as batch-local ids (= rows of x for layer 0).
constexpr int kStageMaxHops = 8;
struct stage_hop {
  const int* offsets;
  const int* row_local;
  const int* f_seg;
  const int* f_local0;
  int row_cap, edge_cap;
  int* row_ptr;       // [row_cap + 1]
  int64_t* self0;     // [row_cap]: input row of the entry itself for layer 0 (its batch-local id)
  int* col;           // [edge_cap]
  int* col_seg;       // [edge_cap] (nullable)
};
struct stage_args {
  stage_hop hop[kStageMaxHops];
  int n_hops, batch, node_cap;
  const void* nodes;
  const int* node_seg;
  void* n_id;         // [node_cap]
  int* sizes;         // [2 n_hops + 2]: rows and edges per hop, vertices, overflow flag
  float* inv_deg_all; // [sum row_cap] (nullable): 1 / max(degree, 1) of every row of that CSR (the mean's backward scales by it)
  float* seed_mask;   // [hop[0].row_cap] (nullable): 1 for a live seed row, 0 for the padding (the weight of a row in the loss)
  int* row_ptr_all;   // [sum row_cap + 1] (nullable): the hops' CSRs back to back as ONE CSR — hop k's rows start at
                      // sum(row_cap[:k]), its edges at sum(edge_cap[:k]).  With the hops' col / self arrays allocated back to
                      // back by the caller, a layer over hops 0..j is one launch over a prefix of it instead of j + 1 launches
};

template <typename IdT>
__global__ void __launch_bounds__(256) stage_batch_kernel(const stage_args a)
{
  __shared__ int s_l0[kStageMaxHops + 1];     // first local id of every hop's frontier (+ the vertex count)
  __shared__ int s_base[kStageMaxHops + 1];   // first row of every hop's segment in a layer's output layout
  __shared__ int s_ebase[kStageMaxHops + 1];  // first edge of every hop in the back-to-back CSR
  const int b = a.batch, t = threadIdx.x;
  const int n0 = a.node_seg[b], n_nodes = a.node_seg[b + 1] - n0;
  if (t == 0) {
    int base = 0, ebase = 0;
    for (int k = 0; k < a.n_hops; k++) {
      s_l0[k]    = a.hop[k].f_local0[b];
      s_base[k]  = base;
      s_ebase[k] = ebase;
      base += a.hop[k].row_cap;
      ebase += a.hop[k].edge_cap;
    }
    s_l0[a.n_hops] = n_nodes;
  }
  __syncthreads();
  const int gtid = blockIdx.x * blockDim.x + t, gsize = gridDim.x * blockDim.x;
  int over = n_nodes > a.node_cap;
  // vertices (global ids, the layer-0 kernel reads the feature table through them); the padding repeats the first one
  const IdT* nodes = static_cast<const IdT*>(a.nodes) + n0;
  IdT* out_ids     = static_cast<IdT*>(a.n_id);
  for (int i = gtid; i < a.node_cap; i += gsize) out_ids[i] = nodes[i < n_nodes ? i : 0];
  for (int k = 0; k < a.n_hops; k++) {
    const stage_hop h = a.hop[k];
    const int j0 = h.f_seg[b], n_rows = h.f_seg[b + 1] - j0;
    const int e0 = h.offsets[j0], n_edges = h.offsets[j0 + n_rows] - e0;
    // (one padded row at least: the padding edges below must belong to rows that are not the mini-batch's)
    over |= (n_rows >= h.row_cap) | (n_edges > h.edge_cap);
    const int rows = min(n_rows, h.row_cap - 1), edges = min(n_edges, h.edge_cap);
    if (gtid == 0 && a.sizes) {
      a.sizes[2 * k]     = n_rows;
      a.sizes[2 * k + 1] = n_edges;
    }
    // Every entry of the fixed-size arrays must be a well-formed edge (kernels take the edge count from the array, and so does
    // the transpose of the backward pass): the slack edges are dealt evenly to the slack rows, with sources spread over the
    // input rows — no long row, no hub source.  Nothing reads a slack row's output and its gradient is zero.
    const int64_t pad_rows = h.row_cap - rows, pad_edges = h.edge_cap - edges;
    for (int r = gtid; r <= h.row_cap; r += gsize) {
      auto at_of = [&](int q) { return q <= rows ? min(h.offsets[j0 + q] - e0, edges) : edges + (int)((int64_t)(q - rows) * pad_edges / pad_rows); };
      const int at = at_of(r);
      h.row_ptr[r] = at;
      if (a.row_ptr_all) a.row_ptr_all[s_base[k] + r] = s_ebase[k] + at;   // (r = row_cap: the next hop's first entry, same value)
      if (a.inv_deg_all && r < h.row_cap) a.inv_deg_all[s_base[k] + r] = 1.f / (float)max(at_of(r + 1) - at, 1);
      if (k == 0 && a.seed_mask && r < h.row_cap) a.seed_mask[r] = r < rows ? 1.f : 0.f;
      if (r < h.row_cap) h.self0[r] = r < rows ? (int64_t)(s_l0[k] + r) : 0;
    }
    for (int e = gtid; e < h.edge_cap; e += gsize) {
      if (e < edges) {
        const int local = h.row_local[e0 + e];
        h.col[e]        = local;
        if (h.col_seg) {
          int s = 0;
          for (int q = 1; q < a.n_hops; q++) s = s_l0[q] <= local ? q : s;
          h.col_seg[e] = s_base[s] + (local - s_l0[s]);
        }
      } else {
        h.col[e] = (e - edges) % max(min(n_nodes, a.node_cap), 1);
        if (h.col_seg) h.col_seg[e] = (e - edges) % a.hop[0].row_cap;
      }
    }
  }
  if (gtid == 0 && a.sizes) {
    a.sizes[2 * a.n_hops]     = n_nodes;
    a.sizes[2 * a.n_hops + 1] = over;
  }
}

}  // namespace
}  // namespace wgamd

extern "C" {

wholememory_error_code_t wgamd_call_group_stage_batch(int n_hops, const int* const* offsets, const int* const* row_local,
                                                      const int* const* frontier_seg, const int* const* frontier_local0,
                                                      const void* nodes, wholememory_dtype_t id_dtype, const int* node_seg,
                                                      int batch, const int* row_cap, const int* edge_cap, int node_cap,
                                                      int* const* row_ptr_out, int64_t* const* self_rows_out, int* const* col_out,
                                                      int* const* col_seg_out, void* n_id_out, int* sizes_out, int* row_ptr_all_out,
                                                      float* inv_deg_all_out, float* seed_mask_out, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_call_group_stage_batch", [&] {
    WG_REQUIRE_INPUT(n_hops >= 1 && n_hops <= kStageMaxHops, "1 <= n_hops <= 8");
    WG_REQUIRE_INPUT(id_dtype == WHOLEMEMORY_DT_INT || id_dtype == WHOLEMEMORY_DT_INT64, "ids must be INT or INT64");
    WG_REQUIRE_INPUT(offsets && row_local && frontier_seg && frontier_local0 && nodes && node_seg && row_cap && edge_cap &&
                       row_ptr_out && self_rows_out && col_out && n_id_out && batch >= 0 && node_cap >= 1,
                     "null pointer / bad sizes");
    stage_args a{};
    int64_t work = node_cap;
    for (int k = 0; k < n_hops; k++) {
      WG_REQUIRE_INPUT(offsets[k] && row_local[k] && frontier_seg[k] && frontier_local0[k] && row_ptr_out[k] && self_rows_out[k] &&
                         col_out[k] && row_cap[k] >= 1 && edge_cap[k] >= 1,
                       "hop %d: null pointer / empty capacity", k);
      a.hop[k] = stage_hop{offsets[k], row_local[k], frontier_seg[k], frontier_local0[k], row_cap[k], edge_cap[k], row_ptr_out[k],
                           self_rows_out[k], col_out[k], col_seg_out ? col_seg_out[k] : nullptr};
      work     = std::max<int64_t>(work, std::max(row_cap[k], edge_cap[k]));
    }
    a.n_hops = n_hops, a.batch = batch, a.node_cap = node_cap, a.nodes = nodes, a.node_seg = node_seg, a.n_id = n_id_out;
    a.sizes  = sizes_out, a.row_ptr_all = row_ptr_all_out, a.inv_deg_all = inv_deg_all_out, a.seed_mask = seed_mask_out;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(work, (int64_t)1024), 256));
    if (id_dtype == WHOLEMEMORY_DT_INT64)
      stage_batch_kernel<int64_t><<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(a);
    else
      stage_batch_kernel<int32_t><<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(a);
    WG_HIP_CHECK(hipGetLastError());
  });
}

void wgamd_set_sample_locality_min(int64_t min_capacity)
{
  wgamd::sample_locality_min().store(min_capacity > 0 ? min_capacity : INT64_MAX, std::memory_order_relaxed);
}

wholememory_error_code_t wgamd_frontier_list(const int64_t* nodes, int64_t n_nodes, const int* seg, const int* begin, int n_batches,
                                             int64_t capacity, int64_t* ids, int* batch, int* f_seg, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_frontier_list", [&] {
    WG_REQUIRE_INPUT(n_batches >= 1 && n_batches <= kFrontierMaxG, "1 <= n_batches <= 4095");
    WG_REQUIRE_INPUT(capacity >= 0 && n_nodes >= 1, "bad capacity / empty node list");
    WG_REQUIRE_INPUT(nodes && seg && begin && f_seg && (capacity == 0 || (ids && batch)), "null pointer");
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((capacity + 255) / 256, 256 * 8));
    frontier_list_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(nodes, n_nodes, seg, begin, n_batches, capacity, ids,
                                                                              batch, f_seg);
    WG_HIP_CHECK(hipGetLastError());
  });
}

wholememory_error_code_t wgamd_call_group_layer_cols(const int* offsets, const int* frontier_batch, const int* frontier_seg,
                                                     const int* frontier_local0, const int* row_local, int64_t n_frontier,
                                                     int n_batches, int n_segments, const int* seg_tab, const int64_t* seg_base,
                                                     int64_t* self_rows, int* col, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_call_group_layer_cols", [&] {
    WG_REQUIRE_INPUT(n_frontier >= 0 && n_frontier < ((int64_t)1 << 27), "bad frontier count");
    WG_REQUIRE_INPUT(n_batches >= 1 && n_batches < (1 << 20) && n_segments >= 1 && n_segments <= kLayerSegMax, "bad batch / segment count");
    if (n_frontier == 0) return;
    WG_REQUIRE_INPUT(offsets && frontier_seg && frontier_local0 && row_local && seg_tab && seg_base && col, "null pointer");
    (void)frontier_batch;   // (the batch of a block's entries follows from frontier_seg)
    // ~512 frontier entries (and their edges) per block: the block's prelude is three dependent loads
    const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(256, ceil_div(ceil_div(n_frontier, (int64_t)n_batches), 512)));
    layer_cols_kernel<<<n_batches * chunks, 256, 0, static_cast<hipStream_t>(stream)>>>(
      offsets, frontier_seg, frontier_local0, row_local, n_batches, chunks, n_segments, seg_tab, seg_base, self_rows, col);
    WG_HIP_CHECK(hipGetLastError());
  });
}

size_t wgamd_sample_hop_weighted_workspace_bytes(int64_t target_cap, int64_t edge_cap, wholememory_dtype_t id_dtype,
                                                 int64_t max_row_len)
{
  if (target_cap < 0 || edge_cap < 0 || max_row_len <= 0) return 0;
  size_t idb = id_dtype == WHOLEMEMORY_DT_INT64 ? 8 : 4;
  return wgamd::plan(target_cap, edge_cap, idb, true, max_row_len).total;
}

size_t wgamd_sample_hop_workspace_bytes(int64_t target_cap, int64_t edge_cap, wholememory_dtype_t id_dtype)
{
  if (target_cap < 0 || edge_cap < 0) return 0;
  size_t idb = id_dtype == WHOLEMEMORY_DT_INT64 ? 8 : 4;
  return wgamd::plan(target_cap, edge_cap, idb, true).total;  // sized for the batched table (int64 keys)
}

wholememory_error_code_t wgamd_sample_hop_nosync(const int64_t* csr_row_ptr, const void* csr_col,
                                                 wholememory_dtype_t id_dtype, const void* targets,
                                                 const int* n_targets_dev, int64_t target_cap, int max_sample_count,
                                                 unsigned long long random_seed, int* offsets, int* neighbor_lid,
                                                 int* center_lid, int64_t* edge_gid, int64_t edge_cap, void* unique,
                                                 int* counts_dev, void* workspace, size_t workspace_bytes, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_sample_hop_nosync", [&] {
    WG_REQUIRE_INPUT(counts_dev != nullptr, "counts_dev is NULL");
    hop_args a{};
    a.csr_row_ptr = csr_row_ptr; a.csr_col = csr_col; a.id_dtype = id_dtype; a.targets = targets;
    a.n_targets_dev = n_targets_dev; a.target_cap = target_cap; a.M = max_sample_count;
    a.rng = rng_plan{(uint64_t)random_seed, nullptr, nullptr, nullptr};
    a.bv.G = 1;
    a.offsets = offsets; a.neighbor_row = neighbor_lid; a.center_row = center_lid; a.edge_gid = edge_gid;
    a.edge_cap = edge_cap; a.unique = unique; a.counts_dev = counts_dev; a.workspace = workspace;
    a.workspace_bytes = workspace_bytes; a.stream = static_cast<hipStream_t>(stream);
    run_hop(a);
  });
}

wholememory_error_code_t wgamd_sample_hop_batched_nosync(
  const int64_t* csr_row_ptr, const void* csr_col, wholememory_dtype_t id_dtype, const void* targets,
  const int* target_batch, const int* target_seg, int n_batches, int64_t target_cap, int max_sample_count,
  const unsigned long long* random_seeds_dev, int* offsets, int* neighbor_row, int* center_row, int64_t* edge_gid,
  int64_t edge_cap, void* unique, int* unique_batch, int* unique_seg, int* counts_dev, void* workspace,
  size_t workspace_bytes, int64_t n_vertices, void* stream)
{
  return wgamd_sample_hop_batched_nosync_ex(csr_row_ptr, csr_col, id_dtype, targets, target_batch, target_seg, n_batches,
                                            target_cap, max_sample_count, random_seeds_dev, offsets, neighbor_row, center_row,
                                            edge_gid, edge_cap, unique, unique_batch, unique_seg, counts_dev, workspace,
                                            workspace_bytes, n_vertices, 0u, stream);
}

wholememory_error_code_t wgamd_sample_hop_batched_nosync_ex(
  const int64_t* csr_row_ptr, const void* csr_col, wholememory_dtype_t id_dtype, const void* targets,
  const int* target_batch, const int* target_seg, int n_batches, int64_t target_cap, int max_sample_count,
  const unsigned long long* random_seeds_dev, int* offsets, int* neighbor_row, int* center_row, int64_t* edge_gid,
  int64_t edge_cap, void* unique, int* unique_batch, int* unique_seg, int* counts_dev, void* workspace,
  size_t workspace_bytes, int64_t n_vertices, unsigned flags, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_sample_hop_batched_nosync", [&] {
    WG_REQUIRE_INPUT((flags & ~(WGAMD_HOP_NO_UNIQUE_PAD | WGAMD_HOP_COL_INT32)) == 0, "unknown flag bits");
    WG_REQUIRE_INPUT(!(flags & WGAMD_HOP_COL_INT32) || (id_dtype == WHOLEMEMORY_DT_INT64 && n_vertices > 0 &&
                                                        n_vertices < ((int64_t)1 << 31)),
                     "WGAMD_HOP_COL_INT32 needs INT64 ids and 0 < n_vertices < 2^31");
    // unique_batch may be NULL: the batch of every unique entry is then not produced (the last hop of a walk)
    WG_REQUIRE_INPUT(target_batch && target_seg && random_seeds_dev && center_row && unique_seg && counts_dev,
                     "null pointer");
    WG_REQUIRE_INPUT(n_batches >= 1 && n_batches < (1 << 20), "bad batch count");
    hop_args a{};
    a.csr_row_ptr = csr_row_ptr; a.csr_col = csr_col; a.id_dtype = id_dtype; a.targets = targets;
    a.n_targets_dev = target_seg + n_batches;  // the last segment boundary IS the live target count
    a.target_cap = target_cap; a.M = max_sample_count;
    a.rng = rng_plan{0, random_seeds_dev, target_batch, target_seg};
    a.bv.target_batch = target_batch; a.bv.target_seg = target_seg; a.bv.G = n_batches;
    a.bv.id_bound = n_vertices > 0 ? n_vertices : 0;
    a.bv.unique_batch = unique_batch; a.bv.unique_seg = unique_seg;
    a.bv.no_pad = (flags & WGAMD_HOP_NO_UNIQUE_PAD) ? 1 : 0;
    a.col32     = (flags & WGAMD_HOP_COL_INT32) != 0;
    a.offsets = offsets; a.neighbor_row = neighbor_row; a.center_row = center_row; a.edge_gid = edge_gid;
    a.edge_cap = edge_cap; a.unique = unique; a.counts_dev = counts_dev; a.workspace = workspace;
    a.workspace_bytes = workspace_bytes; a.stream = static_cast<hipStream_t>(stream);
    run_hop(a);
  });
}

wholememory_error_code_t wgamd_call_group_target_rows(const int* unique_seg, const int* target_seg, const int* target_batch,
                                                      int64_t n_targets, int64_t* rows, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_call_group_target_rows", [&] {
    WG_REQUIRE_INPUT(n_targets >= 0 && n_targets < ((int64_t)1 << 31), "bad target count");
    if (n_targets == 0) return;
    WG_REQUIRE_INPUT(unique_seg && target_seg && target_batch && rows, "null pointer");
    target_rows_kernel<<<ceil_div(n_targets, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(unique_seg, target_seg, target_batch,
                                                                                              (int)n_targets, rows);
    WG_HIP_CHECK(hipGetLastError());
  });
}

wholememory_error_code_t wgamd_call_group_hop_rows(const int* offsets, const int* frontier_batch, const int* frontier_seg,
                                                   const int* frontier_local0, const int* row_local, int64_t n_frontier,
                                                   const int* seg_dst, const int64_t* compact_seg_dst, const int* seg_src,
                                                   const int64_t* compact_seg_src, int64_t* dst_full, int64_t* dst_compact,
                                                   int* col_full, int* col_compact, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_call_group_hop_rows", [&] {
    WG_REQUIRE_INPUT(n_frontier >= 0 && n_frontier < ((int64_t)1 << 27), "bad frontier count");
    if (n_frontier == 0) return;
    WG_REQUIRE_INPUT(offsets && frontier_batch && frontier_seg && frontier_local0 && row_local && seg_dst && seg_src && dst_full &&
                       col_full,
                     "null pointer");
    WG_REQUIRE_INPUT((dst_compact == nullptr) == (compact_seg_dst == nullptr) && (col_compact == nullptr) == (compact_seg_src == nullptr),
                     "a compact output needs its compact segment array (and the other way round)");
    hetero_hop_rows_kernel<<<ceil_div(n_frontier * 16, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      offsets, frontier_batch, frontier_seg, frontier_local0, row_local, (int)n_frontier, seg_dst, compact_seg_dst, seg_src,
      compact_seg_src, dst_full, dst_compact, col_full, col_compact);
    WG_HIP_CHECK(hipGetLastError());
  });
}

wholememory_error_code_t wgamd_call_group_hop_rows_batched(const int* offsets, const int* frontier_seg, const int* frontier_local0,
                                                           const int* row_local, int64_t n_frontier, int n_batches,
                                                           const int* seg_dst, const int64_t* compact_seg_dst, const int* seg_src,
                                                           const int64_t* compact_seg_src, int64_t* dst_full, int64_t* dst_compact,
                                                           int* col_full, int* col_compact, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_call_group_hop_rows_batched", [&] {
    WG_REQUIRE_INPUT(n_frontier >= 0 && n_frontier < ((int64_t)1 << 27) && n_batches >= 1 && n_batches < (1 << 20), "bad counts");
    if (n_frontier == 0) return;
    WG_REQUIRE_INPUT(offsets && frontier_seg && frontier_local0 && row_local && seg_dst && seg_src && dst_full && col_full,
                     "null pointer");
    WG_REQUIRE_INPUT((dst_compact == nullptr) == (compact_seg_dst == nullptr) && (col_compact == nullptr) == (compact_seg_src == nullptr),
                     "a compact output needs its compact segment array (and the other way round)");
    const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(256, ceil_div(ceil_div(n_frontier, (int64_t)n_batches), 512)));
    hetero_hop_rows_batched_kernel<<<n_batches * chunks, 256, 0, static_cast<hipStream_t>(stream)>>>(
      offsets, frontier_seg, frontier_local0, row_local, chunks, seg_dst, compact_seg_dst, seg_src, compact_seg_src, dst_full,
      dst_compact, col_full, col_compact);
    WG_HIP_CHECK(hipGetLastError());
  });
}

wholememory_error_code_t wgamd_sample_hop_pyg_nosync(const wgamd_pyg_hop_t* p, void* stream)
{
  using namespace wgamd;
  return guarded("wgamd_sample_hop_pyg_nosync", [&] {
    WG_REQUIRE_INPUT(p != nullptr, "null parameter block");
    WG_REQUIRE_INPUT(p->nodes && p->node_batch && p->node_seg && p->frontier && p->frontier_batch && p->frontier_seg &&
                       p->frontier_local0 && p->random_seeds_dev && p->center_local && p->neighbor_local &&
                       p->nodes_out && p->nodes_out_batch && p->nodes_out_seg && p->frontier_out &&
                       p->frontier_out_batch && p->frontier_out_seg && p->frontier_out_local0 && p->counts_dev &&
                       p->center_row_scratch,
                     "null pointer");
    WG_REQUIRE_INPUT(p->n_batches >= 1 && p->n_batches < (1 << 20), "bad batch count");
    hop_args a{};
    a.csr_row_ptr = p->csr_row_ptr; a.csr_col = p->csr_col; a.id_dtype = p->id_dtype;
    a.targets = p->nodes; a.n_targets_dev = p->node_seg + p->n_batches; a.target_cap = p->node_cap;
    a.sample_targets = p->frontier; a.n_sample_dev = p->frontier_seg + p->n_batches; a.sample_cap = p->frontier_cap;
    a.M = p->max_sample_count;
    a.rng = rng_plan{0, p->random_seeds_dev, p->frontier_batch, p->frontier_seg};
    a.bv.target_batch = p->node_batch; a.bv.target_seg = p->node_seg; a.bv.G = p->n_batches;
    a.bv.id_bound = p->n_vertices > 0 ? p->n_vertices : 0;
    a.bv.sample_batch = p->frontier_batch; a.bv.sample_seg = p->frontier_seg; a.bv.sample_local0 = p->frontier_local0;
    a.bv.unique_batch = p->nodes_out_batch; a.bv.unique_seg = p->nodes_out_seg;
    a.bv.frontier_out = p->frontier_out; a.bv.frontier_batch_out = p->frontier_out_batch;
    a.bv.frontier_seg_out = p->frontier_out_seg; a.bv.frontier_local0_out = p->frontier_out_local0;
    a.bv.neighbor_local_out = p->neighbor_local; a.bv.center_local_out = p->center_local;
    WG_REQUIRE_INPUT((p->flags & ~(WGAMD_HOP_NO_UNIQUE_PAD | WGAMD_HOP_COL_INT32)) == 0, "unknown flag bits");
    WG_REQUIRE_INPUT(!(p->flags & WGAMD_HOP_COL_INT32) || (p->id_dtype == WHOLEMEMORY_DT_INT64 && p->n_vertices > 0 &&
                                                           p->n_vertices < ((int64_t)1 << 31)),
                     "WGAMD_HOP_COL_INT32 needs INT64 ids and 0 < n_vertices < 2^31");
    a.bv.no_pad = (p->flags & WGAMD_HOP_NO_UNIQUE_PAD) ? 1 : 0;
    a.col32     = (p->flags & WGAMD_HOP_COL_INT32) != 0;
    a.offsets = p->offsets; a.neighbor_row = p->neighbor_row_scratch; a.center_row = p->center_row_scratch;
    a.edge_gid = p->edge_gid; a.edge_cap = p->edge_cap; a.unique = p->nodes_out; a.counts_dev = p->counts_dev;
    a.workspace = p->workspace; a.workspace_bytes = p->workspace_bytes; a.stream = static_cast<hipStream_t>(stream);
    if (p->csr_weight != nullptr) {
      WG_REQUIRE_INPUT(p->weight_dtype == WHOLEMEMORY_DT_FLOAT || p->weight_dtype == WHOLEMEMORY_DT_DOUBLE,
                       "csr_weight must be FLOAT or DOUBLE");
      a.csr_weight = p->csr_weight; a.weight64 = p->weight_dtype == WHOLEMEMORY_DT_DOUBLE; a.max_row_len = p->max_row_len;
    }
    WG_REQUIRE_INPUT(a.neighbor_row != nullptr, "neighbor_row_scratch is NULL");
    run_hop(a);
  });
}

}  // extern "C"
