// Trainable embedding tables with sparse optimizers (include/wgamd_embedding.h).
//
// What the reference does (cpp/src/wholememory/embedding.cpp:136-313, embedding_optimizer.cpp,
// cpp/src/wholememory_ops/functions/embedding_optimizer_func.cu:166-1024): route every (row id, gradient row) to the rank
// that owns the row, de-duplicate the ids there while summing their gradients into a second buffer, then run one
// workgroup per unique row through the optimizer formula.  Here, for one MI355X partition in HBM:
//   * routing reuses the all-to-all-v of the feature fetch (wg_comm.hip); pairs a rank owns itself are not moved at all:
//     the update kernel reads their gradient rows where the caller left them;
//   * the received ids are radix-sorted ONCE as (local row, arrival position) pairs — only the bits a local row number
//     needs — and a single kernel walks the sorted order: the lane group that sits on the first pair of a row sums that
//     row's gradients (ordered by sender rank, then by the sender's own order: reproducible, and the same order
//     whatever the row partition is) and applies the update in the same pass.  No compaction, no
//     de-duplicated gradient buffer, no host round trip for the unique count;
//   * rows are 16-byte padded, so 4-wide vector accesses whenever the gradient rows allow it.
// Update formulas: embedding_optimizer_func.cu:203-214 (SGD), :394-421 (LazyAdam / AdamW), :657-671 (AdaGrad),
// :867-881 (RMSProp); the CPU twin the reference tests against is
// cpp/tests/wholememory_ops/wholememory_embedding_gradient_apply_tests.cu:221-300.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <cstring>
#include <rocprim/rocprim.hpp>
#include <string>
#include <type_traits>
#include <vector>

#include "wg_common.hpp"
#include "wgamd_embedding.h"

struct wholememory_embedding_cache_policy_ {
  wholememory_comm_t cache_comm                 = nullptr;
  wholememory_memory_type_t memory_type         = WHOLEMEMORY_MT_NONE;
  wholememory_memory_location_t memory_location = WHOLEMEMORY_ML_NONE;
  wholememory_access_type_t access_type         = WHOLEMEMORY_AT_NONE;
  float ratio                                   = 0.0f;
};

namespace wgamd {
// The READONLY cache of one rank (private HBM): set-associative, kCacheWays lines per set, one 32-lane group per looked-up
// id.  tags = cached entry id (-1: empty line), counts = hits since the line was filled (halved whenever the set refuses
// a newcomer), locks = one try-lock per set, taken only by the insert kernel.
constexpr int kCacheWays      = 32;
constexpr int kCacheLockTries = 64;   // then the row is simply not cached this time
struct cache_view {
  int64_t* tags;
  int* counts;
  int* locks;
  char* data;
  int64_t n_sets, entries;
  int64_t line_bytes;          // padded row
  unsigned long long* stats;   // {hits, lookups} since creation / the last drop
  // READWRITE (write-back) cache only — see "READWRITE device cache" below
  int* dirty;                  // per line: the line is newer than its table row
  float* st;                   // per line: the row's optimizer state [n_states * state_dim | beta1^t beta2^t], or null
  int64_t st_stride;           // floats per state line
  int st_floats, rs_off;       // state floats copied per line; offset of the per-row pair in a state line (-1: none)
};
}  // namespace wgamd

struct wholememory_embedding_optimizer_ {
  wholememory_optimizer_type_t type = WHOLEMEMORY_OPT_NONE;
  float weight_decay = 0.0f, epsilon = 1e-8f, beta1 = 0.9f, beta2 = 0.999f, adam_w = 0.0f, alpha = 0.99f;
};

struct wholememory_embedding_ {
  wholememory_tensor_t allocated = nullptr;  // [entries, padded_dim]
  wholememory_tensor_t user      = nullptr;  // view [entries, dim]
  wholememory_comm_t comm        = nullptr;
  wholememory_dtype_t dtype      = WHOLEMEMORY_DT_UNKNOWN;
  int64_t entries = 0, dim = 0, padded_dim = 0;
  wholememory_embedding_optimizer_t optimizer = nullptr;
  wholememory_tensor_t state_table = nullptr;  // fp32 [entries, n_states * state_dim], same row partition
  wholememory_tensor_t row_state   = nullptr;  // fp32 [entries, 2] (LazyAdam beta1^t, beta2^t)
  int64_t state_dim                = 0;        // dim rounded up to 4 floats
  std::vector<std::string> state_names;
  std::vector<wholememory_tensor_t> state_views;
  std::vector<const char*> names_c;  // NULL-terminated
  bool cached = false;               // a cache sits in front of `user` (READONLY: asker-side; READWRITE: owner-side)
  bool cache_rw = false;             // ... and it is the write-back cache of this rank's own rows
  wholememory_memory_location_t location = WHOLEMEMORY_ML_DEVICE;
  wgamd::cache_view cache{};
  wholememory_tensor_t cache_rows = nullptr;  // [lines, dim] view of cache.data (caller-storage tensor, no handle)
};

namespace wgamd {

namespace {

enum { kSgd = 0, kLazyAdam, kAdaGrad, kRmsProp };

struct step_params {
  float lr, weight_decay, epsilon, beta1, beta2, alpha;
  int adam_w;
};

// where the update kernel finds a row that sits in the write-back cache: line_of[i] = cache line of the row whose first
// sorted pair is i (-1: not resident, update the table row); line_of == nullptr: no write-back cache
struct cache_redirect {
  const int* line_of = nullptr;
  char* data         = nullptr;
  int64_t line_bytes = 0;
  float* st          = nullptr;
  int64_t st_stride  = 0;
  int rs_off         = 0;
};

template <typename T>
__device__ __forceinline__ float emb_to_f32(T v);
template <>
__device__ __forceinline__ float emb_to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float emb_to_f32<__half>(__half v) { return __half2float(v); }
template <>
__device__ __forceinline__ float emb_to_f32<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }
template <typename T>
__device__ __forceinline__ T emb_from_f32(float v);
template <>
__device__ __forceinline__ float emb_from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ __half emb_from_f32<__half>(float v) { return __float2half(v); }
template <>
__device__ __forceinline__ __hip_bfloat16 emb_from_f32<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

template <typename T, int V>
struct alignas(sizeof(T) * V) pack {
  T v[V];
};

template <int V> struct cvec;
template <> struct cvec<16> { using type = uint4; };
template <> struct cvec<8> { using type = uint2; };
template <> struct cvec<4> { using type = uint32_t; };
template <> struct cvec<2> { using type = uint16_t; };
template <> struct cvec<1> { using type = uint8_t; };

// sort key of a routed pair: its local row, or `local_rows` (one past the last row) for ids to skip
// Arrival order = sender rank, then the sender's own order: pairs [0, n_before) came from lower ranks, the next n_self are
// my own (still in the caller's buffers), the rest came from higher ranks.  All ids are local row numbers already.
struct arrival_map {
  int64_t n_before, n_self;
  // arrival position -> index into the received rows (>= 0) or ~(index into my own pairs) (< 0)
  __host__ __device__ int64_t source(int64_t v) const
  {
    return v < n_before ? v : v < n_before + n_self ? ~(v - n_before) : v - n_self;
  }
};

__global__ void __launch_bounds__(256) sort_keys_kernel(const int64_t* __restrict__ recv_ids, arrival_map am,
                                                        const int64_t* __restrict__ self_ids, int64_t n, int64_t local_rows,
                                                        uint64_t* __restrict__ keys, int* __restrict__ vals)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t src = am.source(i);
  const int64_t id  = src >= 0 ? recv_ids[src] : self_ids[~src];
  keys[i] = (id < 0 || id >= local_rows) ? (uint64_t)local_rows : (uint64_t)id;
  vals[i] = (int)i;
}

__global__ void __launch_bounds__(256) fill_f32_kernel(float* p, int64_t n, float v)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// One lane group per sorted pair; only the group on the FIRST pair of a row works: it sums the row's gradients over the
// run of equal keys and updates embedding + states.  st = [m | v] / [state_sum] / [v], each `sdim` floats wide.
template <typename EmbT, int OPT, int V>
__global__ void __launch_bounds__(256)
sparse_apply_kernel(const uint64_t* __restrict__ keys, const int* __restrict__ vals, int64_t n, int64_t local_rows,
                    const float* __restrict__ recv_rows, arrival_map am, const float* __restrict__ grads, int64_t ldg,
                    const int64_t* __restrict__ self_pos, EmbT* __restrict__ emb, int64_t lde, float* __restrict__ st,
                    int64_t lds, int64_t sdim, float* __restrict__ row_state, int dim, step_params p, int log2_lanes,
                    cache_redirect cr)
{
  const int lanes       = 1 << log2_lanes;
  const int64_t tid     = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int sub         = (int)(tid & (lanes - 1));
  const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> log2_lanes;
  for (int64_t i = tid >> log2_lanes; i < n; i += ngroups) {
    const uint64_t key = keys[i];
    if (key >= (uint64_t)local_rows) continue;
    if (i > 0 && keys[i - 1] == key) continue;
    int64_t end = i + 1;
    while (end < n && keys[end] == key) end++;
    EmbT* e_row  = emb + key * lde;
    float* s_row = OPT == kSgd ? nullptr : st + key * lds;
    float* rs    = OPT == kLazyAdam ? row_state + key * 2 : nullptr;
    if (cr.line_of != nullptr) {   // write-back cache: a resident row is updated in its line (marked dirty by the locate pass)
      const int64_t ln = cr.line_of[i];
      if (ln >= 0) {
        e_row = reinterpret_cast<EmbT*>(cr.data + ln * cr.line_bytes);
        if (OPT != kSgd) s_row = cr.st + ln * cr.st_stride;
        if (OPT == kLazyAdam) rs = cr.st + ln * cr.st_stride + cr.rs_off;
      }
    }
    float b1t = 0.f, b2t = 0.f;
    if (OPT == kLazyAdam) {
      b1t = rs[0] * p.beta1;
      b2t = rs[1] * p.beta2;
    }
    for (int c = sub * V; c < dim; c += lanes * V) {
      float g[V];
#pragma unroll
      for (int j = 0; j < V; j++) g[j] = 0.f;
      for (int64_t k = i; k < end; k++) {
        const int64_t from = am.source(vals[k]);  // a routed row, or one of my own pairs read in place
        const float* src   = from >= 0 ? recv_rows + from * dim : grads + self_pos[~from] * ldg;
        const pack<float, V> t = *reinterpret_cast<const pack<float, V>*>(src + c);
#pragma unroll
        for (int j = 0; j < V; j++) g[j] += t.v[j];
      }
      pack<EmbT, V> ev = *reinterpret_cast<const pack<EmbT, V>*>(e_row + c);
      pack<float, V> s0, s1;
      if (OPT != kSgd) s0 = *reinterpret_cast<const pack<float, V>*>(s_row + c);
      if (OPT == kLazyAdam) s1 = *reinterpret_cast<const pack<float, V>*>(s_row + sdim + c);
#pragma unroll
      for (int j = 0; j < V; j++) {
        float x  = emb_to_f32<EmbT>(ev.v[j]);
        float gv = g[j];
        if (OPT == kLazyAdam && p.adam_w) {
          x -= p.lr * p.weight_decay * x;
        } else {
          gv += p.weight_decay * x;
        }
        if (OPT == kSgd) {
          x -= p.lr * gv;
        } else if (OPT == kLazyAdam) {
          const float m = p.beta1 * s0.v[j] + (1.f - p.beta1) * gv;
          const float v = p.beta2 * s1.v[j] + (1.f - p.beta2) * gv * gv;
          const float mhat = m / (1.f - b1t);
          const float vhat = v / (1.f - b2t);
          x -= p.lr * mhat / (sqrtf(vhat) + p.epsilon);
          s0.v[j] = m;
          s1.v[j] = v;
        } else if (OPT == kAdaGrad) {
          const float sum = s0.v[j] + gv * gv;
          x -= p.lr * gv / (sqrtf(sum) + p.epsilon);
          s0.v[j] = sum;
        } else {
          const float v = p.alpha * s0.v[j] + (1.f - p.alpha) * gv * gv;
          x -= p.lr * gv / (sqrtf(v) + p.epsilon);
          s0.v[j] = v;
        }
        ev.v[j] = emb_from_f32<EmbT>(x);
      }
      *reinterpret_cast<pack<EmbT, V>*>(e_row + c) = ev;
      if (OPT != kSgd) *reinterpret_cast<pack<float, V>*>(s_row + c) = s0;
      if (OPT == kLazyAdam) *reinterpret_cast<pack<float, V>*>(s_row + sdim + c) = s1;
    }
    if (OPT == kLazyAdam && sub == 0) {  // after every lane of the group (same wave) has read the old powers
      rs[0] = b1t;
      rs[1] = b2t;
    }
  }
}

template <typename EmbT, int OPT>
void launch_apply(bool vec4, const uint64_t* keys, const int* vals, int64_t n, int64_t local_rows, const float* recv_rows,
                  arrival_map am, const float* grads, int64_t ldg, const int64_t* self_pos, void* emb, int64_t lde, float* st, int64_t lds, int64_t sdim, float* row_state, int dim,
                  step_params p, cache_redirect cr, hipStream_t stream)
{
  const int V      = vec4 ? 4 : 1;
  int l2           = 0;
  while ((1 << l2) * V < dim && l2 < 6) l2++;
  const int64_t gpb = 256 >> l2;
  const int grid    = (int)std::min<int64_t>((n + gpb - 1) / gpb, 256 * 16);
  if (vec4)
    sparse_apply_kernel<EmbT, OPT, 4><<<grid, 256, 0, stream>>>(keys, vals, n, local_rows, recv_rows, am, grads, ldg, self_pos, static_cast<EmbT*>(emb),
                                                               lde, st, lds, sdim, row_state, dim, p, l2, cr);
  else
    sparse_apply_kernel<EmbT, OPT, 1><<<grid, 256, 0, stream>>>(keys, vals, n, local_rows, recv_rows, am, grads, ldg, self_pos, static_cast<EmbT*>(emb),
                                                               lde, st, lds, sdim, row_state, dim, p, l2, cr);
  WG_HIP_CHECK(hipGetLastError());
}

template <typename EmbT>
void dispatch_opt(int opt, bool vec4, const uint64_t* keys, const int* vals, int64_t n, int64_t local_rows, const float* recv_rows,
                  arrival_map am, const float* grads, int64_t ldg, const int64_t* self_pos, void* emb, int64_t lde, float* st, int64_t lds, int64_t sdim, float* row_state, int dim,
                  step_params p, cache_redirect cr, hipStream_t stream)
{
  switch (opt) {
    case kSgd: return launch_apply<EmbT, kSgd>(vec4, keys, vals, n, local_rows, recv_rows, am, grads, ldg, self_pos, emb, lde, st, lds, sdim, row_state, dim, p, cr, stream);
    case kLazyAdam: return launch_apply<EmbT, kLazyAdam>(vec4, keys, vals, n, local_rows, recv_rows, am, grads, ldg, self_pos, emb, lde, st, lds, sdim, row_state, dim, p, cr, stream);
    case kAdaGrad: return launch_apply<EmbT, kAdaGrad>(vec4, keys, vals, n, local_rows, recv_rows, am, grads, ldg, self_pos, emb, lde, st, lds, sdim, row_state, dim, p, cr, stream);
    default: return launch_apply<EmbT, kRmsProp>(vec4, keys, vals, n, local_rows, recv_rows, am, grads, ldg, self_pos, emb, lde, st, lds, sdim, row_state, dim, p, cr, stream);
  }
}

int opt_code(wholememory_optimizer_type_t t)
{
  switch (t) {
    case WHOLEMEMORY_OPT_SGD: return kSgd;
    case WHOLEMEMORY_OPT_LAZY_ADAM: return kLazyAdam;
    case WHOLEMEMORY_OPT_ADAGRAD: return kAdaGrad;
    case WHOLEMEMORY_OPT_RMSPROP: return kRmsProp;
    default: return -1;
  }
}

void* local_pointer(wholememory_tensor_t t, size_t* local_bytes = nullptr)
{
  void* ptr = nullptr;
  size_t sz = 0, off = 0;
  if (wholememory_get_local_memory(&ptr, &sz, &off, wholememory_tensor_get_memory_handle(t)) != WHOLEMEMORY_SUCCESS)
    throw logic_error("no local memory");
  if (local_bytes) *local_bytes = sz;
  return ptr;
}

std::vector<size_t> entry_partition(wholememory_tensor_t t)
{
  wholememory_handle_t h = wholememory_tensor_get_memory_handle(t);
  int W                  = 1;
  wholememory_comm_t c   = nullptr;
  wholememory_get_communicator(&c, h);
  wholememory_communicator_get_size(&W, c);
  std::vector<size_t> sizes(W);
  wholememory_get_rank_partition_sizes(sizes.data(), h);
  const size_t g = wholememory_get_data_granularity(h);
  for (auto& s : sizes) s /= g;
  return sizes;
}

void destroy_states(wholememory_embedding_t e)
{
  for (auto v : e->state_views) wholememory_destroy_tensor(v);
  e->state_views.clear();
  if (e->state_table) wholememory_destroy_tensor(e->state_table);
  if (e->row_state) wholememory_destroy_tensor(e->row_state);
  e->state_table = e->row_state = nullptr;
}

// write-back cache (defined with the cache kernels below): bring the step's rows in (adjust), find the resident ones, mark
// them dirty; returns what the update kernel needs to reach their lines
cache_redirect rw_prepare_step(wholememory_embedding_t e, const uint64_t* sorted_keys, int64_t n, bool adjust_cache,
                               int* line_of, hipStream_t stream);

void step(wholememory_embedding_t e, wholememory_tensor_t indices, wholememory_tensor_t grads, bool adjust_cache, float lr,
          wholememory_env_func_t* env, hipStream_t stream)
{
  WG_REQUIRE_INPUT(e && indices && grads && env, "null argument");
  if (!e->optimizer) throw logic_error("no optimizer set on this embedding");
  const auto* id = wholememory_tensor_get_tensor_description(indices);
  const auto* gd = wholememory_tensor_get_tensor_description(grads);
  WG_REQUIRE_INPUT(id->dim == 1 && (id->dtype == WHOLEMEMORY_DT_INT || id->dtype == WHOLEMEMORY_DT_INT64),
                   "indices must be a 1-D int32 / int64 tensor");
  WG_REQUIRE_INPUT(gd->dim == 2 && gd->dtype == WHOLEMEMORY_DT_FLOAT, "grads must be a 2-D float tensor");
  WG_REQUIRE_INPUT(gd->sizes[0] == id->sizes[0] && gd->sizes[1] == e->dim && gd->strides[0] >= e->dim && gd->strides[1] == 1,
                   "grads must be [len(indices), embedding_dim]");
  WG_REQUIRE_INPUT(!wholememory_tensor_has_handle(indices) && !wholememory_tensor_has_handle(grads),
                   "indices / grads must be plain device tensors");
  const int64_t n  = id->sizes[0];
  const size_t ies = dtype_size(id->dtype), es = dtype_size(e->dtype);
  const char* idx  = static_cast<const char*>(wholememory_tensor_get_data_pointer(indices)) + id->storage_offset * ies;
  const char* g    = static_cast<const char*>(wholememory_tensor_get_data_pointer(grads)) + gd->storage_offset * 4;
  int64_t sz2[2]   = {n, e->dim};
  wholememory_matrix_description_t gm = wholememory_create_matrix_desc(sz2, gd->strides[0], 0, WHOLEMEMORY_DT_FLOAT);

  // (1) route: who owns each pair; remote pairs travel (ids, then gradient rows), my own stay where they are
  id_exchange x(env);
  x.plan(wholememory_tensor_get_memory_handle(e->allocated), (size_t)e->padded_dim * es, 0, idx, id->dtype, n, true, stream);
  const int64_t local_rows = x.local_rows, n_recv = x.recv_total, R = x.recv_total + x.self_cnt, D = e->dim;
  WG_EXPECTS(R < (int64_t)1 << 31, "too many gradient rows in one call");
  unsigned bits = 1;
  while (((uint64_t)1 << bits) <= (uint64_t)local_rows && bits < 63) bits++;  // keys are in [0, local_rows]
  size_t sort_bytes = 0;
  if (R > 0)
    WG_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, sort_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (int*)nullptr,
                                           (int*)nullptr, (size_t)R, 0u, bits, stream));
  temp_arena arena(env);
  const size_t o_ids = arena.add(sizeof(int64_t) * n_recv), o_rows = arena.add(sizeof(float) * n_recv * D),
               o_send = arena.add(sizeof(float) * x.n_remote * D), o_k1 = arena.add(sizeof(uint64_t) * R),
               o_k2 = arena.add(sizeof(uint64_t) * R), o_v1 = arena.add(sizeof(int) * R), o_v2 = arena.add(sizeof(int) * R),
               o_tmp = arena.add(sort_bytes), o_line = arena.add(sizeof(int) * (e->cache_rw ? R : 0));
  arena.commit();
  x.exchange_ids(arena.at<int64_t>(o_ids), stream);
  gm.storage_offset = 0;
  int64_t ps[2]     = {x.n_remote, D};
  wholememory_matrix_description_t packed_m = wholememory_create_matrix_desc(ps, D, 0, WHOLEMEMORY_DT_FLOAT);
  local_rows_gather(g, gm, x.d_pos, WHOLEMEMORY_DT_INT64, x.n_remote, arena.at<char>(o_send), packed_m, stream);
  x.rows_to_owners(arena.at<char>(o_send), arena.at<char>(o_rows), sizeof(float) * D, stream);
  if (R > 0 && local_rows > 0) {
    // (2) one stable sort of (local row, arrival position) over the bits a local row number needs
    auto *keys = arena.at<uint64_t>(o_k1), *keys2 = arena.at<uint64_t>(o_k2);
    auto *vals = arena.at<int>(o_v1), *vals2 = arena.at<int>(o_v2);
    const arrival_map am{(int64_t)x.recv_at[x.me], x.self_cnt};
    sort_keys_kernel<<<(int)((R + 255) / 256), 256, 0, stream>>>(x.d_recv_ids, am, x.d_self_ids, R, local_rows, keys, vals);
    WG_HIP_CHECK(hipGetLastError());
    WG_HIP_CHECK(rocprim::radix_sort_pairs(arena.at<void>(o_tmp), sort_bytes, keys, keys2, vals, vals2, (size_t)R, 0u, bits, stream));

    // (3) sum the gradients of every row and update it, one pass
    const auto* o = e->optimizer;
    step_params p{lr, o->weight_decay, o->epsilon, o->beta1, o->beta2, o->alpha, o->adam_w > 0.5f ? 1 : 0};
    void* emb        = local_pointer(e->allocated);
    float* st        = e->state_table ? static_cast<float*>(local_pointer(e->state_table)) : nullptr;
    float* row_state = e->row_state ? static_cast<float*>(local_pointer(e->row_state)) : nullptr;
    const int64_t lds = e->state_table ? wholememory_tensor_get_tensor_description(e->state_table)->strides[0] : 0;
    const float* d_rows = arena.at<float>(o_rows);
    const float* gsrc   = reinterpret_cast<const float*>(g);
    const int64_t ldg   = gd->strides[0];
    // 16-byte vector reads need aligned gradient rows on both sources (routed rows are packed [n_recv, dim])
    const bool vec4 = D % 4 == 0 && (x.self_cnt == 0 || (ldg % 4 == 0 && reinterpret_cast<uintptr_t>(gsrc) % 16 == 0));
    const int opt   = opt_code(o->type);
    cache_redirect cr;
    if (e->cache_rw) cr = rw_prepare_step(e, keys2, R, adjust_cache, arena.at<int>(o_line), stream);
    switch (e->dtype) {
      case WHOLEMEMORY_DT_FLOAT:
        dispatch_opt<float>(opt, vec4, keys2, vals2, R, local_rows, d_rows, am, gsrc, ldg, x.d_self_pos, emb, e->padded_dim,
                            st, lds, e->state_dim, row_state, (int)D, p, cr, stream);
        break;
      case WHOLEMEMORY_DT_HALF:
        dispatch_opt<__half>(opt, vec4, keys2, vals2, R, local_rows, d_rows, am, gsrc, ldg, x.d_self_pos, emb, e->padded_dim,
                             st, lds, e->state_dim, row_state, (int)D, p, cr, stream);
        break;
      default:
        dispatch_opt<__hip_bfloat16>(opt, vec4, keys2, vals2, R, local_rows, d_rows, am, gsrc, ldg, x.d_self_pos, emb,
                                     e->padded_dim, st, lds, e->state_dim, row_state, (int)D, p, cr, stream);
        break;
    }
  }
  WG_HIP_CHECK(hipStreamSynchronize(stream));  // scratch is released on return
}

// ---- READONLY local cache ---------------------------------------------------------------------------------------
// What the reference does (cpp/src/wholememory/embedding.cpp:776-894 local_cached_global_readonly_embedding,
// embedding_cache.hpp:34-163): a set-associative cache, 32 lines per set, LFU-ish replacement, held by a smaller
// communicator in front of a table spread over a bigger one; a gather with adjust_cache first refreshes the cache with
// the ids it is about to read, then reads through it.  Here the cache is private to one MI355X (the table already lives
// in HBM, so what a cache can save is xGMI traffic to the peers, never a slower memory tier):
//   1. lookup  — one 32-lane group per id reads its set's 32 tags in one 256-byte access and ballots; writes the cache
//                line of every hit and the id of every miss (the other list gets -1 = "skip this row");
//   2. hits    — the ordinary local row gather, cache lines -> output rows;
//   3. misses  — the ordinary table gather with the miss list: skipped ids stay in the asker's own bucket of the
//                exchange, so only missed rows cross the wire;
//   4. insert  — (adjust_cache) the rows the misses just fetched are copied from the OUTPUT into the cache: no second
//                fetch.  A set is updated under a try-lock; a group that keeps finding the lock taken, or finds every
//                line of the set hit at least once since it was filled, gives up (the latter halves the set's counters
//                first, so a stale hot line is displaced after log2(hits) refusals).  Best effort by design: whatever the
//                cache holds is an exact copy of a table row, so a gather returns the same bytes with or without it.
// hits / lookups of a launch reach the two global counters as ONE pair of atomics per workgroup (a pair per lane group
// was 64 k same-address atomics for 500 k ids — 0.3 ms of a 0.4 ms kernel)
__device__ __forceinline__ void stats_flush(unsigned long long* stats, unsigned long long hits, unsigned long long looked, bool count)
{
  __shared__ unsigned long long sh[2];
  if (threadIdx.x == 0) sh[0] = sh[1] = 0;
  __syncthreads();
  if (count && (threadIdx.x & (kCacheWays - 1)) == 0 && looked) {
    atomicAdd(&sh[0], hits);
    atomicAdd(&sh[1], looked);
  }
  __syncthreads();
  if (count && threadIdx.x == 0 && sh[1]) {
    atomicAdd(&stats[0], sh[0]);
    atomicAdd(&stats[1], sh[1]);
  }
}

__device__ __forceinline__ int64_t cache_set_of(int64_t id, int64_t n_sets)
{
  return (int64_t)((((uint64_t)id * 0x9E3779B97F4A7C15ull) >> 24) % (uint64_t)n_sets);
}

template <typename IdxT>
__global__ void __launch_bounds__(256)
cache_lookup_kernel(cache_view c, const IdxT* __restrict__ idx, int64_t n, bool count_hits, int64_t* __restrict__ line_idx,
                    int64_t* __restrict__ miss_idx)
{
  const int lane          = threadIdx.x & (kCacheWays - 1);
  const int64_t group     = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kCacheWays;
  const int64_t n_groups  = (int64_t)gridDim.x * blockDim.x / kCacheWays;
  unsigned long long hits = 0, looked = 0;
  for (int64_t i = group; i < n; i += n_groups) {
    const int64_t id  = (int64_t)idx[i];
    const bool valid  = id >= 0 && id < c.entries;
    const int64_t set = valid ? cache_set_of(id, c.n_sets) : 0;
    const int64_t tag = c.tags[set * kCacheWays + lane];
    const uint64_t b  = __ballot(valid && tag == id);
    const uint32_t hb = (uint32_t)(b >> (threadIdx.x & 32));
    const int way     = hb ? __ffs(hb) - 1 : -1;
    if (lane == 0) {
      line_idx[i] = way >= 0 ? set * kCacheWays + way : -1;
      miss_idx[i] = way >= 0 ? -1 : id;   // invalid ids go to the table gather as they are: same behaviour as uncached
      looked += valid;
      hits += way >= 0;
    }
    if (count_hits && way == lane) {
      int* cnt = &c.counts[set * kCacheWays + lane];
      if (*cnt < (1 << 24)) atomicAdd(cnt, 1);
    }
  }
  stats_flush(c.stats, hits, looked, true);
}

template <int V>
__global__ void __launch_bounds__(256)
cache_insert_kernel(cache_view c, const int64_t* __restrict__ miss_idx, int64_t n, const char* __restrict__ rows,
                    int64_t row_stride, int row_bytes)
{
  using vec_t = typename cvec<V>::type;
  const int lane         = threadIdx.x & (kCacheWays - 1);
  const int half         = threadIdx.x & 32;
  const int64_t group    = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kCacheWays;
  const int64_t n_groups = (int64_t)gridDim.x * blockDim.x / kCacheWays;
  for (int64_t i = group; i < n; i += n_groups) {
    const int64_t id = miss_idx[i];
    if (id < 0 || id >= c.entries) continue;
    const int64_t set  = cache_set_of(id, c.n_sets);
    const int64_t line = set * kCacheWays + lane;
    // Try-lock with a bounded number of retries.  The critical section sits INSIDE the retry loop: the two groups of one
    // wave may want the same set, and the one that waits must not keep the wave inside a loop the holder has already left.
    bool done = false;
    for (int tries = 0; tries < kCacheLockTries && !done; tries++) {
      int got = 0;
      if (lane == 0) got = atomicCAS(&c.locks[set], 0, 1) == 0;
      got = __shfl(got, half, 64);
      if (!got) {
        __builtin_amdgcn_s_sleep(8);
        continue;
      }
      __threadfence();
      const int64_t tag = __hip_atomic_load(&c.tags[line], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int cnt     = __hip_atomic_load(&c.counts[line], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool dup    = (uint32_t)(__ballot(tag == id) >> half) != 0;   // a duplicate of this id got here first
      if (!dup) {
        // victim = the empty line, else the least-hit one: min over (count + 1, lane) packed in one word
        unsigned key = ((tag < 0 ? 0u : (unsigned)cnt + 1u) << 5) | (unsigned)lane;
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) {
          const unsigned other = (unsigned)__shfl_xor((int)key, d, 64);
          key                  = other < key ? other : key;
        }
        const int victim = (int)(key & 31u);
        if ((key >> 5) <= 1u) {   // empty, or never hit since it was filled
          char* dst       = c.data + (set * kCacheWays + victim) * c.line_bytes;
          const char* src = rows + i * row_stride;
          for (int off = lane * V; off + V <= row_bytes; off += kCacheWays * V)
            *reinterpret_cast<vec_t*>(dst + off) = *reinterpret_cast<const vec_t*>(src + off);
          if (lane == victim) {
            __hip_atomic_store(&c.tags[line], id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&c.counts[line], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        } else {
          __hip_atomic_store(&c.counts[line], cnt >> 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __threadfence();
      if (lane == 0) atomicExch(&c.locks[set], 0);
      done = true;
    }
  }
}

void cache_clear(const cache_view& c, hipStream_t stream)
{
  const size_t lines = (size_t)c.n_sets * kCacheWays;
  WG_HIP_CHECK(hipMemsetAsync(c.tags, 0xff, lines * sizeof(int64_t), stream));
  WG_HIP_CHECK(hipMemsetAsync(c.counts, 0, lines * sizeof(int), stream));
  WG_HIP_CHECK(hipMemsetAsync(c.locks, 0, (size_t)c.n_sets * sizeof(int), stream));
  WG_HIP_CHECK(hipMemsetAsync(c.stats, 0, 2 * sizeof(unsigned long long), stream));
  if (c.dirty) WG_HIP_CHECK(hipMemsetAsync(c.dirty, 0, lines * sizeof(int), stream));
}

void cache_release(wholememory_embedding_t e)
{
  if (e->cache_rows) wholememory_destroy_tensor(e->cache_rows);
  e->cache_rows = nullptr;
  for (void* p : {(void*)e->cache.tags, (void*)e->cache.counts, (void*)e->cache.locks, (void*)e->cache.data, (void*)e->cache.stats,
                  (void*)e->cache.dirty, (void*)e->cache.st})
    if (p) (void)hipFree(p);
  e->cache  = cache_view{};
  e->cached = e->cache_rw = false;
}

void cache_allocate(wholememory_embedding_t e, float ratio, bool readwrite)
{
  cache_view& c = e->cache;
  const size_t es = dtype_size(e->dtype);
  c.entries    = e->entries;
  if (readwrite) {   // the write-back cache covers this rank's own rows, tagged by local row number
    size_t bytes = 0;
    (void)local_pointer(e->allocated, &bytes);
    c.entries = (int64_t)(bytes / ((size_t)e->padded_dim * es));
  }
  c.rs_off     = -1;
  c.line_bytes = e->padded_dim * (int64_t)es;
  int64_t lines = (int64_t)((double)ratio * (double)c.entries);
  c.n_sets      = std::max<int64_t>(1, (lines + kCacheWays - 1) / kCacheWays);
  lines         = c.n_sets * kCacheWays;
  try {
    WG_HIP_CHECK(hipMalloc(&c.tags, (size_t)lines * sizeof(int64_t)));
    WG_HIP_CHECK(hipMalloc(&c.counts, (size_t)lines * sizeof(int)));
    WG_HIP_CHECK(hipMalloc(&c.locks, (size_t)c.n_sets * sizeof(int)));
    WG_HIP_CHECK(hipMalloc(&c.data, (size_t)lines * (size_t)c.line_bytes));
    WG_HIP_CHECK(hipMalloc(&c.stats, 2 * sizeof(unsigned long long)));
    if (readwrite) WG_HIP_CHECK(hipMalloc(&c.dirty, (size_t)lines * sizeof(int)));
    cache_clear(c, nullptr);
    WG_HIP_CHECK(hipStreamSynchronize(nullptr));
    wholememory_tensor_description_t d;
    wholememory_initialize_tensor_desc(&d);
    d.dim        = 2;
    d.dtype      = e->dtype;
    d.sizes[0]   = lines;
    d.sizes[1]   = e->dim;
    d.strides[0] = e->padded_dim;
    d.strides[1] = 1;
    if (wholememory_make_tensor_from_pointer(&e->cache_rows, c.data, &d) != WHOLEMEMORY_SUCCESS)
      throw logic_error("cache line tensor");
  } catch (...) {
    cache_release(e);
    throw;
  }
  e->cached   = true;
  e->cache_rw = readwrite;
}

// steps 1-4 above.  Falls back to the plain table gather when the output converts the dtype (a cache line is a byte copy
// of a table row, and so must be what an insert reads back from the output).
void cached_gather(wholememory_embedding_t e, wholememory_tensor_t indices, wholememory_tensor_t output, bool adjust_cache,
                   wholememory_env_func_t* env, hipStream_t stream)
{
  WG_REQUIRE_INPUT(indices && output && env, "null argument");
  const auto* id = wholememory_tensor_get_tensor_description(indices);
  const auto* od = wholememory_tensor_get_tensor_description(output);
  WG_REQUIRE_INPUT(id->dim == 1 && (id->dtype == WHOLEMEMORY_DT_INT || id->dtype == WHOLEMEMORY_DT_INT64) && id->strides[0] == 1,
                   "indices must be a contiguous 1-D int32 / int64 tensor");
  WG_REQUIRE_INPUT(!wholememory_tensor_has_handle(indices) && !wholememory_tensor_has_handle(output),
                   "indices / output must be plain device tensors");
  const int64_t n = id->sizes[0];
  auto plain = [&] {
    const auto rc = wholememory_gather(e->user, indices, output, env, stream, -1);
    if (rc != WHOLEMEMORY_SUCCESS) throw logic_error(fmt("table gather failed (%d)", (int)rc));
  };
  if (n == 0 || od->dim != 2 || od->dtype != e->dtype || od->sizes[1] != e->dim || od->strides[1] != 1 || od->sizes[0] < n) {
    plain();   // (argument errors are reported by the table gather, with its own messages)
    return;
  }
  temp_arena arena(env);
  const size_t o_line = arena.add(sizeof(int64_t) * n), o_miss = arena.add(sizeof(int64_t) * n);
  arena.commit();
  int64_t* line_idx = arena.at<int64_t>(o_line);
  int64_t* miss_idx = arena.at<int64_t>(o_miss);
  const cache_view& c = e->cache;
  const char* idx = static_cast<const char*>(wholememory_tensor_get_data_pointer(indices)) + id->storage_offset * dtype_size(id->dtype);
  const int grid  = (int)std::max<int64_t>(1, std::min<int64_t>((n * kCacheWays + 255) / 256, 256 * 16));
  if (id->dtype == WHOLEMEMORY_DT_INT)
    cache_lookup_kernel<int32_t><<<grid, 256, 0, stream>>>(c, reinterpret_cast<const int32_t*>(idx), n, adjust_cache, line_idx, miss_idx);
  else
    cache_lookup_kernel<int64_t><<<grid, 256, 0, stream>>>(c, reinterpret_cast<const int64_t*>(idx), n, adjust_cache, line_idx, miss_idx);
  WG_HIP_CHECK(hipGetLastError());

  wholememory_tensor_description_t ld;
  wholememory_initialize_tensor_desc(&ld);
  ld.dim        = 1;
  ld.dtype      = WHOLEMEMORY_DT_INT64;
  ld.sizes[0]   = n;
  ld.strides[0] = 1;
  wholememory_tensor_t line_t = nullptr, miss_t = nullptr;
  WG_EXPECTS(wholememory_make_tensor_from_pointer(&line_t, line_idx, &ld) == WHOLEMEMORY_SUCCESS &&
               wholememory_make_tensor_from_pointer(&miss_t, miss_idx, &ld) == WHOLEMEMORY_SUCCESS,
             "index tensor");
  const auto rc_hit  = wholememory_gather(e->cache_rows, line_t, output, env, stream, -1);
  const auto rc_miss = wholememory_gather(e->user, miss_t, output, env, stream, -1);
  wholememory_destroy_tensor(line_t);
  wholememory_destroy_tensor(miss_t);
  if (rc_hit != WHOLEMEMORY_SUCCESS || rc_miss != WHOLEMEMORY_SUCCESS)
    throw logic_error(fmt("gather through the cache failed (%d, %d)", (int)rc_hit, (int)rc_miss));

  if (adjust_cache) {
    const size_t es      = dtype_size(e->dtype);
    const char* rows     = static_cast<const char*>(wholememory_tensor_get_data_pointer(output)) + od->storage_offset * es;
    const int64_t stride = od->strides[0] * (int64_t)es;
    const int row_bytes  = (int)(e->dim * (int64_t)es);
    int V = 16;
    while (V > 1 && ((row_bytes | stride | (int64_t)reinterpret_cast<uintptr_t>(rows)) & (V - 1)) != 0) V >>= 1;
    switch (V) {
      case 16: cache_insert_kernel<16><<<grid, 256, 0, stream>>>(c, miss_idx, n, rows, stride, row_bytes); break;
      case 8: cache_insert_kernel<8><<<grid, 256, 0, stream>>>(c, miss_idx, n, rows, stride, row_bytes); break;
      case 4: cache_insert_kernel<4><<<grid, 256, 0, stream>>>(c, miss_idx, n, rows, stride, row_bytes); break;
      case 2: cache_insert_kernel<2><<<grid, 256, 0, stream>>>(c, miss_idx, n, rows, stride, row_bytes); break;
      default: cache_insert_kernel<1><<<grid, 256, 0, stream>>>(c, miss_idx, n, rows, stride, row_bytes); break;
    }
    WG_HIP_CHECK(hipGetLastError());
  }
  WG_HIP_CHECK(hipStreamSynchronize(stream));  // the two index lists are released on return
}

// ---- READWRITE device cache over this rank's own rows ------------------------------------------------------------
// What the reference does (cpp/src/wholememory/embedding.cpp:556-759 device_cached_host_embedding, :136-313 the training
// step with adjust_cache, embedding_cache.cpp:256-331, functions/embedding_cache_func.cu:107-723): the table — typically
// in HOST memory — is fronted by a device cache held by the SAME communicator; a gather or a training step with
// adjust_cache first brings the rows it touches into the cache of the rank that OWNS them (writing displaced modified
// lines back), reads and optimizer updates go to the cache line when the row is resident and to the table row when it
// is not, and writeback_all_cache / drop_all_cache flush the modified lines.
// Here: the cache of a rank covers exactly its own partition (tags = LOCAL row numbers), in private HBM; the table
// partition may be pinned host memory the kernels reach over PCIe (wholememory_malloc, WHOLEMEMORY_ML_HOST) or HBM.
//   * every id is routed to its owner by the feature-fetch exchange (id_exchange); only the owner touches its cache, in
//     stream order, so lookups need no lock and an insert serialises on its set's try-lock only against the other
//     groups of the same kernel;
//   * a cache line holds the padded embedding row AND, once an optimizer is set, the row's optimizer state (m / v /
//     state_sum and LazyAdam's beta powers) behind the same tag: one lookup per row per step, and a step on a resident
//     row moves no byte over PCIe;
//   * replacement = LFU with ageing: a line enters with count 1, a newcomer takes an empty line or one whose count has
//     aged to 0, otherwise it halves the set's counts and stays out (it is served from the table: caching is best
//     effort, the valid copy of a row is its line when resident and its table row otherwise — always).
struct rw_table {
  char* emb;            // this rank's partition
  int64_t emb_stride;   // bytes between rows (= line_bytes)
  float* st;            // state table partition [rows, st_floats] or null
  int64_t st_stride;    // floats
  float* rs;            // [rows, 2] or null
};

__device__ __forceinline__ void line_copy16(char* dst, const char* src, int bytes, int lane)
{
  for (int off = lane * 16; off < bytes; off += kCacheWays * 16)
    *reinterpret_cast<uint4*>(dst + off) = *reinterpret_cast<const uint4*>(src + off);
}

// moves one row (embedding + state) between its line and its table row; 32 lanes
template <bool TO_TABLE>
__device__ __forceinline__ void rw_move_row(const cache_view& c, const rw_table& t, int64_t line, int64_t row, int lane)
{
  char* l_emb = c.data + line * c.line_bytes;
  char* t_emb = t.emb + row * t.emb_stride;
  if (TO_TABLE) line_copy16(t_emb, l_emb, (int)c.line_bytes, lane); else line_copy16(l_emb, t_emb, (int)c.line_bytes, lane);
  if (c.st != nullptr) {
    float* l_st = c.st + line * c.st_stride;
    if (c.st_floats > 0) {
      float* t_st = t.st + row * t.st_stride;
      if (TO_TABLE) line_copy16((char*)t_st, (const char*)l_st, c.st_floats * 4, lane);
      else line_copy16((char*)l_st, (const char*)t_st, c.st_floats * 4, lane);
    }
    if (c.rs_off >= 0 && lane < 2) {
      if (TO_TABLE) t.rs[row * 2 + lane] = l_st[c.rs_off + lane]; else l_st[c.rs_off + lane] = t.rs[row * 2 + lane];
    }
  }
}

// (adjust_cache) one 32-lane group per id: resident -> count the use; else insert under the set's try-lock, writing a
// displaced modified line back first.  skip_repeats: `ids` is sorted, only the first of a run works.
template <typename IdT>
__global__ void __launch_bounds__(256)
rw_fill_kernel(cache_view c, rw_table t, const IdT* __restrict__ ids, int64_t n, bool skip_repeats, bool count_stats)
{
  const int lane         = threadIdx.x & (kCacheWays - 1);
  const int half         = threadIdx.x & 32;
  const int64_t group    = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kCacheWays;
  const int64_t n_groups = (int64_t)gridDim.x * blockDim.x / kCacheWays;
  unsigned long long hits = 0, looked = 0;   // a hit = the row was resident BEFORE this call brought it in
  for (int64_t i = group; i < n; i += n_groups) {
    const int64_t id = (int64_t)ids[i];
    if (id < 0 || id >= c.entries) continue;
    if (skip_repeats && i > 0 && (int64_t)ids[i - 1] == id) continue;
    looked++;
    const int64_t set  = cache_set_of(id, c.n_sets);
    const int64_t line = set * kCacheWays + lane;
    {
      // resident already (the common case once the cache is warm): count the use, no lock.  A neighbour that displaces the
      // line a moment later only makes the bump land on the newcomer's count.
      const int64_t tag0 = __hip_atomic_load(&c.tags[line], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t hb0 = (uint32_t)(__ballot(tag0 == id) >> half);
      if (hb0) {
        hits++;
        if (lane == __ffs(hb0) - 1 && __hip_atomic_load(&c.counts[line], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (1 << 24))
          atomicAdd(&c.counts[line], 1);
        continue;
      }
    }
    bool done = false;
    for (int tries = 0; tries < kCacheLockTries && !done; tries++) {
      int got = 0;
      if (lane == 0) got = atomicCAS(&c.locks[set], 0, 1) == 0;
      got = __shfl(got, half, 64);
      if (!got) {
        __builtin_amdgcn_s_sleep(8);
        continue;
      }
      __threadfence();
      const int64_t tag = __hip_atomic_load(&c.tags[line], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int cnt     = __hip_atomic_load(&c.counts[line], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t hb = (uint32_t)(__ballot(tag == id) >> half);
      if (hb) {
        // a duplicate of this id (same call) got here first: nothing to do
      } else {
        unsigned key = ((tag < 0 ? 0u : (unsigned)cnt + 1u) << 5) | (unsigned)lane;
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) {
          const unsigned other = (unsigned)__shfl_xor((int)key, d, 64);
          key                  = other < key ? other : key;
        }
        const int victim = (int)(key & 31u);
        if ((key >> 5) <= 1u) {   // empty, or aged to count 0
          const int64_t vline = set * kCacheWays + victim;
          const int64_t vtag  = __shfl(tag, half + victim, 64);
          int vdirty          = 0;
          if (lane == victim && vtag >= 0) vdirty = __hip_atomic_load(&c.dirty[vline], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          vdirty = __shfl(vdirty, half + victim, 64);
          if (vdirty) rw_move_row<true>(c, t, vline, vtag, lane);
          rw_move_row<false>(c, t, vline, id, lane);
          if (lane == victim) {
            __hip_atomic_store(&c.tags[line], id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&c.counts[line], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&c.dirty[line], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        } else {
          __hip_atomic_store(&c.counts[line], cnt >> 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __threadfence();
      if (lane == 0) atomicExch(&c.locks[set], 0);
      done = true;
    }
  }
  stats_flush(c.stats, hits, looked, count_stats);
}

// owner-side read: row i of `out` (padded rows, 16-byte multiples) = the line of ids[i] when resident, its table row when
// not, zeros for an id outside this partition
__global__ void __launch_bounds__(256)
rw_read_kernel(cache_view c, rw_table t, const int64_t* __restrict__ ids, int64_t n, char* __restrict__ out, bool count_stats)
{
  const int lane          = threadIdx.x & (kCacheWays - 1);
  const int64_t group     = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kCacheWays;
  const int64_t n_groups  = (int64_t)gridDim.x * blockDim.x / kCacheWays;
  unsigned long long hits = 0, looked = 0;
  for (int64_t i = group; i < n; i += n_groups) {
    const int64_t id  = ids[i];
    const bool valid  = id >= 0 && id < c.entries;
    const int64_t set = valid ? cache_set_of(id, c.n_sets) : 0;
    const int64_t tag = c.tags[set * kCacheWays + lane];
    const uint32_t hb = (uint32_t)(__ballot(valid && tag == id) >> (threadIdx.x & 32));
    char* dst         = out + i * c.line_bytes;
    if (!valid) {
      for (int off = lane * 16; off < (int)c.line_bytes; off += kCacheWays * 16) *reinterpret_cast<uint4*>(dst + off) = uint4{0, 0, 0, 0};
      continue;
    }
    const char* src = hb ? c.data + (set * kCacheWays + __ffs(hb) - 1) * c.line_bytes : t.emb + id * t.emb_stride;
    line_copy16(dst, src, (int)c.line_bytes, lane);
    if (lane == 0) {
      looked++;
      hits += hb != 0;
    }
  }
  stats_flush(c.stats, hits, looked, count_stats);
}

// the same read for ids THIS rank owns itself, straight into the caller's output rows (pos[i] < 0: the row is skipped)
template <int V, typename IdT>
__global__ void __launch_bounds__(256)
rw_read_direct_kernel(cache_view c, rw_table t, const IdT* __restrict__ ids, const int64_t* __restrict__ pos, int64_t n,
                      char* __restrict__ out, int64_t out_stride, int row_bytes, bool count_stats)
{
  using vec_t             = typename cvec<V>::type;
  const int lane          = threadIdx.x & (kCacheWays - 1);
  const int64_t group     = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kCacheWays;
  const int64_t n_groups  = (int64_t)gridDim.x * blockDim.x / kCacheWays;
  unsigned long long hits = 0, looked = 0;
  for (int64_t i = group; i < n; i += n_groups) {
    const int64_t id = (int64_t)ids[i];
    const int64_t p  = pos != nullptr ? pos[i] : (id < 0 ? -1 : i);   // no position list: the caller's own order, ids < 0 skipped
    if (p < 0) continue;
    const bool valid  = id >= 0 && id < c.entries;
    const int64_t set = valid ? cache_set_of(id, c.n_sets) : 0;
    const int64_t tag = c.tags[set * kCacheWays + lane];
    const uint32_t hb = (uint32_t)(__ballot(valid && tag == id) >> (threadIdx.x & 32));
    char* dst         = out + p * out_stride;
    if (!valid) {
      for (int off = lane * V; off + V <= row_bytes; off += kCacheWays * V) *reinterpret_cast<vec_t*>(dst + off) = vec_t{};
      continue;
    }
    const char* src = hb ? c.data + (set * kCacheWays + __ffs(hb) - 1) * c.line_bytes : t.emb + id * t.emb_stride;
    for (int off = lane * V; off + V <= row_bytes; off += kCacheWays * V)
      *reinterpret_cast<vec_t*>(dst + off) = *reinterpret_cast<const vec_t*>(src + off);
    if (lane == 0) {
      looked++;
      hits += hb != 0;
    }
  }
  stats_flush(c.stats, hits, looked, count_stats);
}

// training step: for the first sorted pair of every row, the line the row sits in (or -1); resident rows become dirty
__global__ void __launch_bounds__(256)
rw_locate_kernel(cache_view c, const uint64_t* __restrict__ keys, int64_t n, int* __restrict__ line_of)
{
  const int lane         = threadIdx.x & (kCacheWays - 1);
  const int64_t group    = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kCacheWays;
  const int64_t n_groups = (int64_t)gridDim.x * blockDim.x / kCacheWays;
  for (int64_t i = group; i < n; i += n_groups) {
    const uint64_t key = keys[i];
    const bool first   = key < (uint64_t)c.entries && (i == 0 || keys[i - 1] != key);
    if (!first) {
      if (lane == 0) line_of[i] = -1;
      continue;
    }
    const int64_t set = cache_set_of((int64_t)key, c.n_sets);
    const int64_t tag = c.tags[set * kCacheWays + lane];
    const uint32_t hb = (uint32_t)(__ballot(tag == (int64_t)key) >> (threadIdx.x & 32));
    const int way     = hb ? __ffs(hb) - 1 : -1;
    if (lane == 0) line_of[i] = way >= 0 ? (int)(set * kCacheWays + way) : -1;
    if (lane == way) c.dirty[set * kCacheWays + lane] = 1;
  }
}

// writeback_all_cache / drop_all_cache: one group per line
__global__ void __launch_bounds__(256) rw_writeback_kernel(cache_view c, rw_table t, bool drop)
{
  const int lane         = threadIdx.x & (kCacheWays - 1);
  const int64_t group    = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kCacheWays;
  const int64_t n_groups = (int64_t)gridDim.x * blockDim.x / kCacheWays;
  const int64_t lines    = c.n_sets * kCacheWays;
  for (int64_t l = group; l < lines; l += n_groups) {
    const int64_t tag = c.tags[l];
    if (tag < 0) continue;
    if (c.dirty[l]) rw_move_row<true>(c, t, l, tag, lane);
    if (lane == 0) {
      c.dirty[l] = 0;
      if (drop) {
        c.tags[l]   = -1;
        c.counts[l] = 0;
      }
    }
  }
}

rw_table rw_table_of(wholememory_embedding_t e)
{
  rw_table t{};
  t.emb        = static_cast<char*>(local_pointer(e->allocated));
  t.emb_stride = e->cache.line_bytes;
  if (e->state_table) {
    t.st        = static_cast<float*>(local_pointer(e->state_table));
    t.st_stride = wholememory_tensor_get_tensor_description(e->state_table)->strides[0];
  }
  if (e->row_state) t.rs = static_cast<float*>(local_pointer(e->row_state));
  return t;
}

int rw_grid(int64_t groups) { return (int)std::max<int64_t>(1, std::min<int64_t>((groups * kCacheWays + 255) / 256, 256 * 16)); }

void rw_writeback(wholememory_embedding_t e, bool drop, hipStream_t stream)
{
  const cache_view& c = e->cache;
  if (c.entries > 0) {
    rw_writeback_kernel<<<rw_grid(c.n_sets * kCacheWays), 256, 0, stream>>>(c, rw_table_of(e), drop);
    WG_HIP_CHECK(hipGetLastError());
  }
  if (drop) WG_HIP_CHECK(hipMemsetAsync(c.stats, 0, 2 * sizeof(unsigned long long), stream));
  WG_HIP_CHECK(hipStreamSynchronize(stream));
  // embedding_cache.cpp:313-317: the flush is complete on every rank when the call returns
  const auto rc = wholememory_communicator_barrier(e->comm);
  if (rc != WHOLEMEMORY_SUCCESS) throw logic_error(fmt("barrier after the cache write-back failed (%d)", (int)rc));
}

cache_redirect rw_prepare_step(wholememory_embedding_t e, const uint64_t* sorted_keys, int64_t n, bool adjust_cache,
                               int* line_of, hipStream_t stream)
{
  const cache_view& c = e->cache;
  cache_redirect cr;
  if (n == 0 || c.entries == 0) return cr;
  if (adjust_cache) {
    rw_fill_kernel<int64_t><<<rw_grid(n), 256, 0, stream>>>(c, rw_table_of(e), reinterpret_cast<const int64_t*>(sorted_keys), n, true, false);
    WG_HIP_CHECK(hipGetLastError());
  }
  rw_locate_kernel<<<rw_grid(n), 256, 0, stream>>>(c, sorted_keys, n, line_of);
  WG_HIP_CHECK(hipGetLastError());
  cr.line_of    = line_of;
  cr.data       = c.data;
  cr.line_bytes = c.line_bytes;
  cr.st         = c.st;
  cr.st_stride  = c.st_stride;
  cr.rs_off     = c.rs_off;
  return cr;
}

// gather through the write-back cache: ids -> owners, (adjust_cache: owners bring the rows in), owners read line-or-row,
// rows -> askers, un-permute (+ dtype conversion) into the output
void rw_cached_gather(wholememory_embedding_t e, wholememory_tensor_t indices, wholememory_tensor_t output, bool adjust_cache,
                      wholememory_env_func_t* env, hipStream_t stream)
{
  WG_REQUIRE_INPUT(indices && output && env, "null argument");
  const auto* id = wholememory_tensor_get_tensor_description(indices);
  const auto* od = wholememory_tensor_get_tensor_description(output);
  WG_REQUIRE_INPUT(id->dim == 1 && (id->dtype == WHOLEMEMORY_DT_INT || id->dtype == WHOLEMEMORY_DT_INT64) && id->strides[0] == 1,
                   "indices must be a contiguous 1-D int32 / int64 tensor");
  WG_REQUIRE_INPUT(!wholememory_tensor_has_handle(indices) && !wholememory_tensor_has_handle(output),
                   "indices / output must be plain device tensors");
  WG_REQUIRE_INPUT(od->dim == 2 && od->sizes[1] == e->dim && od->strides[1] == 1 && od->strides[0] >= e->dim,
                   "output must be [len(indices), embedding_dim]");
  const int64_t n = id->sizes[0];
  WG_REQUIRE_INPUT(od->sizes[0] >= n, "output has fewer rows than there are indices");
  const size_t es = dtype_size(e->dtype), oes = dtype_size(od->dtype);
  WG_REQUIRE_INPUT(oes != 0, "bad output dtype");
  const cache_view& c = e->cache;
  const char* idx = static_cast<const char*>(wholememory_tensor_get_data_pointer(indices)) + id->storage_offset * dtype_size(id->dtype);
  char* out       = static_cast<char*>(wholememory_tensor_get_data_pointer(output)) + od->storage_offset * oes;

  // ids this rank owns itself stay out of the exchange when the output keeps the table's dtype: one lookup-and-copy kernel
  // takes them from line-or-row straight to their output rows.  On a single-rank communicator that is the whole call, and
  // the ids are used where the caller left them: no plan, no host synchronisation, no scratch.
  const bool direct = od->dtype == e->dtype;
  const int64_t ostride = od->strides[0] * (int64_t)oes;
  const int row_bytes   = (int)(e->dim * (int64_t)es);
  int V = 16;
  while (V > 1 && ((row_bytes | ostride | (int64_t)reinterpret_cast<uintptr_t>(out)) & (V - 1)) != 0) V >>= 1;
  auto read_direct = [&](auto* ids_p, const int64_t* pos_p, int64_t cnt) {
    using IdT = std::remove_cv_t<std::remove_pointer_t<decltype(ids_p)>>;
    const rw_table t = rw_table_of(e);
    if (adjust_cache) {
      rw_fill_kernel<IdT><<<rw_grid(cnt), 256, 0, stream>>>(c, t, ids_p, cnt, false, true);
      WG_HIP_CHECK(hipGetLastError());
    }
    const int grid = rw_grid(cnt);
#define WG_RW_DIRECT(VV) \
  rw_read_direct_kernel<VV, IdT><<<grid, 256, 0, stream>>>(c, t, ids_p, pos_p, cnt, out, ostride, row_bytes, !adjust_cache)
    switch (V) {
      case 16: WG_RW_DIRECT(16); break;
      case 8: WG_RW_DIRECT(8); break;
      case 4: WG_RW_DIRECT(4); break;
      case 2: WG_RW_DIRECT(2); break;
      default: WG_RW_DIRECT(1); break;
    }
#undef WG_RW_DIRECT
    WG_HIP_CHECK(hipGetLastError());
  };
  int world = 1;
  wholememory_communicator_get_size(&world, e->comm);
  if (world == 1 && direct) {
    if (n > 0 && c.entries > 0) {
      if (id->dtype == WHOLEMEMORY_DT_INT) read_direct(reinterpret_cast<const int32_t*>(idx), nullptr, n);
      else read_direct(reinterpret_cast<const int64_t*>(idx), nullptr, n);
    }
    return;
  }
  id_exchange x(env);
  x.plan(wholememory_tensor_get_memory_handle(e->allocated), (size_t)c.line_bytes, 0, idx, id->dtype, n, direct, stream);
  temp_arena arena(env);
  const size_t o_ids = arena.add(sizeof(int64_t) * x.recv_total), o_rows = arena.add((size_t)c.line_bytes * x.recv_total),
               o_back = arena.add((size_t)c.line_bytes * x.n_remote);
  arena.commit();
  x.exchange_ids(arena.at<int64_t>(o_ids), stream);
  if (direct && x.self_cnt > 0 && c.entries > 0) read_direct(static_cast<const int64_t*>(x.d_self_ids), x.d_self_pos, x.self_cnt);
  if (x.recv_total > 0 && c.entries > 0) {
    const rw_table t = rw_table_of(e);
    if (adjust_cache) {
      rw_fill_kernel<int64_t><<<rw_grid(x.recv_total), 256, 0, stream>>>(c, t, x.d_recv_ids, x.recv_total, false, true);
      WG_HIP_CHECK(hipGetLastError());
    }
    rw_read_kernel<<<rw_grid(x.recv_total), 256, 0, stream>>>(c, t, x.d_recv_ids, x.recv_total, arena.at<char>(o_rows), !adjust_cache);
    WG_HIP_CHECK(hipGetLastError());
  }
  x.rows_to_askers(arena.at<char>(o_rows), arena.at<char>(o_back), (size_t)c.line_bytes, stream);
  int64_t sz2[2] = {x.n_remote, e->dim};
  wholememory_matrix_description_t back_m = wholememory_create_matrix_desc(sz2, e->padded_dim, 0, e->dtype);
  sz2[0] = od->sizes[0];
  wholememory_matrix_description_t out_m = wholememory_create_matrix_desc(sz2, od->strides[0], 0, od->dtype);
  local_rows_scatter(arena.at<char>(o_back), back_m, x.d_pos, WHOLEMEMORY_DT_INT64, x.n_remote, out, out_m, stream);
  WG_HIP_CHECK(hipStreamSynchronize(stream));  // the scratch is released on return
}

}  // namespace
}  // namespace wgamd

extern "C" {

using namespace wgamd;

wholememory_error_code_t wholememory_create_embedding_optimizer(wholememory_embedding_optimizer_t* optimizer,
                                                                wholememory_optimizer_type_t optimizer_type)
{
  if (!optimizer) return WHOLEMEMORY_INVALID_INPUT;
  if (opt_code(optimizer_type) < 0) return WHOLEMEMORY_NOT_IMPLEMENTED;  // embedding_optimizer.cpp:496-498
  *optimizer         = new wholememory_embedding_optimizer_();
  (*optimizer)->type = optimizer_type;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_optimizer_set_parameter(wholememory_embedding_optimizer_t o, const char* name, void* value)
{
  if (!o || !name || !value) return WHOLEMEMORY_INVALID_INPUT;
  const float v    = *static_cast<const float*>(value);
  const auto t     = o->type;
  const bool adam  = t == WHOLEMEMORY_OPT_LAZY_ADAM;
  float* slot      = nullptr;
  if (!strcmp(name, "weight_decay")) slot = &o->weight_decay;
  else if (!strcmp(name, "epsilon") && t != WHOLEMEMORY_OPT_SGD) slot = &o->epsilon;
  else if (!strcmp(name, "beta1") && adam) slot = &o->beta1;
  else if (!strcmp(name, "beta2") && adam) slot = &o->beta2;
  else if (!strcmp(name, "adam_w") && adam) slot = &o->adam_w;
  else if (!strcmp(name, "alpha") && t == WHOLEMEMORY_OPT_RMSPROP) slot = &o->alpha;
  if (!slot) {
    fprintf(stderr, "[wholegraph_amd] parameter name %s is not valid for this optimizer\n", name);
    return WHOLEMEMORY_INVALID_INPUT;
  }
  *slot = v;
  return WHOLEMEMORY_SUCCESS;
}

void wholememory_destroy_embedding_optimizer(wholememory_embedding_optimizer_t optimizer) { delete optimizer; }

wholememory_error_code_t wholememory_create_embedding_cache_policy(wholememory_embedding_cache_policy_t* cache_policy,
                                                                   wholememory_comm_t cache_level_comm,
                                                                   wholememory_memory_type_t memory_type,
                                                                   wholememory_memory_location_t memory_location,
                                                                   wholememory_access_type_t access_type, float cache_ratio)
{
  if (!cache_policy) return WHOLEMEMORY_INVALID_INPUT;
  *cache_policy = nullptr;
  if (cache_ratio > 1.0f || cache_ratio < 1.0f / 512) {  // embedding.cpp:917-920
    fprintf(stderr, "[wholegraph_amd] cache_ratio should in range [1/512, 1.0]\n");
    return WHOLEMEMORY_INVALID_VALUE;
  }
  auto* p            = new wholememory_embedding_cache_policy_();
  p->cache_comm      = cache_level_comm;
  p->memory_type     = memory_type;
  p->memory_location = memory_location;
  p->access_type     = access_type;
  p->ratio           = cache_ratio;
  *cache_policy      = p;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_destroy_embedding_cache_policy(wholememory_embedding_cache_policy_t cache_policy)
{
  delete cache_policy;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_create_embedding(wholememory_embedding_t* out, wholememory_tensor_description_t* desc,
                                                      wholememory_comm_t comm, wholememory_memory_type_t memory_type,
                                                      wholememory_memory_location_t memory_location,
                                                      wholememory_embedding_cache_policy_t cache_policy,
                                                      size_t* embedding_entry_partition, int /*user_defined_sms*/,
                                                      int round_robin_size)
{
  if (!out || !desc || !comm) return WHOLEMEMORY_INVALID_INPUT;
  if (desc->dim != 2 || desc->storage_offset != 0 || desc->sizes[1] <= 0) {
    fprintf(stderr, "[wholegraph_amd] wholememory_create_embedding: the description must be a 2-D matrix\n");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  if (round_robin_size != 0) {
    fprintf(stderr, "[wholegraph_amd] wholememory_create_embedding: round-robin sharding is not supported\n");
    return WHOLEMEMORY_NOT_SUPPORTED;
  }
  bool readwrite = false;
  if (cache_policy != nullptr) {
    // embedding.cpp:957-1009.  Two kinds of cache:
    //   * READWRITE, held by the table's own communicator (the reference's device_cached_host_embedding): the write-back
    //     cache of every rank's own rows, in HBM, in front of a table partition in pinned host memory (or HBM);
    //   * READONLY (the reference's local_cached_global_readonly_embedding, and here also a READONLY policy on the
    //     table's communicator): a private per-rank cache of the rows a rank asks for, saving peer traffic.
    if (cache_policy->access_type == WHOLEMEMORY_AT_READWRITE) {
      if (cache_policy->cache_comm != comm) {
        fprintf(stderr, "[wholegraph_amd] wholememory_create_embedding: Only ReadOnly access type supported for local "
                        "cached global readonly embedding.\n");
        return WHOLEMEMORY_INVALID_INPUT;  // embedding.cpp:1000-1004
      }
      if (cache_policy->memory_location != WHOLEMEMORY_ML_DEVICE) {
        fprintf(stderr, "[wholegraph_amd] wholememory_create_embedding: Cache has same communicator with raw embedding, "
                        "should be device cached host embedding, but cache memory location is not WHOLEMEMORY_ML_DEVICE.\n");
        return WHOLEMEMORY_INVALID_INPUT;  // embedding.cpp:962-967
      }
      if (cache_policy->memory_type < memory_type) {
        fprintf(stderr, "[wholegraph_amd] wholememory_create_embedding: For device cached host memory, raw embedding "
                        "should cover cache's address modes.\n");
        return WHOLEMEMORY_INVALID_INPUT;  // embedding.cpp:968-972
      }
      readwrite = true;
    } else if (cache_policy->access_type != WHOLEMEMORY_AT_READONLY) {
      return WHOLEMEMORY_INVALID_INPUT;
    }
    if (!readwrite && cache_policy->cache_comm != comm && cache_policy->memory_type == WHOLEMEMORY_MT_DISTRIBUTED) {
      fprintf(stderr,
              "[wholegraph_amd] wholememory_create_embedding: for local cached global readonly embedding, "
              "cache_memory_type should be chunked or continuous\n");
      return WHOLEMEMORY_INVALID_INPUT;  // embedding.cpp:986-992
    }
    embedding_entry_partition = nullptr;  // embedding.cpp:1009
  }
  const size_t es = wholememory_dtype_get_element_size(desc->dtype);
  if (es == 0 || es > 16) return WHOLEMEMORY_INVALID_INPUT;
  auto* e       = new wholememory_embedding_();
  e->comm       = comm;
  e->location   = memory_location;
  e->dtype      = desc->dtype;
  e->entries    = desc->sizes[0];
  e->dim        = desc->sizes[1];
  const int64_t align = es >= 16 ? 1 : (int64_t)(16 / es);
  e->padded_dim = (e->dim + align - 1) / align * align;
  wholememory_tensor_description_t padded = *desc;
  padded.sizes[1]   = e->padded_dim;
  padded.strides[0] = e->padded_dim;
  padded.strides[1] = 1;
  auto rc = wholememory_create_tensor(&e->allocated, &padded, comm, memory_type, memory_location, embedding_entry_partition);
  if (rc != WHOLEMEMORY_SUCCESS) {
    delete e;
    return rc;
  }
  int64_t starts[2] = {0, 0}, ends[2] = {-1, e->dim};
  rc = wholememory_tensor_get_subtensor(e->allocated, starts, ends, &e->user);
  if (rc != WHOLEMEMORY_SUCCESS) {
    wholememory_destroy_tensor(e->allocated);
    delete e;
    return rc;
  }
  e->names_c = {nullptr};
  if (cache_policy != nullptr) {
    rc = guarded("wholememory_create_embedding", [&] { cache_allocate(e, cache_policy->ratio, readwrite); });
    if (rc != WHOLEMEMORY_SUCCESS) {
      wholememory_destroy_tensor(e->user);
      wholememory_destroy_tensor(e->allocated);
      delete e;
      return rc;
    }
  }
  *out = e;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_destroy_embedding(wholememory_embedding_t e)
{
  if (!e) return WHOLEMEMORY_INVALID_INPUT;
  destroy_states(e);
  cache_release(e);
  wholememory_destroy_tensor(e->user);
  wholememory_destroy_tensor(e->allocated);
  delete e;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_tensor_t wholememory_embedding_get_embedding_tensor(wholememory_embedding_t e) { return e ? e->user : nullptr; }

wholememory_error_code_t wholememory_embedding_set_optimizer(wholememory_embedding_t e, wholememory_embedding_optimizer_t opt)
{
  if (!e || !opt) return WHOLEMEMORY_INVALID_INPUT;
  if (e->optimizer) {
    fprintf(stderr, "[wholegraph_amd] wholememory_embedding_set_optimizer: the optimizer can only be set once\n");
    return WHOLEMEMORY_LOGIC_ERROR;  // embedding.cpp:487-491
  }
  if (e->cached && !e->cache_rw) {
    fprintf(stderr, "[wholegraph_amd] optimizer not supported for local cached global readonly embedding.\n");
    return WHOLEMEMORY_INVALID_INPUT;  // embedding.cpp:55-60
  }
  if (e->dtype != WHOLEMEMORY_DT_FLOAT && e->dtype != WHOLEMEMORY_DT_HALF && e->dtype != WHOLEMEMORY_DT_BF16) {
    fprintf(stderr, "[wholegraph_amd] wholememory_embedding_set_optimizer: trainable tables are float / half / bf16\n");
    return WHOLEMEMORY_INVALID_INPUT;  // embedding_optimizer_func.cu:74-76
  }
  return guarded("wholememory_embedding_set_optimizer", [&] {
    switch (opt->type) {
      case WHOLEMEMORY_OPT_LAZY_ADAM: e->state_names = {"m", "v"}; break;
      case WHOLEMEMORY_OPT_ADAGRAD: e->state_names = {"state_sum"}; break;
      case WHOLEMEMORY_OPT_RMSPROP: e->state_names = {"v"}; break;
      default: e->state_names.clear(); break;
    }
    e->state_dim           = (e->dim + 3) / 4 * 4;
    std::vector<size_t> pt = entry_partition(e->allocated);
    const int n_states     = (int)e->state_names.size();
    auto fail_if = [&](wholememory_error_code_t rc, const char* what) {
      if (rc != WHOLEMEMORY_SUCCESS) {
        destroy_states(e);
        e->state_names.clear();
        throw logic_error(fmt("%s failed (%d)", what, (int)rc));
      }
    };
    if (n_states > 0) {
      wholememory_tensor_description_t d;
      wholememory_initialize_tensor_desc(&d);
      d.dim        = 2;
      d.dtype      = WHOLEMEMORY_DT_FLOAT;
      d.sizes[0]   = e->entries;
      d.sizes[1]   = n_states * e->state_dim;
      d.strides[0] = d.sizes[1];
      d.strides[1] = 1;
      fail_if(wholememory_create_tensor(&e->state_table, &d, e->comm, WHOLEMEMORY_MT_DISTRIBUTED, e->location, pt.data()),
              "state table allocation");
      size_t bytes = 0;
      void* p      = local_pointer(e->state_table, &bytes);
      if (bytes) WG_HIP_CHECK(hipMemsetAsync(p, 0, bytes, nullptr));  // embedding_optimizer.cpp:203-207: states start at zero
      for (int k = 0; k < n_states; k++) {
        int64_t starts[2] = {0, k * e->state_dim}, ends[2] = {-1, k * e->state_dim + e->dim};
        wholememory_tensor_t view = nullptr;
        fail_if(wholememory_tensor_get_subtensor(e->state_table, starts, ends, &view), "state view");
        e->state_views.push_back(view);
      }
    }
    if (opt->type == WHOLEMEMORY_OPT_LAZY_ADAM) {
      wholememory_tensor_description_t d;
      wholememory_initialize_tensor_desc(&d);
      d.dim        = 2;
      d.dtype      = WHOLEMEMORY_DT_FLOAT;
      d.sizes[0]   = e->entries;
      d.sizes[1]   = 2;
      d.strides[0] = 2;
      d.strides[1] = 1;
      fail_if(wholememory_create_tensor(&e->row_state, &d, e->comm, WHOLEMEMORY_MT_DISTRIBUTED, e->location, pt.data()),
              "per-row state allocation");
      size_t bytes = 0;
      auto* p      = static_cast<float*>(local_pointer(e->row_state, &bytes));
      if (bytes) {  // embedding_optimizer.cpp:208-209: beta1^0 = beta2^0 = 1
        fill_f32_kernel<<<256, 256, 0, nullptr>>>(p, (int64_t)(bytes / 4), 1.0f);
        WG_HIP_CHECK(hipGetLastError());
      }
      e->state_names.push_back("beta12t");
      e->state_views.push_back(nullptr);  // the whole per-row table
    }
    WG_HIP_CHECK(hipDeviceSynchronize());
    if (e->cache_rw) try {
      // the cache lines grow a state part behind the same tags (embedding.cpp:437-470: the reference's state embedding is
      // created with the embedding's cache policy).  Lines resident now have no state loaded: start from an empty cache.
      cache_view& c = e->cache;
      rw_writeback(e, /*drop=*/true, nullptr);
      c.st_floats = n_states * (int)e->state_dim;
      c.rs_off    = opt->type == WHOLEMEMORY_OPT_LAZY_ADAM ? c.st_floats : -1;
      c.st_stride = c.st_floats + (c.rs_off >= 0 ? 4 : 0);
      if (c.st_stride > 0) WG_HIP_CHECK(hipMalloc(&c.st, (size_t)c.n_sets * kCacheWays * (size_t)c.st_stride * sizeof(float)));
    } catch (...) {
      destroy_states(e);
      e->state_names.clear();
      e->cache.st_floats = 0, e->cache.st_stride = 0, e->cache.rs_off = -1;
      throw;
    }
    e->names_c.clear();
    for (auto& s : e->state_names) e->names_c.push_back(s.c_str());
    e->names_c.push_back(nullptr);
    e->optimizer = opt;
  });
}

wholememory_error_code_t wholememory_embedding_gather(wholememory_embedding_t e, wholememory_tensor_t indices,
                                                      wholememory_tensor_t output, bool adjust_cache,
                                                      wholememory_env_func_t* p_env_fns, int64_t stream_int)
{
  if (!e) return WHOLEMEMORY_INVALID_INPUT;
  if (e->cache_rw)
    return guarded("wholememory_embedding_gather", [&] {
      rw_cached_gather(e, indices, output, adjust_cache, p_env_fns, reinterpret_cast<hipStream_t>(stream_int));
    });
  if (e->cached)
    return guarded("wholememory_embedding_gather", [&] {
      cached_gather(e, indices, output, adjust_cache, p_env_fns, reinterpret_cast<hipStream_t>(stream_int));
    });
  return wholememory_gather(e->user, indices, output, p_env_fns, reinterpret_cast<void*>(stream_int), -1);
}

wholememory_error_code_t wholememory_embedding_gather_gradient_apply(wholememory_embedding_t e, wholememory_tensor_t indices,
                                                                     wholememory_tensor_t grads, bool adjust_cache,
                                                                     float lr, wholememory_env_func_t* p_env_fns,
                                                                     int64_t stream_int)
{
  return guarded("wholememory_embedding_gather_gradient_apply",
                 [&] { step(e, indices, grads, adjust_cache, lr, p_env_fns, reinterpret_cast<hipStream_t>(stream_int)); });
}

const char* const* wholememory_embedding_get_optimizer_state_names(wholememory_embedding_t e)
{
  return e ? e->names_c.data() : nullptr;
}

wholememory_tensor_t wholememory_embedding_get_optimizer_state(wholememory_embedding_t e, const char* name)
{
  if (!e || !name) return nullptr;
  for (size_t k = 0; k < e->state_names.size(); k++)
    if (e->state_names[k] == name) return e->state_views[k] ? e->state_views[k] : e->row_state;
  return nullptr;
}

wholememory_error_code_t wholememory_embedding_writeback_cache(wholememory_embedding_t e, int64_t stream_int)
{
  if (!e) return WHOLEMEMORY_INVALID_INPUT;
  if (!e->cache_rw) return WHOLEMEMORY_SUCCESS;   // a READONLY cache holds copies only
  return guarded("wholememory_embedding_writeback_cache",
                 [&] { rw_writeback(e, /*drop=*/false, reinterpret_cast<hipStream_t>(stream_int)); });
}

wholememory_error_code_t wholememory_embedding_drop_all_cache(wholememory_embedding_t e, int64_t stream_int)
{
  if (!e) return WHOLEMEMORY_INVALID_INPUT;
  if (!e->cached) return WHOLEMEMORY_SUCCESS;
  if (e->cache_rw)   // embedding_cache.cpp:321-331: write the modified lines back, then empty the cache
    return guarded("wholememory_embedding_drop_all_cache",
                   [&] { rw_writeback(e, /*drop=*/true, reinterpret_cast<hipStream_t>(stream_int)); });
  return guarded("wholememory_embedding_drop_all_cache", [&] {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_int);
    cache_clear(e->cache, stream);
    WG_HIP_CHECK(hipStreamSynchronize(stream));
  });
}

wholememory_error_code_t wgamd_embedding_cache_stats(wholememory_embedding_t e, int64_t* hits, int64_t* lookups, int64_t* lines)
{
  if (!e || !hits || !lookups) return WHOLEMEMORY_INVALID_INPUT;
  *hits = *lookups = 0;
  if (lines) *lines = 0;
  if (!e->cached) return WHOLEMEMORY_SUCCESS;
  return guarded("wgamd_embedding_cache_stats", [&] {
    unsigned long long h[2];
    WG_HIP_CHECK(hipMemcpy(h, e->cache.stats, sizeof(h), hipMemcpyDeviceToHost));
    *hits    = (int64_t)h[0];
    *lookups = (int64_t)h[1];
    if (lines) *lines = e->cache.n_sets * kCacheWays;
  });
}

}  // extern "C"
