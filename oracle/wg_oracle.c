/*
 * wg_oracle.c — CPU restatement of the cugraph-gnn (WholeGraph) mini-batch hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cugraph-gnn_amd/ may include, link or call this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only
 * as the checker / reported CPU baseline.
 *
 * Every function restates, in plain sequential C, the normative host code that the
 * reference's OWN tests compare the device ops against (all paths relative to
 * /root/reference):
 *   - RNG               cpp/src/wholegraph_ops/raft_random_gen.cu:15-97 (call sites) and
 *                       raft::random::detail::PCGenerator (rapidsai/raft 26.10, NOT vendored
 *                       in the reference tree -> restated from its published algorithm, PCG32
 *                       XSH-RR; "parity unpinned" at that boundary)
 *   - uniform sampling  cpp/tests/wholegraph_ops/graph_sampling_test_utils.cu:226-250 (offsets),
 *                       :252-291 (sample-all), :295-310 (swap-table selection), :312-401
 *                       (RNG -> index mapping, tables :344-350); M>1024 reservoir from the
 *                       device kernel cpp/src/wholegraph_ops/unweighted_sample_without_replacement_func.cuh:50-113
 *   - weighted sampling cpp/tests/wholegraph_ops/graph_sampling_test_utils.cu:530-656
 *   - append_unique     cpp/tests/graph_ops/append_unique_test_utils.cu:52-157
 *   - csr_add_self_loop cpp/src/graph_ops/csr_add_self_loop_func.cuh:13-32
 *   - gather / scatter  cpp/src/wholememory_ops/functions/gather_scatter_func.cuh:242-305,508-587
 *   - SAGE mean / GAT   torch_geometric.nn.{SAGEConv,GATConv} public formulas (third party,
 *                       not in tree; call sites python/pylibwholegraph/pylibwholegraph/torch/gnn_model.py:25-59)
 *
 * Build: make -C oracle   (gcc -O2 -fno-fast-math -ffp-contract=off, optional -fopenmp)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * A.1  RNG: PCG32 XSH-RR as wrapped by raft::random::detail::PCGenerator
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  uint64_t state;
  uint64_t inc;
} wgo_pcg_t;

static inline uint32_t wgo_pcg_u32(wgo_pcg_t* g)
{
  uint64_t old = g->state;
  g->state     = old * 6364136223846793005ULL + g->inc;
  uint32_t x   = (uint32_t)(((old >> 18u) ^ old) >> 27u);
  uint32_t rot = (uint32_t)(old >> 59u);
  return (x >> rot) | (x << ((-rot) & 31u));
}

/* LCG jump-ahead by `delta` steps (Brown, "Random number generation with arbitrary strides"). */
static inline void wgo_pcg_skipahead(wgo_pcg_t* g, uint64_t delta)
{
  uint64_t acc_mult = 1u, acc_plus = 0u;
  uint64_t cur_mult = 6364136223846793005ULL, cur_plus = g->inc;
  while (delta) {
    if (delta & 1u) {
      acc_mult *= cur_mult;
      acc_plus = acc_plus * cur_mult + cur_plus;
    }
    cur_plus = (cur_mult + 1u) * cur_plus;
    cur_mult *= cur_mult;
    delta >>= 1u;
  }
  g->state = acc_mult * g->state + acc_plus;
}

/* pcg32_srandom(seed, subsequence) followed by skipahead(offset): raft's (seed,subseq,offset) ctor */
void wgo_pcg_init_raw(wgo_pcg_t* g, uint64_t seed, uint64_t subsequence, uint64_t offset)
{
  g->state = 0u;
  g->inc   = (subsequence << 1u) | 1u;
  (void)wgo_pcg_u32(g);
  g->state += seed;
  (void)wgo_pcg_u32(g);
  wgo_pcg_skipahead(g, offset);
}

/* raft's PCGenerator(DeviceState{seed, base_subsequence = 0}, subsequence) ctor, the one every
 * call site in the reference uses (e.g. raft_random_gen.cu:32-34).  ASSUMPTION A1:
 * that ctor seeds stream `subsequence` and then skips ahead by `subsequence` draws. */
void wgo_pcg_init(wgo_pcg_t* g, uint64_t seed, uint64_t subsequence)
{
  wgo_pcg_init_raw(g, seed, subsequence, subsequence);
}

static inline int32_t wgo_pcg_i31(wgo_pcg_t* g) { return (int32_t)(wgo_pcg_u32(g) & 0x7fffffffu); }
static inline uint64_t wgo_pcg_u64(wgo_pcg_t* g)
{
  uint64_t lo = wgo_pcg_u32(g);
  uint64_t hi = wgo_pcg_u32(g);
  return lo | (hi << 32u);
}
static inline int64_t wgo_pcg_i63(wgo_pcg_t* g)
{
  return (int64_t)(wgo_pcg_u64(g) & 0x7fffffffffffffffULL);
}
static inline float wgo_pcg_f32(wgo_pcg_t* g)
{
  return (float)(wgo_pcg_u32(g) >> 8u) / 16777216.0f;
}

/* raw draws, for the known-answer tests */
void wgo_pcg_raw_u32(uint64_t seed, uint64_t subsequence, uint64_t offset, uint32_t* out, int64_t n)
{
  wgo_pcg_t g;
  wgo_pcg_init_raw(&g, seed, subsequence, offset);
  for (int64_t i = 0; i < n; i++) out[i] = wgo_pcg_u32(&g);
}

/* generate_random_positive_int_cpu  (raft_random_gen.cu:15-54) */
void wgo_generate_random_positive_int(int64_t seed, int64_t subsequence, void* out, int64_t n, int is64)
{
  wgo_pcg_t g;
  wgo_pcg_init(&g, (uint64_t)seed, (uint64_t)subsequence);
  for (int64_t i = 0; i < n; i++) {
    if (is64)
      ((int64_t*)out)[i] = wgo_pcg_i63(&g);
    else
      ((int32_t*)out)[i] = wgo_pcg_i31(&g);
  }
}

static inline int wgo_clz64(uint64_t x)
{
  int c = 0;
  while (x) {
    x >>= 1u;
    c++;
  }
  return 64 - c;
}

/* generate_exponential_distribution_negative_float_cpu (raft_random_gen.cu:56-97); double math */
void wgo_generate_exponential_distribution_negative_float(int64_t seed,
                                                          int64_t subsequence,
                                                          float* out,
                                                          int64_t n)
{
  wgo_pcg_t g;
  wgo_pcg_init(&g, (uint64_t)seed, (uint64_t)subsequence);
  for (int64_t i = 0; i < n; i++) {
    float u = wgo_pcg_f32(&g);
    u       = (float)(-(0.5 + 0.5 * (double)u));
    uint64_t x;
    int zero_draws = -1;
    do {
      x = wgo_pcg_u64(&g);
      zero_draws++;
    } while (!x);
    int one_bit = wgo_clz64(x) + zero_draws * 64;
    u           = (float)((double)u * pow(2.0, -one_bit));
    out[i]      = (float)(log1p((double)u) / log(2.0));
  }
}

/* A-Res key (graph_sampling_test_utils.cu:540-557): fp32 math */
static inline float wgo_key_from_weight(float weight_as_float, wgo_pcg_t* g)
{
  float u = wgo_pcg_f32(g);
  u       = (float)(-(0.5 + 0.5 * (double)u));
  uint64_t x;
  int zero_draws = -1;
  do {
    x = wgo_pcg_u64(g);
    zero_draws++;
  } while (!x);
  int one_bit = wgo_clz64(x) + zero_draws * 64;
  u *= exp2f((float)(-one_bit));
  return (log1pf(u) / logf(2.0f)) * (1.0f / weight_as_float);
}

void wgo_weighted_keys(int64_t seed, int64_t subsequence, const float* w, float* out, int64_t n)
{
  wgo_pcg_t g;
  wgo_pcg_init(&g, (uint64_t)seed, (uint64_t)subsequence);
  for (int64_t i = 0; i < n; i++) out[i] = wgo_key_from_weight(w[i], &g);
}

/* ------------------------------------------------------------------------------------------
 * A.2  one-hop sampling
 * ---------------------------------------------------------------------------------------- */
static inline int64_t wgo_idx(const void* p, int is64, int64_t i)
{
  return is64 ? ((const int64_t*)p)[i] : (int64_t)((const int32_t*)p)[i];
}
static inline void wgo_set(void* p, int is64, int64_t i, int64_t v)
{
  if (is64)
    ((int64_t*)p)[i] = v;
  else
    ((int32_t*)p)[i] = (int32_t)v;
}

/* host_get_sample_offset + prefix sum (graph_sampling_test_utils.cu:226-250); offsets[n+1]; returns total */
int wgo_sample_offsets(const int64_t* row_ptr,
                       const void* seeds,
                       int seeds_is64,
                       int64_t n,
                       int max_sample_count,
                       int32_t* offsets)
{
  int32_t acc = 0;
  for (int64_t i = 0; i < n; i++) {
    int64_t nid = wgo_idx(seeds, seeds_is64, i);
    int cnt     = (int)(row_ptr[nid + 1] - row_ptr[nid]);
    if (max_sample_count > 0 && cnt > max_sample_count) cnt = max_sample_count;
    offsets[i] = acc;
    acc += cnt;
  }
  offsets[n] = acc;
  return acc;
}

static const int k_warp_count[32] = {1, 1, 1, 2, 2, 2, 4, 4, 4, 4, 4, 4, 8, 8, 8, 8,
                                     8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8};
static const int k_items[32]      = {1, 2, 3, 2, 3, 3, 2, 2, 3, 3, 3, 3, 2, 2, 2, 2,
                                     3, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4};

/* random_sample_without_replacement_cpu_base (graph_sampling_test_utils.cu:295-310), dense Q */
static void wgo_swap_table_select(int* a, const int32_t* r, int M, int N, int* Q)
{
  for (int i = 0; i < N; i++) Q[i] = i;
  for (int i = 0; i < M; i++) {
    a[i]    = Q[r[i]];
    Q[r[i]] = Q[N - i - 1];
  }
}

/* One seed of the uniform sampler.  `i` is the seed's index in the call (= block index). */
static void wgo_uniform_one(const int64_t* row_ptr,
                            const void* col,
                            int col_is64,
                            int64_t nid,
                            int64_t i,
                            int M,
                            uint64_t random_seed,
                            int out_base,
                            void* dst,
                            int32_t* src_lid,
                            int64_t* edge_gid)
{
  int64_t start = row_ptr[nid], end = row_ptr[nid + 1];
  int N = (int)(end - start);
  if (N <= 0) return;
  if (M <= 0 || N <= M) {
    for (int j = 0; j < N; j++) {
      wgo_set(dst, col_is64, out_base + j, wgo_idx(col, col_is64, start + j));
      if (src_lid) src_lid[out_base + j] = (int32_t)i;
      if (edge_gid) edge_gid[out_base + j] = start + j;
    }
    return;
  }
  if (M > 1024) {
    /* reservoir with atomicMax (…_func.cuh:50-113): block = 32 threads, thread j draws in
     * sequence for idx = M+j, M+j+32, …; slot s ends holding max{idx : draw % (idx+1) == s},
     * or s itself when nobody hit it. */
    int* slot = (int*)malloc(sizeof(int) * (size_t)M);
    for (int s = 0; s < M; s++) slot[s] = s;
    for (int j = 0; j < 32; j++) {
      wgo_pcg_t g;
      wgo_pcg_init(&g, random_seed, (uint64_t)(int64_t)(int32_t)(i * 32 + j));
      for (int idx = M + j; idx < N; idx += 32) {
        int32_t rn = wgo_pcg_i31(&g) % (idx + 1);
        if (rn < M && slot[rn] < idx) slot[rn] = idx;
      }
    }
    for (int s = 0; s < M; s++) {
      wgo_set(dst, col_is64, out_base + s, wgo_idx(col, col_is64, start + slot[s]));
      if (src_lid) src_lid[out_base + s] = (int32_t)i;
      if (edge_gid) edge_gid[out_base + s] = start + slot[s];
    }
    free(slot);
    return;
  }
  int f = (M - 1) / 32;
  int B = k_warp_count[f] * 32, items = k_items[f];
  int32_t* r = (int32_t*)malloc(sizeof(int32_t) * (size_t)(B * items));
  for (int j = 0; j < B; j++) {
    wgo_pcg_t g;
    /* device: int gidx = threadIdx.x + blockIdx.x*blockDim.x; PCGenerator(rngstate,(uint64_t)gidx) */
    wgo_pcg_init(&g, random_seed, (uint64_t)(int64_t)(int32_t)(i * B + j));
    for (int k = 0; k < items; k++) {
      int id     = k * B + j;
      int32_t rn = wgo_pcg_i31(&g);
      if (id < M && id < N) r[id] = rn % (N - id); /* draws with id >= M are made and discarded */
    }
  }
  int* a = (int*)malloc(sizeof(int) * (size_t)M);
  int* Q = NULL;
  if (M <= 128 && N > 8 * M) {
    /* Same selection without materialising Q[0..N): only the <= M positions written so far can
     * differ from the identity, so keep them in a small list (latest write wins).  Identical
     * result; keeps the CPU baseline from paying O(deg) per hub. */
    int wpos[128], wval[128], nw = 0;
    for (int t = 0; t < M; t++) {
      int rt = r[t], tail = N - t - 1, q_rt = rt, q_tail = tail;
      for (int s = nw - 1; s >= 0; s--)
        if (wpos[s] == rt) { q_rt = wval[s]; break; }
      for (int s = nw - 1; s >= 0; s--)
        if (wpos[s] == tail) { q_tail = wval[s]; break; }
      a[t]     = q_rt;
      wpos[nw] = rt;
      wval[nw] = q_tail;
      nw++;
    }
  } else {
    Q = (int*)malloc(sizeof(int) * (size_t)N);
    wgo_swap_table_select(a, r, M, N, Q);
  }
  for (int t = 0; t < M; t++) {
    wgo_set(dst, col_is64, out_base + t, wgo_idx(col, col_is64, start + a[t]));
    if (src_lid) src_lid[out_base + t] = (int32_t)i;
    if (edge_gid) edge_gid[out_base + t] = start + a[t];
  }
  free(Q);
  free(a);
  free(r);
}

/* Uniform sampling WITH replacement.  NOT in the reference tree (cugraph_pyg forwards `with_replacement` to libcugraph,
 * sampler/distributed_sampler.py:775-792,864): parity unpinned against the reference; this is the normative statement of
 * the layout csrc/wg_sample_replace.hip implements — a seed with N > 0 neighbours yields exactly M picks, pick t =
 * col[start + G(seed64, i*M + t).i31() % N] (one stream per draw, the op generator of A.1), a seed without neighbours
 * none; offsets[i] = M x #{j < i : N_j > 0} (n + 1 entries).  src_lid / edge_gid may be NULL. */
void wgo_uniform_sample_with_replacement(const int64_t* row_ptr, const void* col, int col_is64, const void* seeds, int seeds_is64,
                                         int64_t n, int M, uint64_t random_seed, int32_t* offsets, void* dst, int32_t* src_lid,
                                         int64_t* edge_gid)
{
  int32_t acc = 0;
  for (int64_t i = 0; i < n; i++) {
    int64_t nid = wgo_idx(seeds, seeds_is64, i);
    offsets[i]  = acc;
    if (row_ptr[nid + 1] > row_ptr[nid]) acc += M;
  }
  offsets[n] = acc;
  if (dst == NULL) return;
  for (int64_t i = 0; i < n; i++) {
    int64_t nid   = wgo_idx(seeds, seeds_is64, i);
    int64_t start = row_ptr[nid], N = row_ptr[nid + 1] - start;
    if (N <= 0) continue;
    for (int t = 0; t < M; t++) {
      wgo_pcg_t g;
      int64_t sid = i * (int64_t)M + t;
      wgo_pcg_init(&g, random_seed, (uint64_t)sid);
      int64_t a   = start + (int64_t)(wgo_pcg_i31(&g) % N);
      int64_t out = offsets[i] + t;
      wgo_set(dst, col_is64, out, wgo_idx(col, col_is64, a));
      if (src_lid) src_lid[out] = (int32_t)i;
      if (edge_gid) edge_gid[out] = a;
    }
  }
}

/* host_unweighted_sample_without_replacement (graph_sampling_test_utils.cu:312-401).
 * `offsets` must come from wgo_sample_offsets.  src_lid / edge_gid may be NULL.
 * Seeds are independent -> optional OpenMP over seeds (used only by bench.py's cpu_baseline). */
void wgo_unweighted_sample(const int64_t* row_ptr,
                           const void* col,
                           int col_is64,
                           const void* seeds,
                           int seeds_is64,
                           int64_t n,
                           int max_sample_count,
                           uint64_t random_seed,
                           const int32_t* offsets,
                           void* dst,
                           int32_t* src_lid,
                           int64_t* edge_gid)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64)
#endif
  for (int64_t i = 0; i < n; i++) {
    wgo_uniform_one(row_ptr,
                    col,
                    col_is64,
                    wgo_idx(seeds, seeds_is64, i),
                    i,
                    max_sample_count,
                    random_seed,
                    offsets[i],
                    dst,
                    src_lid,
                    edge_gid);
  }
}

/* ------------------------------------------------------------------------------------------
 * A.3  weighted (A-Res) sampling  (graph_sampling_test_utils.cu:559-656)
 * Output order inside a seed: the reference leaves it unspecified (its tests sort per segment);
 * this restatement selects the M largest keys (ties: lower neighbour index wins) and emits the
 * selected edges in CSR order (neighbour index ascending).
 * `keys_out` (optional, same length as dst) receives the key of every sampled edge (NaN-free;
 * 0 for rows copied whole) so that tests can reason about 1-ulp libm differences.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  float key;
  int idx;
} wgo_kv_t;

static int wgo_kv_cmp_idx(const void* pa, const void* pb)
{
  const wgo_kv_t* a = (const wgo_kv_t*)pa;
  const wgo_kv_t* b = (const wgo_kv_t*)pb;
  return (a->idx > b->idx) - (a->idx < b->idx);
}

static int wgo_kv_cmp_desc(const void* pa, const void* pb)
{
  const wgo_kv_t* a = (const wgo_kv_t*)pa;
  const wgo_kv_t* b = (const wgo_kv_t*)pb;
  if (a->key > b->key) return -1;
  if (a->key < b->key) return 1;
  return (a->idx > b->idx) - (a->idx < b->idx);
}

void wgo_weighted_sample(const int64_t* row_ptr,
                         const void* col,
                         int col_is64,
                         const void* weights,
                         int w_is_double,
                         const void* seeds,
                         int seeds_is64,
                         int64_t n,
                         int max_sample_count,
                         uint64_t random_seed,
                         const int32_t* offsets,
                         void* dst,
                         int32_t* src_lid,
                         int64_t* edge_gid,
                         float* keys_out)
{
  int M = max_sample_count;
  int B = (M > 256) ? 256 : 128;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64)
#endif
  for (int64_t i = 0; i < n; i++) {
    int64_t nid   = wgo_idx(seeds, seeds_is64, i);
    int64_t start = row_ptr[nid], end = row_ptr[nid + 1];
    int N    = (int)(end - start);
    int base = offsets[i];
    if (N <= 0) continue;
    if (M <= 0 || N <= M) {
      for (int j = 0; j < N; j++) {
        wgo_set(dst, col_is64, base + j, wgo_idx(col, col_is64, start + j));
        if (src_lid) src_lid[base + j] = (int32_t)i;
        if (edge_gid) edge_gid[base + j] = start + j;
        if (keys_out) keys_out[base + j] = 0.0f;
      }
      continue;
    }
    wgo_kv_t* kv = (wgo_kv_t*)malloc(sizeof(wgo_kv_t) * (size_t)N);
    for (int j = 0; j < B; j++) {
      wgo_pcg_t g;
      wgo_pcg_init(&g, random_seed, (uint64_t)(int64_t)(int32_t)(i * B + j));
      for (int id = j; id < N; id += B) {
        float w = w_is_double ? (float)((const double*)weights)[start + id]
                              : ((const float*)weights)[start + id];
        kv[id].key = wgo_key_from_weight(w, &g);
        kv[id].idx = id;
      }
    }
    qsort(kv, (size_t)N, sizeof(wgo_kv_t), wgo_kv_cmp_desc);
    qsort(kv, (size_t)M, sizeof(wgo_kv_t), wgo_kv_cmp_idx);
    for (int t = 0; t < M; t++) {
      wgo_set(dst, col_is64, base + t, wgo_idx(col, col_is64, start + kv[t].idx));
      if (src_lid) src_lid[base + t] = (int32_t)i;
      if (edge_gid) edge_gid[base + t] = start + kv[t].idx;
      if (keys_out) keys_out[base + t] = kv[t].key;
    }
    free(kv);
  }
}

/* ------------------------------------------------------------------------------------------
 * A.4  append_unique  (append_unique_test_utils.cu:52-157): targets verbatim, then neighbours
 * not among the targets in FIRST-APPEARANCE order (the order the reference's host code
 * produces; its device op leaves the order of that tail unspecified).  Returns U (new nodes).
 * unique_out has capacity T+E; map_out (nullable) has E entries.
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t wgo_mix(uint64_t k)
{
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

int wgo_append_unique(const void* targets,
                      int64_t T,
                      const void* neighbors,
                      int64_t E,
                      int is64,
                      void* unique_out,
                      int32_t* map_out)
{
  uint64_t cap = 16;
  while (cap < (uint64_t)(2 * (T + E) + 2)) cap <<= 1;
  int64_t* keys = (int64_t*)malloc(sizeof(int64_t) * cap);
  int32_t* vals = (int32_t*)malloc(sizeof(int32_t) * cap);
  uint8_t* used = (uint8_t*)calloc(cap, 1);
  int32_t count = (int32_t)T;
  for (int64_t t = 0; t < T; t++) {
    int64_t k = wgo_idx(targets, is64, t);
    wgo_set(unique_out, is64, t, k);
    uint64_t h = wgo_mix((uint64_t)k) & (cap - 1);
    while (used[h] && keys[h] != k) h = (h + 1) & (cap - 1);
    if (!used[h]) { /* std::unordered_map::insert: first occurrence wins */
      used[h] = 1;
      keys[h] = k;
      vals[h] = (int32_t)t;
    }
  }
  for (int64_t e = 0; e < E; e++) {
    int64_t k  = wgo_idx(neighbors, is64, e);
    uint64_t h = wgo_mix((uint64_t)k) & (cap - 1);
    while (used[h] && keys[h] != k) h = (h + 1) & (cap - 1);
    if (!used[h]) {
      used[h] = 1;
      keys[h] = k;
      vals[h] = count;
      wgo_set(unique_out, is64, count, k);
      count++;
    }
    if (map_out) map_out[e] = vals[h];
  }
  free(used);
  free(vals);
  free(keys);
  return count - (int32_t)T;
}

/* ------------------------------------------------------------------------------------------
 * csr_add_self_loop (csr_add_self_loop_func.cuh:13-32): row i -> [i] ++ row i.  int32 only.
 * ---------------------------------------------------------------------------------------- */
void wgo_csr_add_self_loop(const int32_t* row_ptr, const int32_t* col, int64_t n_rows, int32_t* out_row_ptr, int32_t* out_col)
{
  for (int64_t i = 0; i < n_rows; i++) {
    int32_t s = row_ptr[i], e = row_ptr[i + 1];
    out_row_ptr[i]          = s + (int32_t)i;
    out_col[s + (int32_t)i] = (int32_t)i;
    for (int32_t j = s; j < e; j++) out_col[j + (int32_t)i + 1] = col[j];
  }
  out_row_ptr[n_rows] = row_ptr[n_rows] + (int32_t)n_rows;
}

/* ------------------------------------------------------------------------------------------
 * gather / scatter of rows, same dtype (byte copy); negative index => row skipped
 * (gather_scatter_func.cuh:285).  Strides in BYTES here.
 * ---------------------------------------------------------------------------------------- */
void wgo_gather_rows(const uint8_t* table,
                     int64_t table_stride_bytes,
                     const void* idx,
                     int idx_is64,
                     int64_t n,
                     int64_t row_bytes,
                     uint8_t* out,
                     int64_t out_stride_bytes)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int64_t i = 0; i < n; i++) {
    int64_t r = wgo_idx(idx, idx_is64, i);
    if (r < 0) continue;
    memcpy(out + i * out_stride_bytes, table + r * table_stride_bytes, (size_t)row_bytes);
  }
}

void wgo_scatter_rows(const uint8_t* in,
                      int64_t in_stride_bytes,
                      const void* idx,
                      int idx_is64,
                      int64_t n,
                      int64_t row_bytes,
                      uint8_t* table,
                      int64_t table_stride_bytes)
{
  for (int64_t i = 0; i < n; i++) {
    int64_t r = wgo_idx(idx, idx_is64, i);
    if (r < 0) continue;
    memcpy(table + r * table_stride_bytes, in + i * in_stride_bytes, (size_t)row_bytes);
  }
}

/* ------------------------------------------------------------------------------------------
 * Aggregation (third-party semantics, PyG formulas):
 *  SAGE mean:  out[i,:] = (sum_{e in row i} x[col[e],:]) / max(deg_i,1)     (CSR order, fp32)
 *  GAT:        s_e = leaky_relu(a_src[col[e],h] + a_dst[i,h], slope); alpha = softmax_e(s_e)
 *              out[i,h,:] = sum_e alpha_e * x[col[e],h,:]
 * acc_double != 0 accumulates in fp64 (tolerance reference); else fp32 sequential (bit pattern
 * the kernel is expected to reproduce when it sums in CSR order).
 * ---------------------------------------------------------------------------------------- */
void wgo_spmm_csr(const int32_t* row_ptr,
                  const int32_t* col,
                  int64_t n_rows,
                  const float* x,
                  int64_t F,
                  int64_t ldx,
                  int mean,
                  int acc_double,
                  float* out,
                  int64_t ldo)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256)
#endif
  for (int64_t i = 0; i < n_rows; i++) {
    int32_t s = row_ptr[i], e = row_ptr[i + 1];
    float denom = (mean && e > s) ? (float)(e - s) : 1.0f;
    for (int64_t f = 0; f < F; f++) {
      if (acc_double) {
        double acc = 0.0;
        for (int32_t j = s; j < e; j++) acc += (double)x[(int64_t)col[j] * ldx + f];
        out[i * ldo + f] = (float)(acc / (double)denom);
      } else {
        float acc = 0.0f;
        for (int32_t j = s; j < e; j++) acc += x[(int64_t)col[j] * ldx + f];
        out[i * ldo + f] = acc / denom;
      }
    }
  }
}

void wgo_gat_csr(const int32_t* row_ptr,
                 const int32_t* col,
                 int64_t n_rows,
                 const float* x, /* [N_src, H, C] */
                 const float* a_src, /* [N_src, H] */
                 const float* a_dst, /* [n_rows, H] */
                 int64_t H,
                 int64_t C,
                 float slope,
                 float* alpha_out, /* nullable [E, H] */
                 float* out /* [n_rows, H, C] */)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256)
#endif
  for (int64_t i = 0; i < n_rows; i++) {
    int32_t s = row_ptr[i], e = row_ptr[i + 1];
    for (int64_t h = 0; h < H; h++) {
      double mx = -INFINITY;
      for (int32_t j = s; j < e; j++) {
        double v = (double)a_src[(int64_t)col[j] * H + h] + (double)a_dst[i * H + h];
        v        = v > 0 ? v : v * (double)slope;
        if (v > mx) mx = v;
      }
      double den = 0.0;
      for (int32_t j = s; j < e; j++) {
        double v = (double)a_src[(int64_t)col[j] * H + h] + (double)a_dst[i * H + h];
        v        = v > 0 ? v : v * (double)slope;
        den += exp(v - mx);
      }
      for (int64_t c = 0; c < C; c++) {
        double acc = 0.0;
        for (int32_t j = s; j < e; j++) {
          double v = (double)a_src[(int64_t)col[j] * H + h] + (double)a_dst[i * H + h];
          v        = v > 0 ? v : v * (double)slope;
          double a = exp(v - mx) / den;
          acc += a * (double)x[((int64_t)col[j] * H + h) * C + c];
        }
        out[(i * H + h) * C + c] = (float)acc;
      }
      if (alpha_out) {
        for (int32_t j = s; j < e; j++) {
          double v = (double)a_src[(int64_t)col[j] * H + h] + (double)a_dst[i * H + h];
          v        = v > 0 ? v : v * (double)slope;
          alpha_out[(int64_t)j * H + h] = (float)(exp(v - mx) / den);
        }
      }
    }
  }
}

/* GAT aggregation BEFORE the dense transform (the restatement wgamd_gat_aggregate_heads_f32 is checked against; also what the
 * CPU baseline of bench_mag.py runs).  Same attention as wgo_gat_csr (torch_geometric.nn.GATConv semantics, SURVEY.md §8 row
 * a18: e = leaky_relu(a_src[j] + a_dst[i]), alpha = softmax over the row's edges, per head), but the weighted sum runs over
 * the UNTRANSFORMED source rows x [N_src, F]:  out[i, h, :] = sum_e alpha_e^h x[col[e], :]  — the attention-weighted sum is
 * linear, so W_h applied afterwards to out[i, h, :] gives GATConv's sum_e alpha_e^h (W x)[col[e], h, :].  fp64 accumulation.
 * a_dst row of output row i: dst_rows ? dst_rows[i] : i. */
void wgo_gat_aggregate_heads(const int32_t* row_ptr, const int32_t* col, int64_t n_rows, const float* x, int64_t F,
                             const float* a_src, const float* a_dst, const int64_t* dst_rows, int64_t H, float slope,
                             float* out /* [n_rows, H, F] */)
{
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
    int64_t cap   = 64;
    double* alpha = (double*)malloc(sizeof(double) * (size_t)cap);
    double* acc   = (double*)malloc(sizeof(double) * (size_t)F);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 256)
#endif
    for (int64_t i = 0; i < n_rows; i++) {
      const int32_t s = row_ptr[i], e = row_ptr[i + 1];
      const int64_t ar = dst_rows ? dst_rows[i] : i;
      if (e - s > cap) {
        cap   = 2 * (int64_t)(e - s);
        alpha = (double*)realloc(alpha, sizeof(double) * (size_t)cap);
      }
      for (int64_t h = 0; h < H; h++) {
        double mx = -INFINITY, den = 0.0;
        for (int32_t j = s; j < e; j++) {
          double v = (double)a_src[(int64_t)col[j] * H + h] + (double)a_dst[ar * H + h];
          v        = v > 0 ? v : v * (double)slope;
          alpha[j - s] = v;
          if (v > mx) mx = v;
        }
        for (int32_t j = s; j < e; j++) {
          alpha[j - s] = exp(alpha[j - s] - mx);
          den += alpha[j - s];
        }
        for (int64_t c = 0; c < F; c++) acc[c] = 0.0;
        for (int32_t j = s; j < e; j++) {
          const double a  = alpha[j - s] / den;
          const float* xr = x + (int64_t)col[j] * F;
          for (int64_t c = 0; c < F; c++) acc[c] += a * (double)xr[c];
        }
        float* o = out + (i * H + h) * F;
        for (int64_t c = 0; c < F; c++) o[c] = (float)acc[c];
      }
    }
    free(alpha);
    free(acc);
  }
}

int wgo_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void wgo_set_num_threads(int n)
{
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}
