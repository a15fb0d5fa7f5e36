/* Plain-C client of libwholegraph_amd.so (no Python, no torch): the drop-in boundary exercised the way a C/C++
 * caller of libwholegraph would — device buffers from hipMalloc, tensors from wholememory_make_tensor_from_pointer,
 * variable-size outputs through the DEFAULT wholememory_env_func_t callbacks (include/wgamd_types.h), results compared
 * on the host with the C oracle (oracle/wg_oracle.c, TEST INFRASTRUCTURE, linked only into this test binary).
 * Mirrors the shape of the reference's gtests (cpp/tests/wholegraph_ops/
 * wholegraph_csr_unweighted_sample_without_replacement_tests.cu:330-353, cpp/tests/graph_ops/append_unique_tests.cu:160-199,
 * cpp/tests/wholememory_ops/wholememory_gather_tests.cu).  Prints C_ABI_PARITY_OK on success. */
#include <hip/hip_runtime_api.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "wholegraph_amd.h"

/* oracle entry points (oracle/wg_oracle.c) */
int wgo_sample_offsets(const int64_t*, const void*, int, int64_t, int, int32_t*);
void wgo_unweighted_sample(const int64_t*, const void*, int, const void*, int, int64_t, int, uint64_t, const int32_t*, void*,
                           int32_t*, int64_t*);
int wgo_append_unique(const void*, int64_t, const void*, int64_t, int, void*, int32_t*);
void wgo_gather_rows(const uint8_t*, int64_t, const void*, int, int64_t, int64_t, uint8_t*, int64_t);

#define CHECK(cond)                                                         \
  do {                                                                      \
    if (!(cond)) {                                                          \
      fprintf(stderr, "%s:%d check failed: %s\n", __FILE__, __LINE__, #cond); \
      exit(1);                                                              \
    }                                                                       \
  } while (0)
#define HIP(x) CHECK((x) == hipSuccess)
#define WM(x) CHECK((x) == WHOLEMEMORY_SUCCESS)

static uint64_t lcg_state = 12345;
static uint32_t rnd(void)
{
  lcg_state = lcg_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return (uint32_t)(lcg_state >> 33);
}

static void* to_device(const void* host, size_t bytes)
{
  void* d = NULL;
  HIP(hipMalloc(&d, bytes ? bytes : 1));
  if (bytes) HIP(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
  return d;
}

static wholememory_tensor_t wrap1d(void* dev, int64_t n, wholememory_dtype_t dt)
{
  wholememory_tensor_description_t d;
  wholememory_initialize_tensor_desc(&d);
  d.dim = 1; d.sizes[0] = n; d.strides[0] = 1; d.dtype = dt;
  wholememory_tensor_t t = NULL;
  WM(wholememory_make_tensor_from_pointer(&t, dev, &d));
  return t;
}

static wholememory_tensor_t wrap2d(void* dev, int64_t rows, int64_t cols, int64_t stride, wholememory_dtype_t dt)
{
  wholememory_tensor_description_t d;
  wholememory_initialize_tensor_desc(&d);
  d.dim = 2; d.sizes[0] = rows; d.sizes[1] = cols; d.strides[0] = stride; d.strides[1] = 1; d.dtype = dt;
  wholememory_tensor_t t = NULL;
  WM(wholememory_make_tensor_from_pointer(&t, dev, &d));
  return t;
}

int main(void)
{
  /* ---- a power-law-ish CSR: V vertices, degrees 0..~300 ---------------------------------------- */
  const int64_t V = 5000, n_seeds = 1500;
  const int M = 10;
  int64_t* row_ptr = (int64_t*)malloc(sizeof(int64_t) * (V + 1));
  row_ptr[0] = 0;
  for (int64_t v = 0; v < V; v++) {
    uint32_t r = rnd() % 100;
    int deg    = r < 5 ? 0 : r < 60 ? (int)(rnd() % 8) : r < 95 ? (int)(rnd() % 40) : (int)(rnd() % 300);
    row_ptr[v + 1] = row_ptr[v] + deg;
  }
  const int64_t E = row_ptr[V];
  int64_t* col = (int64_t*)malloc(sizeof(int64_t) * E);
  for (int64_t e = 0; e < E; e++) col[e] = rnd() % V;
  int64_t* seeds = (int64_t*)malloc(sizeof(int64_t) * n_seeds);
  for (int64_t i = 0; i < n_seeds; i++) seeds[i] = (i * 7919) % V; /* distinct */

  wholememory_env_func_t* env = wholememory_get_default_env_func();
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));

  /* ---- a1: one hop of uniform sampling ---------------------------------------------------------- */
  void *d_row = to_device(row_ptr, sizeof(int64_t) * (V + 1)), *d_col = to_device(col, sizeof(int64_t) * E);
  void* d_seeds = to_device(seeds, sizeof(int64_t) * n_seeds);
  void* d_off   = NULL;
  HIP(hipMalloc(&d_off, sizeof(int) * (n_seeds + 1)));
  wholememory_tensor_t t_row = wrap1d(d_row, V + 1, WHOLEMEMORY_DT_INT64), t_col = wrap1d(d_col, E, WHOLEMEMORY_DT_INT64);
  wholememory_tensor_t t_seeds = wrap1d(d_seeds, n_seeds, WHOLEMEMORY_DT_INT64);
  wholememory_tensor_t t_off   = wrap1d(d_off, n_seeds + 1, WHOLEMEMORY_DT_INT);
  wgamd_default_memory_context_t *c_dst = wgamd_create_default_memory_context(), *c_lid = wgamd_create_default_memory_context(),
                                 *c_gid = wgamd_create_default_memory_context();
  const unsigned long long rs = 0x1234567ULL;
  WM(wholegraph_csr_unweighted_sample_without_replacement(t_row, t_col, t_seeds, M, t_off, c_dst, c_lid, c_gid, rs, env, stream));
  int32_t* off = (int32_t*)malloc(sizeof(int32_t) * (n_seeds + 1));
  HIP(hipMemcpy(off, d_off, sizeof(int32_t) * (n_seeds + 1), hipMemcpyDeviceToHost));
  const int64_t total = off[n_seeds];
  CHECK(c_dst->desc.dim == 1 && c_dst->desc.sizes[0] == total && c_dst->desc.dtype == WHOLEMEMORY_DT_INT64);
  CHECK(c_lid->desc.sizes[0] == total && c_lid->desc.dtype == WHOLEMEMORY_DT_INT);
  CHECK(c_gid->desc.sizes[0] == total && c_gid->desc.dtype == WHOLEMEMORY_DT_INT64);
  int64_t* dst  = (int64_t*)malloc(sizeof(int64_t) * total);
  int32_t* lid  = (int32_t*)malloc(sizeof(int32_t) * total);
  int64_t* gid  = (int64_t*)malloc(sizeof(int64_t) * total);
  HIP(hipMemcpy(dst, c_dst->ptr, sizeof(int64_t) * total, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(lid, c_lid->ptr, sizeof(int32_t) * total, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(gid, c_gid->ptr, sizeof(int64_t) * total, hipMemcpyDeviceToHost));
  int32_t* o_off = (int32_t*)malloc(sizeof(int32_t) * (n_seeds + 1));
  wgo_sample_offsets(row_ptr, seeds, 1, n_seeds, M, o_off);
  CHECK(memcmp(off, o_off, sizeof(int32_t) * (n_seeds + 1)) == 0);
  int64_t* o_dst = (int64_t*)malloc(sizeof(int64_t) * total);
  int32_t* o_lid = (int32_t*)malloc(sizeof(int32_t) * total);
  int64_t* o_gid = (int64_t*)malloc(sizeof(int64_t) * total);
  wgo_unweighted_sample(row_ptr, col, 1, seeds, 1, n_seeds, M, rs, o_off, o_dst, o_lid, o_gid);
  CHECK(memcmp(dst, o_dst, sizeof(int64_t) * total) == 0);
  CHECK(memcmp(lid, o_lid, sizeof(int32_t) * total) == 0);
  CHECK(memcmp(gid, o_gid, sizeof(int64_t) * total) == 0);
  printf("sample: %lld seeds -> %lld edges, bit-exact\n", (long long)n_seeds, (long long)total);

  /* ---- a7: renumber the hop ------------------------------------------------------------------------ */
  void* d_map = NULL;
  HIP(hipMalloc(&d_map, sizeof(int) * (total ? total : 1)));
  wholememory_tensor_t t_nbr = wrap1d(c_dst->ptr, total, WHOLEMEMORY_DT_INT64), t_map = wrap1d(d_map, total, WHOLEMEMORY_DT_INT);
  wgamd_default_memory_context_t* c_uniq = wgamd_create_default_memory_context();
  WM(graph_append_unique(t_seeds, t_nbr, c_uniq, t_map, env, stream));
  const int64_t n_uniq = c_uniq->desc.sizes[0];
  int64_t* uniq   = (int64_t*)malloc(sizeof(int64_t) * n_uniq);
  int32_t* map    = (int32_t*)malloc(sizeof(int32_t) * total);
  HIP(hipMemcpy(uniq, c_uniq->ptr, sizeof(int64_t) * n_uniq, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(map, d_map, sizeof(int32_t) * total, hipMemcpyDeviceToHost));
  int64_t* o_uniq = (int64_t*)malloc(sizeof(int64_t) * (n_seeds + total));
  int32_t* o_map  = (int32_t*)malloc(sizeof(int32_t) * total);
  const int o_n   = wgo_append_unique(seeds, n_seeds, dst, total, 1, o_uniq, o_map);
  CHECK(n_seeds + o_n == n_uniq); /* the oracle returns the number of NEW nodes */
  CHECK(memcmp(uniq, o_uniq, sizeof(int64_t) * n_uniq) == 0);
  CHECK(memcmp(map, o_map, sizeof(int32_t) * total) == 0);
  printf("append_unique: %lld targets + %lld neighbours -> %lld unique, bit-exact\n", (long long)n_seeds, (long long)total,
         (long long)n_uniq);

  /* ---- a10: feature fetch of the unique nodes (row stride > row length, one negative index) ------------- */
  const int64_t F = 100, stride = 104;
  float* table = (float*)malloc(sizeof(float) * V * stride);
  for (int64_t i = 0; i < V * stride; i++) table[i] = (float)(rnd() % 100000) * 0.25f;
  void* d_table = to_device(table, sizeof(float) * V * stride);
  uniq[3] = -1;
  void* d_idx = to_device(uniq, sizeof(int64_t) * n_uniq);
  float* out  = (float*)malloc(sizeof(float) * n_uniq * F);
  for (int64_t i = 0; i < n_uniq * F; i++) out[i] = -7.0f;
  void* d_out = to_device(out, sizeof(float) * n_uniq * F);
  wholememory_tensor_t t_table = wrap2d(d_table, V, F, stride, WHOLEMEMORY_DT_FLOAT);
  wholememory_tensor_t t_idx = wrap1d(d_idx, n_uniq, WHOLEMEMORY_DT_INT64), t_out = wrap2d(d_out, n_uniq, F, F, WHOLEMEMORY_DT_FLOAT);
  WM(wholememory_gather(t_table, t_idx, t_out, env, stream, -1));
  HIP(hipStreamSynchronize(stream));
  float* got = (float*)malloc(sizeof(float) * n_uniq * F);
  HIP(hipMemcpy(got, d_out, sizeof(float) * n_uniq * F, hipMemcpyDeviceToHost));
  wgo_gather_rows((const uint8_t*)table, stride * 4, uniq, 1, n_uniq, F * 4, (uint8_t*)out, F * 4);
  CHECK(memcmp(got, out, sizeof(float) * n_uniq * F) == 0); /* includes the untouched row of the negative index */
  printf("gather: %lld rows x %lld fp32, bit-exact\n", (long long)n_uniq, (long long)F);

  /* ---- (f4) trainable embedding: SGD step with duplicate indices, closed-form answer --------------------------
   * table[r, :] = r, every pair carries a gradient row of ones, lr = 0.5  =>  table[r, :] = r - 0.5 * (times r was hit).
   * World of one rank over the RCCL the library finds with dlopen (embedding.h:63-198 call sequence). */
  {
    wholememory_unique_id_t uid;
    wholememory_comm_t comm = NULL;
    WM(wholememory_init(0, 0));
    WM(wholememory_create_unique_id(&uid));
    WM(wholememory_create_communicator(&comm, uid, 0, 1));
    const int64_t rows = 3000, dim = 10, pairs = 7000; /* dim 10 -> rows padded to 12 floats */
    wholememory_tensor_description_t ed;
    wholememory_initialize_tensor_desc(&ed);
    ed.dim = 2; ed.sizes[0] = rows; ed.sizes[1] = dim; ed.strides[0] = dim; ed.strides[1] = 1; ed.dtype = WHOLEMEMORY_DT_FLOAT;
    wholememory_embedding_t emb = NULL;
    WM(wholememory_create_embedding(&emb, &ed, comm, WHOLEMEMORY_MT_DISTRIBUTED, WHOLEMEMORY_ML_DEVICE, NULL, NULL, -1, 0));
    wholememory_embedding_optimizer_t opt = NULL;
    CHECK(wholememory_create_embedding_optimizer(&opt, WHOLEMEMORY_OPT_NONE) == WHOLEMEMORY_NOT_IMPLEMENTED);
    WM(wholememory_create_embedding_optimizer(&opt, WHOLEMEMORY_OPT_SGD));
    float wd = 0.0f;
    WM(wholememory_optimizer_set_parameter(opt, "weight_decay", &wd));
    CHECK(wholememory_optimizer_set_parameter(opt, "beta1", &wd) == WHOLEMEMORY_INVALID_INPUT);
    WM(wholememory_embedding_set_optimizer(emb, opt));
    CHECK(wholememory_embedding_get_optimizer_state_names(emb)[0] == NULL);
    wholememory_tensor_t t_emb = wholememory_embedding_get_embedding_tensor(emb);
    CHECK(wholememory_tensor_get_tensor_description(t_emb)->sizes[1] == dim &&
          wholememory_tensor_get_tensor_description(t_emb)->strides[0] == 12);
    /* fill through wholememory_scatter, like a loader would */
    float* init = (float*)malloc(sizeof(float) * rows * dim);
    int64_t* all_rows = (int64_t*)malloc(sizeof(int64_t) * rows);
    for (int64_t r = 0; r < rows; r++) {
      all_rows[r] = r;
      for (int64_t j = 0; j < dim; j++) init[r * dim + j] = (float)r;
    }
    void *d_init = to_device(init, sizeof(float) * rows * dim), *d_all = to_device(all_rows, sizeof(int64_t) * rows);
    wholememory_tensor_t t_init = wrap2d(d_init, rows, dim, dim, WHOLEMEMORY_DT_FLOAT), t_all = wrap1d(d_all, rows, WHOLEMEMORY_DT_INT64);
    WM(wholememory_scatter(t_init, t_all, t_emb, env, stream, -1));
    int32_t* pidx  = (int32_t*)malloc(sizeof(int32_t) * pairs);
    int* hits      = (int*)calloc(rows, sizeof(int));
    float* ones    = (float*)malloc(sizeof(float) * pairs * dim);
    for (int64_t i = 0; i < pairs; i++) {
      pidx[i] = (i % 13 == 0) ? -1 : (int32_t)(rnd() % (rows / 2)); /* skipped pairs; the upper half is never hit */
      if (pidx[i] >= 0) hits[pidx[i]]++;
      for (int64_t j = 0; j < dim; j++) ones[i * dim + j] = 1.0f;
    }
    void *d_pidx = to_device(pidx, sizeof(int32_t) * pairs), *d_ones = to_device(ones, sizeof(float) * pairs * dim);
    wholememory_tensor_t t_pidx = wrap1d(d_pidx, pairs, WHOLEMEMORY_DT_INT), t_ones = wrap2d(d_ones, pairs, dim, dim, WHOLEMEMORY_DT_FLOAT);
    WM(wholememory_embedding_gather_gradient_apply(emb, t_pidx, t_ones, false, 0.5f, env, (int64_t)(intptr_t)stream));
    WM(wholememory_embedding_gather(emb, t_all, t_init, false, env, (int64_t)(intptr_t)stream));
    HIP(hipStreamSynchronize(stream));
    HIP(hipMemcpy(init, d_init, sizeof(float) * rows * dim, hipMemcpyDeviceToHost));
    for (int64_t r = 0; r < rows; r++)
      for (int64_t j = 0; j < dim; j++) CHECK(init[r * dim + j] == (float)r - 0.5f * (float)hits[r]);
    CHECK(wholememory_embedding_gather_gradient_apply(emb, t_pidx, t_init, false, 0.5f, env, (int64_t)(intptr_t)stream) ==
          WHOLEMEMORY_INVALID_INPUT); /* rows != pairs */
    WM(wholememory_embedding_writeback_cache(emb, 0));
    /* READONLY cache in front of a second table (embedding.h:96-144): same rows through the cache, twice; the second
     * pass is served from the cache lines.  Policy rules: ratio range, a READWRITE cache the table's addressing does not
     * cover (embedding.cpp:968-972), no optimizer on a READONLY cache. */
    wholememory_embedding_cache_policy_t pol = NULL, rw = NULL;
    CHECK(wholememory_create_embedding_cache_policy(&pol, comm, WHOLEMEMORY_MT_CHUNKED, WHOLEMEMORY_ML_DEVICE,
                                                    WHOLEMEMORY_AT_READONLY, 2.0f) == WHOLEMEMORY_INVALID_VALUE);
    WM(wholememory_create_embedding_cache_policy(&pol, comm, WHOLEMEMORY_MT_CHUNKED, WHOLEMEMORY_ML_DEVICE,
                                                 WHOLEMEMORY_AT_READONLY, 1.0f));
    WM(wholememory_create_embedding_cache_policy(&rw, comm, WHOLEMEMORY_MT_CHUNKED, WHOLEMEMORY_ML_DEVICE,
                                                 WHOLEMEMORY_AT_READWRITE, 0.5f));
    wholememory_embedding_t cached = NULL;
    CHECK(wholememory_create_embedding(&cached, &ed, comm, WHOLEMEMORY_MT_DISTRIBUTED, WHOLEMEMORY_ML_DEVICE, rw, NULL, -1, 0) ==
          WHOLEMEMORY_INVALID_INPUT);
    WM(wholememory_create_embedding(&cached, &ed, comm, WHOLEMEMORY_MT_DISTRIBUTED, WHOLEMEMORY_ML_DEVICE, pol, NULL, -1, 0));
    CHECK(wholememory_embedding_set_optimizer(cached, opt) == WHOLEMEMORY_INVALID_INPUT);
    for (int64_t r = 0; r < rows; r++)
      for (int64_t j = 0; j < dim; j++) init[r * dim + j] = (float)(r * 16 + j);
    HIP(hipMemcpy(d_init, init, sizeof(float) * rows * dim, hipMemcpyHostToDevice));
    WM(wholememory_scatter(t_init, t_all, wholememory_embedding_get_embedding_tensor(cached), env, stream, -1));
    float* through = (float*)malloc(sizeof(float) * rows * dim);
    void* d_through = to_device(through, sizeof(float) * rows * dim);
    wholememory_tensor_t t_through = wrap2d(d_through, rows, dim, dim, WHOLEMEMORY_DT_FLOAT);
    int64_t hits_c = 0, looked = 0, lines = 0;
    for (int pass = 0; pass < 2; pass++) {
      HIP(hipMemset(d_through, 0, sizeof(float) * rows * dim));
      WM(wholememory_embedding_gather(cached, t_all, t_through, true, env, (int64_t)(intptr_t)stream));
      HIP(hipStreamSynchronize(stream));
      HIP(hipMemcpy(through, d_through, sizeof(float) * rows * dim, hipMemcpyDeviceToHost));
      CHECK(memcmp(through, init, sizeof(float) * rows * dim) == 0);
      WM(wgamd_embedding_cache_stats(cached, &hits_c, &looked, &lines));
      CHECK(looked == rows * (pass + 1) && (pass == 0 ? hits_c == 0 : hits_c > rows / 2) && lines >= rows);
    }
    WM(wholememory_embedding_drop_all_cache(cached, 0));
    WM(wgamd_embedding_cache_stats(cached, &hits_c, &looked, NULL));
    CHECK(hits_c == 0 && looked == 0);
    WM(wholememory_destroy_tensor(t_through));
    WM(wholememory_destroy_embedding(cached));
    WM(wholememory_destroy_embedding_cache_policy(pol));
    WM(wholememory_destroy_embedding_cache_policy(rw));
    printf("cached embedding: %lld rows through a READONLY cache, bit-exact, second pass from the cache lines\n", (long long)rows);
    /* READWRITE device cache in front of a HOST-resident table (embedding.cpp:556-759): the table partition is pinned host
     * memory this C client writes with plain stores; gather -> SGD step through the cache -> the host table is stale until
     * writeback_cache and holds the closed form after it. */
    {
      wholememory_embedding_cache_policy_t rwp = NULL;
      WM(wholememory_create_embedding_cache_policy(&rwp, comm, WHOLEMEMORY_MT_DISTRIBUTED, WHOLEMEMORY_ML_DEVICE,
                                                   WHOLEMEMORY_AT_READWRITE, 1.0f));
      CHECK(wholememory_communicator_support_type_location(comm, WHOLEMEMORY_MT_DISTRIBUTED, WHOLEMEMORY_ML_HOST) == WHOLEMEMORY_SUCCESS);
      wholememory_embedding_t hemb = NULL;
      WM(wholememory_create_embedding(&hemb, &ed, comm, WHOLEMEMORY_MT_DISTRIBUTED, WHOLEMEMORY_ML_HOST, rwp, NULL, -1, 0));
      wholememory_tensor_t t_h = wholememory_embedding_get_embedding_tensor(hemb);
      CHECK(wholememory_get_memory_location(wholememory_tensor_get_memory_handle(t_h)) == WHOLEMEMORY_ML_HOST);
      void* hp = NULL; size_t hbytes = 0, hoff = 0;
      WM(wholememory_get_local_memory(&hp, &hbytes, &hoff, wholememory_tensor_get_memory_handle(t_h)));
      const int64_t hstride = wholememory_tensor_get_tensor_description(t_h)->strides[0];
      CHECK(hp != NULL && hbytes == sizeof(float) * (size_t)(rows * hstride));
      float* host_table = (float*)hp;   /* a HOST pointer: written and read by the CPU below */
      for (int64_t r = 0; r < rows; r++)
        for (int64_t j = 0; j < dim; j++) host_table[r * hstride + j] = (float)r;
      WM(wholememory_embedding_set_optimizer(hemb, opt));
      float* through2 = (float*)malloc(sizeof(float) * rows * dim);
      void* d_through2 = to_device(through2, sizeof(float) * rows * dim);
      wholememory_tensor_t t_through2 = wrap2d(d_through2, rows, dim, dim, WHOLEMEMORY_DT_FLOAT);
      WM(wholememory_embedding_gather(hemb, t_all, t_through2, true, env, (int64_t)(intptr_t)stream));   /* rows enter the cache */
      WM(wholememory_embedding_gather_gradient_apply(hemb, t_pidx, t_ones, true, 0.5f, env, (int64_t)(intptr_t)stream));
      HIP(hipStreamSynchronize(stream));
      int64_t stale = 0;
      for (int64_t r = 0; r < rows; r++) stale += hits[r] > 0 && host_table[r * hstride] == (float)r;
      CHECK(stale > 0);   /* resident rows were updated in their lines only */
      WM(wholememory_embedding_gather(hemb, t_all, t_through2, true, env, (int64_t)(intptr_t)stream));
      HIP(hipStreamSynchronize(stream));
      HIP(hipMemcpy(through2, d_through2, sizeof(float) * rows * dim, hipMemcpyDeviceToHost));
      for (int64_t r = 0; r < rows; r++)
        for (int64_t j = 0; j < dim; j++) CHECK(through2[r * dim + j] == (float)r - 0.5f * (float)hits[r]);
      WM(wholememory_embedding_writeback_cache(hemb, (int64_t)(intptr_t)stream));
      for (int64_t r = 0; r < rows; r++)
        for (int64_t j = 0; j < dim; j++) CHECK(host_table[r * hstride + j] == (float)r - 0.5f * (float)hits[r]);
      WM(wholememory_embedding_drop_all_cache(hemb, (int64_t)(intptr_t)stream));
      WM(wholememory_destroy_tensor(t_through2));
      WM(wholememory_destroy_embedding(hemb));
      WM(wholememory_destroy_embedding_cache_policy(rwp));
      free(through2);
      printf("host embedding: %lld rows in pinned host memory behind a READWRITE device cache, %lld stale until the write-back, "
             "closed form exact after it\n", (long long)rows, (long long)stale);
    }
    wholememory_tensor_t tmp[] = {t_init, t_all, t_pidx, t_ones};
    for (size_t i = 0; i < 4; i++) WM(wholememory_destroy_tensor(tmp[i]));
    WM(wholememory_destroy_embedding(emb));
    wholememory_destroy_embedding_optimizer(opt);
    WM(wholememory_destroy_communicator(comm));
    printf("embedding: %lld pairs into %lld x %lld fp32 rows (SGD), closed form exact\n", (long long)pairs, (long long)rows,
           (long long)dim);
  }

  /* ---- error contract: wrong dtype -> return code + stderr line, no abort ------------------------------- */
  wholememory_tensor_t t_bad = wrap1d(d_row, V + 1, WHOLEMEMORY_DT_INT);
  wgamd_default_memory_context_t* c_tmp = wgamd_create_default_memory_context();
  CHECK(wholegraph_csr_unweighted_sample_without_replacement(t_bad, t_col, t_seeds, M, t_off, c_tmp, NULL, NULL, rs, env, stream) !=
        WHOLEMEMORY_SUCCESS);

  wholememory_tensor_t all[] = {t_row, t_col, t_seeds, t_off, t_nbr, t_map, t_table, t_idx, t_out, t_bad};
  for (size_t i = 0; i < sizeof(all) / sizeof(all[0]); i++) WM(wholememory_destroy_tensor(all[i]));
  CHECK(get_wholememory_tensor_count() == 0);
  wgamd_destroy_default_memory_context(c_dst);
  wgamd_destroy_default_memory_context(c_lid);
  wgamd_destroy_default_memory_context(c_gid);
  wgamd_destroy_default_memory_context(c_uniq);
  wgamd_destroy_default_memory_context(c_tmp);
  printf("C_ABI_PARITY_OK\n");
  return 0;
}
