// Descriptor arithmetic, wholememory_tensor_t plumbing and the default (hipMalloc) allocator
// callbacks.  Behavioural spec: /root/reference/cpp/src/wholememory/tensor_description.cpp,
// wholememory_tensor.cpp, env_func_ptrs.cpp:17-72.
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "wg_common.hpp"
#include "wgamd_comm.h"

namespace {

// indexed by wholememory_dtype_t
constexpr size_t kDtypeBytes[WHOLEMEMORY_DT_COUNT] = {0, 4, 2, 8, 2, 4, 8, 2, 1};
constexpr bool kDtypeIsFloat[WHOLEMEMORY_DT_COUNT] = {false, true, true, true, true,
                                                      false, false, false, false};

inline bool valid_dtype(wholememory_dtype_t d) { return d > WHOLEMEMORY_DT_UNKNOWN && d < WHOLEMEMORY_DT_COUNT; }

std::atomic<int64_t> g_live_tensors{0};

}  // namespace

extern "C" {

size_t wholememory_dtype_get_element_size(wholememory_dtype_t dtype)
{
  if (dtype < 0 || dtype >= WHOLEMEMORY_DT_COUNT) return (size_t)-1;
  return kDtypeBytes[dtype];
}

bool wholememory_dtype_is_floating_number(wholememory_dtype_t dtype)
{
  return valid_dtype(dtype) && kDtypeIsFloat[dtype];
}

bool wholememory_dtype_is_integer_number(wholememory_dtype_t dtype)
{
  return valid_dtype(dtype) && !kDtypeIsFloat[dtype];
}

wholememory_array_description_t wholememory_create_array_desc(int64_t size, int64_t storage_offset,
                                                              wholememory_dtype_t dtype)
{
  return wholememory_array_description_t{size, storage_offset, dtype};
}

wholememory_matrix_description_t wholememory_create_matrix_desc(int64_t sizes[2], int64_t stride,
                                                                int64_t storage_offset,
                                                                wholememory_dtype_t dtype)
{
  return wholememory_matrix_description_t{{sizes[0], sizes[1]}, stride, storage_offset, dtype};
}

void wholememory_initialize_tensor_desc(wholememory_tensor_description_t* d)
{
  for (int i = 0; i < WHOLEMEMORY_MAX_TENSOR_DIM; i++) d->sizes[i] = d->strides[i] = 1;
  d->storage_offset = 0;
  d->dim            = 0;
  d->dtype          = WHOLEMEMORY_DT_UNKNOWN;
}

void wholememory_copy_array_desc_to_matrix(wholememory_matrix_description_t* m,
                                           wholememory_array_description_t* a)
{
  *m = wholememory_matrix_description_t{{a->size, 1}, 1, a->storage_offset, a->dtype};
}

void wholememory_copy_array_desc_to_tensor(wholememory_tensor_description_t* t,
                                           wholememory_array_description_t* a)
{
  wholememory_initialize_tensor_desc(t);
  t->dim            = 1;
  t->sizes[0]       = a->size;
  t->storage_offset = a->storage_offset;
  t->dtype          = a->dtype;
}

void wholememory_copy_matrix_desc_to_tensor(wholememory_tensor_description_t* t,
                                            wholememory_matrix_description_t* m)
{
  wholememory_initialize_tensor_desc(t);
  t->dim            = 2;
  t->sizes[0]       = m->sizes[0];
  t->sizes[1]       = m->sizes[1];
  t->strides[0]     = m->stride;
  t->storage_offset = m->storage_offset;
  t->dtype          = m->dtype;
}

bool wholememory_convert_tensor_desc_to_array(wholememory_array_description_t* a,
                                              wholememory_tensor_description_t* t)
{
  if (!valid_dtype(t->dtype) || t->dim != 1 || t->strides[0] != 1) return false;
  *a = wholememory_array_description_t{t->sizes[0], t->storage_offset, t->dtype};
  return true;
}

bool wholememory_convert_tensor_desc_to_matrix(wholememory_matrix_description_t* m,
                                               wholememory_tensor_description_t* t)
{
  if (!valid_dtype(t->dtype)) return false;
  if (t->dim == 1) {
    *m = wholememory_matrix_description_t{{t->sizes[0], 1}, 1, t->storage_offset, t->dtype};
    return true;
  }
  if (t->dim != 2 || t->strides[1] != 1) return false;
  *m = wholememory_matrix_description_t{{t->sizes[0], t->sizes[1]}, t->strides[0], t->storage_offset, t->dtype};
  return true;
}

int64_t wholememory_get_memory_element_count_from_array(wholememory_array_description_t* p) { return p->size; }
int64_t wholememory_get_memory_size_from_array(wholememory_array_description_t* p)
{
  return p->size * (int64_t)wholememory_dtype_get_element_size(p->dtype);
}
int64_t wholememory_get_memory_element_count_from_matrix(wholememory_matrix_description_t* p)
{
  return p->sizes[0] * p->stride;
}
int64_t wholememory_get_memory_size_from_matrix(wholememory_matrix_description_t* p)
{
  return p->sizes[0] * p->stride * (int64_t)wholememory_dtype_get_element_size(p->dtype);
}
int64_t wholememory_get_memory_element_count_from_tensor(wholememory_tensor_description_t* p)
{
  if (p->dim == 0) return 1;
  if (p->dim < 0 || p->dim >= WHOLEMEMORY_MAX_TENSOR_DIM) return -1;
  return p->sizes[0] * p->strides[0];
}
int64_t wholememory_get_memory_size_from_tensor(wholememory_tensor_description_t* p)
{
  return wholememory_get_memory_element_count_from_tensor(p) * (int64_t)wholememory_dtype_get_element_size(p->dtype);
}

bool wholememory_squeeze_tensor(wholememory_tensor_description_t* t, int dim)
{
  if (!t || dim < 0 || dim >= t->dim || t->sizes[dim] != 1) return false;
  const int last = t->dim - 1;
  if (dim != last && t->strides[dim] != t->strides[dim + 1]) return false;
  memmove(&t->sizes[dim], &t->sizes[dim + 1], sizeof(int64_t) * (size_t)(last - dim));
  memmove(&t->strides[dim], &t->strides[dim + 1], sizeof(int64_t) * (size_t)(last - dim));
  t->dim = last;
  return true;
}

bool wholememory_unsqueeze_tensor(wholememory_tensor_description_t* t, int dim)
{
  if (!t || dim < 0 || dim > t->dim || t->dim + 1 > WHOLEMEMORY_MAX_TENSOR_DIM) return false;
  // the new unit dim takes the stride of the dim it displaces (innermost: the old last stride)
  int64_t new_stride = (dim < t->dim) ? t->strides[dim] : (t->dim > 0 ? t->strides[t->dim - 1] : 1);
  for (int i = t->dim; i > dim; i--) {
    t->sizes[i]   = t->sizes[i - 1];
    t->strides[i] = t->strides[i - 1];
  }
  t->sizes[dim]   = 1;
  t->strides[dim] = new_stride;
  t->dim++;
  return true;
}

// ---- tensors -------------------------------------------------------------------------------
wholememory_error_code_t wholememory_make_tensor_from_pointer(wholememory_tensor_t* out, void* storage_ptr,
                                                              wholememory_tensor_description_t* desc)
{
  if (out == nullptr || desc == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (desc->dim < 0 || desc->dim > WHOLEMEMORY_MAX_TENSOR_DIM) return WHOLEMEMORY_INVALID_INPUT;
  if (desc->dim > 0 && desc->sizes[desc->dim - 1] > 1 && desc->strides[desc->dim - 1] != 1) {
    fprintf(stderr, "[wholegraph_amd] make_tensor_from_pointer: innermost stride must be 1\n");
    return WHOLEMEMORY_INVALID_VALUE;
  }
  if (desc->storage_offset < 0) return WHOLEMEMORY_INVALID_VALUE;
  auto* t        = new wholememory_tensor_;
  t->storage_ptr = storage_ptr;
  t->desc        = *desc;
  t->root        = nullptr;
  t->handle      = nullptr;
  g_live_tensors++;
  *out = t;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_destroy_tensor(wholememory_tensor_t t)
{
  if (t == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  // (the handle may be gone — released with its communicator — and its address reused by an unrelated live handle: the serial
  //  number tells the two apart, a stale free is a no-op)
  if (t->owns_handle && t->handle != nullptr) (void)wgamd_free_if_serial(t->handle, t->handle_serial);
  delete t;
  g_live_tensors--;
  return WHOLEMEMORY_SUCCESS;
}

bool wholememory_tensor_has_handle(wholememory_tensor_t t) { return t != nullptr && t->handle != nullptr; }
wholememory_handle_t wholememory_tensor_get_memory_handle(wholememory_tensor_t t)
{
  return t ? t->handle : nullptr;
}
wholememory_tensor_description_t* wholememory_tensor_get_tensor_description(wholememory_tensor_t t)
{
  return t ? &t->desc : nullptr;
}
void* wholememory_tensor_get_data_pointer(wholememory_tensor_t t)
{
  if (t == nullptr || t->handle != nullptr) return nullptr;
  return t->storage_ptr;
}

wholememory_error_code_t wholememory_tensor_get_subtensor(wholememory_tensor_t t, int64_t* starts, int64_t* ends,
                                                          wholememory_tensor_t* sub)
{
  if (t == nullptr || starts == nullptr || ends == nullptr || sub == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (t->desc.dim < 1 || t->desc.dim > 2) return WHOLEMEMORY_NOT_IMPLEMENTED;
  wholememory_tensor_description_t d = t->desc;
  int64_t offset                     = d.storage_offset;
  for (int i = 0; i < d.dim; i++) {
    int64_t s = starts[i] == -1 ? 0 : starts[i];
    int64_t e = ends[i] == -1 ? d.sizes[i] : ends[i];
    if (s < 0 || e > d.sizes[i] || s >= e) return WHOLEMEMORY_INVALID_VALUE;
    offset += s * d.strides[i];
    d.sizes[i] = e - s;
  }
  d.storage_offset = offset;
  auto* v          = new wholememory_tensor_;
  v->storage_ptr   = t->storage_ptr;
  v->desc          = d;
  v->root          = t->root ? t->root : t;
  v->handle        = t->handle;
  g_live_tensors++;
  *sub = v;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_tensor_t wholememory_tensor_get_root(wholememory_tensor_t t)
{
  if (t == nullptr) return nullptr;
  return t->root ? t->root : t;
}

int64_t get_wholememory_tensor_count(void) { return g_live_tensors.load(); }

// ---- default allocator callbacks -----------------------------------------------------------
static void default_create_ctx(void** memory_context, void* /*global*/)
{
  auto* c = static_cast<wgamd_default_memory_context_t*>(calloc(1, sizeof(wgamd_default_memory_context_t)));
  wholememory_initialize_tensor_desc(&c->desc);
  *memory_context = c;
}

static void default_release(wgamd_default_memory_context_t* c)
{
  if (c->ptr == nullptr) return;
  switch (c->allocation_type) {
    case WHOLEMEMORY_MA_DEVICE: (void)hipFree(c->ptr); break;
    case WHOLEMEMORY_MA_PINNED: (void)hipHostFree(c->ptr); break;
    case WHOLEMEMORY_MA_HOST: free(c->ptr); break;
    default: break;
  }
  c->ptr             = nullptr;
  c->allocation_type = WHOLEMEMORY_MA_NONE;
}

static void default_destroy_ctx(void* memory_context, void* /*global*/)
{
  auto* c = static_cast<wgamd_default_memory_context_t*>(memory_context);
  default_release(c);
  free(c);
}

static void* default_malloc(wholememory_tensor_description_t* desc, wholememory_memory_allocation_type_t type,
                            void* memory_context, void* /*global*/)
{
  auto* c = static_cast<wgamd_default_memory_context_t*>(memory_context);
  default_release(c);
  c->desc            = *desc;
  c->allocation_type = type;
  size_t bytes       = (size_t)wholememory_get_memory_size_from_tensor(desc);
  if (bytes == 0) return nullptr;
  void* p = nullptr;
  if (type == WHOLEMEMORY_MA_DEVICE) {
    if (hipMalloc(&p, bytes) != hipSuccess) p = nullptr;
  } else if (type == WHOLEMEMORY_MA_PINNED) {
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) p = nullptr;
  } else if (type == WHOLEMEMORY_MA_HOST) {
    p = malloc(bytes);
  }
  c->ptr = p;
  return p;
}

static void default_free(void* memory_context, void* /*global*/)
{
  default_release(static_cast<wgamd_default_memory_context_t*>(memory_context));
}

wholememory_env_func_t* wholememory_get_default_env_func(void)
{
  static wholememory_env_func_t env = {
    {default_create_ctx, default_destroy_ctx, default_malloc, default_free, nullptr},
    {default_malloc, default_free, nullptr},
  };
  return &env;
}

wgamd_default_memory_context_t* wgamd_create_default_memory_context(void)
{
  void* c = nullptr;
  default_create_ctx(&c, nullptr);
  return static_cast<wgamd_default_memory_context_t*>(c);
}

void wgamd_destroy_default_memory_context(wgamd_default_memory_context_t* ctx)
{
  if (ctx) default_destroy_ctx(ctx, nullptr);
}

}  // extern "C"
