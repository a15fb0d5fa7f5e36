/*
 * wgamd_types.h — plain-C types of the drop-in boundary (MI355X-native WholeGraph hot path).
 *
 * Names, enumerator ORDER and struct layouts are those of the reference so that a binding
 * compiled against the reference headers keeps working when pointed at libwholegraph_amd.so:
 *   error codes / memory types      /root/reference/cpp/include/wholememory/wholememory.h:21-65
 *   dtypes + array/matrix/tensor    /root/reference/cpp/include/wholememory/tensor_description.h:18-88
 *   allocator callbacks (ownership) /root/reference/cpp/include/wholememory/env_func_ptrs.h:22-62
 * No CUDA/HIP/torch type appears in any signature: streams travel as `void*` (a hipStream_t).
 */
#ifndef WGAMD_TYPES_H_
#define WGAMD_TYPES_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define WGAMD_DEFAULT(v) = v
extern "C" {
#else
#define WGAMD_DEFAULT(v)
#endif

/* ---- return codes (wholememory.h:21-33) ------------------------------------------------- */
typedef enum wholememory_error_code_t {
  WHOLEMEMORY_SUCCESS = 0,
  WHOLEMEMORY_UNKNOW_ERROR,
  WHOLEMEMORY_NOT_IMPLEMENTED,
  WHOLEMEMORY_LOGIC_ERROR,
  WHOLEMEMORY_CUDA_ERROR, /* device-runtime (HIP) error; name kept for source compatibility */
  WHOLEMEMORY_COMMUNICATION_ERROR,
  WHOLEMEMORY_INVALID_INPUT,
  WHOLEMEMORY_INVALID_VALUE,
  WHOLEMEMORY_OUT_OF_MEMORY,
  WHOLEMEMORY_NOT_SUPPORTED,
  WHOLEMEMORY_SYSTEM_ERROR
} wholememory_error_code_t;

/* ---- memory type / location (wholememory.h:48-65).  Only the node-local layouts this build
 * implements are accepted at run time: NONE (raw pointer) and DISTRIBUTED (range partition,
 * remote rows fetched by RCCL all-to-all); CONTINUOUS/CHUNKED/HIERARCHY return NOT_SUPPORTED. */
typedef enum wholememory_memory_type_t {
  WHOLEMEMORY_MT_NONE = 0,
  WHOLEMEMORY_MT_CONTINUOUS,
  WHOLEMEMORY_MT_CHUNKED,
  WHOLEMEMORY_MT_DISTRIBUTED,
  WHOLEMEMORY_MT_HIERARCHY
} wholememory_memory_type_t;

typedef enum wholememory_memory_location_t {
  WHOLEMEMORY_ML_NONE = 0,
  WHOLEMEMORY_ML_DEVICE,
  WHOLEMEMORY_ML_HOST
} wholememory_memory_location_t;

/* ---- element types (tensor_description.h:18-29) ----------------------------------------- */
typedef enum wholememory_dtype_t {
  WHOLEMEMORY_DT_UNKNOWN = 0,
  WHOLEMEMORY_DT_FLOAT,
  WHOLEMEMORY_DT_HALF,
  WHOLEMEMORY_DT_DOUBLE,
  WHOLEMEMORY_DT_BF16,
  WHOLEMEMORY_DT_INT,
  WHOLEMEMORY_DT_INT64,
  WHOLEMEMORY_DT_INT16,
  WHOLEMEMORY_DT_INT8,
  WHOLEMEMORY_DT_COUNT
} wholememory_dtype_t;

size_t wholememory_dtype_get_element_size(wholememory_dtype_t dtype);
bool wholememory_dtype_is_floating_number(wholememory_dtype_t dtype);
bool wholememory_dtype_is_integer_number(wholememory_dtype_t dtype);

/* ---- descriptors: sizes / strides / offsets are in ELEMENTS (tensor_description.h:58-88) --- */
typedef struct wholememory_array_description_t {
  int64_t size;
  int64_t storage_offset;
  wholememory_dtype_t dtype;
} wholememory_array_description_t;

typedef struct wholememory_matrix_description_t {
  int64_t sizes[2];
  int64_t stride;
  int64_t storage_offset;
  wholememory_dtype_t dtype;
} wholememory_matrix_description_t;

#define WHOLEMEMORY_MAX_TENSOR_DIM (8)

typedef struct wholememory_tensor_description_t {
  int64_t sizes[WHOLEMEMORY_MAX_TENSOR_DIM];
  int64_t strides[WHOLEMEMORY_MAX_TENSOR_DIM];
  int64_t storage_offset;
  int dim;
  wholememory_dtype_t dtype;
} wholememory_tensor_description_t;

/* descriptor helpers (tensor_description.h:90-231) */
wholememory_array_description_t wholememory_create_array_desc(int64_t size,
                                                              int64_t storage_offset,
                                                              wholememory_dtype_t dtype);
wholememory_matrix_description_t wholememory_create_matrix_desc(int64_t sizes[2],
                                                                int64_t stride,
                                                                int64_t storage_offset,
                                                                wholememory_dtype_t dtype);
void wholememory_initialize_tensor_desc(wholememory_tensor_description_t* p_tensor_description);
void wholememory_copy_array_desc_to_matrix(wholememory_matrix_description_t* p_matrix_description,
                                           wholememory_array_description_t* p_array_description);
void wholememory_copy_array_desc_to_tensor(wholememory_tensor_description_t* p_tensor_description,
                                           wholememory_array_description_t* p_array_description);
void wholememory_copy_matrix_desc_to_tensor(wholememory_tensor_description_t* p_tensor_description,
                                            wholememory_matrix_description_t* p_matrix_description);
bool wholememory_convert_tensor_desc_to_array(wholememory_array_description_t* p_array_description,
                                              wholememory_tensor_description_t* p_tensor_description);
bool wholememory_convert_tensor_desc_to_matrix(
  wholememory_matrix_description_t* p_matrix_description,
  wholememory_tensor_description_t* p_tensor_description);
int64_t wholememory_get_memory_element_count_from_array(wholememory_array_description_t* p);
int64_t wholememory_get_memory_size_from_array(wholememory_array_description_t* p);
int64_t wholememory_get_memory_element_count_from_matrix(wholememory_matrix_description_t* p);
int64_t wholememory_get_memory_size_from_matrix(wholememory_matrix_description_t* p);
int64_t wholememory_get_memory_element_count_from_tensor(wholememory_tensor_description_t* p);
int64_t wholememory_get_memory_size_from_tensor(wholememory_tensor_description_t* p);
bool wholememory_squeeze_tensor(wholememory_tensor_description_t* p_tensor_description, int dim);
bool wholememory_unsqueeze_tensor(wholememory_tensor_description_t* p_tensor_description, int dim);

/* ---- allocator callbacks: THE ownership convention of the ABI (env_func_ptrs.h:22-62) -------
 * Variable-size op outputs are allocated by the callee THROUGH the caller's
 * output_fns.malloc_fn(desc, DEVICE, memory_context, global_context); the caller reads pointer
 * and shape back from its own memory_context.  Scratch goes through temporary_fns and is
 * released before the op returns.  The callee never frees an output. */
typedef enum wholememory_memory_allocation_type_t {
  WHOLEMEMORY_MA_NONE = 0,
  WHOLEMEMORY_MA_DEVICE,
  WHOLEMEMORY_MA_HOST,
  WHOLEMEMORY_MA_PINNED
} wholememory_memory_allocation_type_t;

typedef void (*wholememory_create_memory_context_func_t)(void** memory_context,
                                                         void* global_context);
typedef void (*wholememory_destroy_memory_context_func_t)(void* memory_context,
                                                          void* global_context);
typedef void* (*wholememory_malloc_func_t)(
  wholememory_tensor_description_t* desc,
  wholememory_memory_allocation_type_t memory_allocation_type,
  void* memory_context,
  void* global_context);
typedef void (*wholememory_free_func_t)(void* memory_context, void* global_context);

typedef struct wholememory_temp_memory_func_t {
  wholememory_create_memory_context_func_t create_memory_context_fn;
  wholememory_destroy_memory_context_func_t destroy_memory_context_fn;
  wholememory_malloc_func_t malloc_fn;
  wholememory_free_func_t free_fn;
  void* global_context;
} wholememory_temp_memory_func_t;

typedef struct wholememory_output_memory_func_t {
  wholememory_malloc_func_t malloc_fn;
  wholememory_free_func_t free_fn;
  void* global_context;
} wholememory_output_memory_func_t;

typedef struct wholememory_env_func_t {
  wholememory_temp_memory_func_t temporary_fns;
  wholememory_output_memory_func_t output_fns;
} wholememory_env_func_t;

/* Default hipMalloc-backed callbacks and the memory context they use
 * (/root/reference/cpp/src/wholememory/env_func_ptrs.cpp:17-72,
 *  python binding .pyx `wholememory_get_default_env_func`). */
typedef struct wgamd_default_memory_context_t {
  wholememory_tensor_description_t desc;
  wholememory_memory_allocation_type_t allocation_type;
  void* ptr;
} wgamd_default_memory_context_t;

wholememory_env_func_t* wholememory_get_default_env_func(void);
/* helpers for plain-C callers of ops with variable-size outputs */
wgamd_default_memory_context_t* wgamd_create_default_memory_context(void);
void wgamd_destroy_default_memory_context(wgamd_default_memory_context_t* ctx); /* frees ptr too */

#ifdef __cplusplus
}
#endif
#endif /* WGAMD_TYPES_H_ */
