/*
 * wgamd_tensor.h — `wholememory_tensor_t`, the handle every op takes.
 * Replaces /root/reference/cpp/include/wholememory/wholememory_tensor.h:17-187 for the subset
 * the hot path uses: tensors that wrap a caller-owned device (or host) pointer, plus the
 * node-local DISTRIBUTED layout (range partition over ranks, see wgamd_comm.h).
 */
#ifndef WGAMD_TENSOR_H_
#define WGAMD_TENSOR_H_

#include "wgamd_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wholememory_tensor_* wholememory_tensor_t;
typedef struct wholememory_handle_* wholememory_handle_t;
typedef struct wholememory_comm_* wholememory_comm_t;

/* wholememory_tensor.h:52-66 — wrap caller-owned storage; storage_ptr is the address of element
 * 0 BEFORE storage_offset is applied. The tensor never owns the storage. */
wholememory_error_code_t wholememory_make_tensor_from_pointer(
  wholememory_tensor_t* wholememory_tensor,
  void* storage_ptr,
  wholememory_tensor_description_t* tensor_description);

/* wholememory_tensor.h:44-50 */
wholememory_error_code_t wholememory_destroy_tensor(wholememory_tensor_t wholememory_tensor);

/* wholememory_tensor.h:81-104 */
bool wholememory_tensor_has_handle(wholememory_tensor_t wholememory_tensor);
wholememory_handle_t wholememory_tensor_get_memory_handle(wholememory_tensor_t wholememory_tensor);
wholememory_tensor_description_t* wholememory_tensor_get_tensor_description(
  wholememory_tensor_t wholememory_tensor);

/* wholememory_tensor.h:124-130 — address of element 0 (storage_offset NOT applied), NULL for
 * tensors backed by a DISTRIBUTED handle. */
void* wholememory_tensor_get_data_pointer(wholememory_tensor_t wholememory_tensor);

/* wholememory_tensor.h:163-176 — view [starts, ends) (−1 = whole dim); shares storage. */
wholememory_error_code_t wholememory_tensor_get_subtensor(
  wholememory_tensor_t wholememory_tensor,
  int64_t* starts,
  int64_t* ends,
  wholememory_tensor_t* sub_wholememory_tensor);
wholememory_tensor_t wholememory_tensor_get_root(wholememory_tensor_t wholememory_tensor);

/* wholememory_tensor.h:184-186 — number of live tensor objects (leak checks in tests) */
int64_t get_wholememory_tensor_count(void);

#ifdef __cplusplus
}
#endif
#endif /* WGAMD_TENSOR_H_ */
