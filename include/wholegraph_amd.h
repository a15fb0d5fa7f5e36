/*
 * wholegraph_amd.h — umbrella header of libwholegraph_amd.so, the MI355X-native drop-in for the
 * hot-path subset of libwholegraph's C ABI (the headers under /root/reference/cpp/include/wholememory/).
 *   wgamd_types.h   error codes, dtypes, descriptors, allocator callbacks (env_func_ptrs.h, tensor_description.h)
 *   wgamd_tensor.h  wholememory_tensor_t (wholememory_tensor.h)
 *   wgamd_ops.h     sampling / append_unique / self-loop / gather / scatter (wholegraph_op.h, graph_op.h, wholememory_op.h)
 *   wgamd_ext.h     aggregation kernels + no-sync walk (no reference counterpart)
 */
#ifndef WHOLEGRAPH_AMD_H_
#define WHOLEGRAPH_AMD_H_
#include "wgamd_comm.h"
#include "wgamd_embedding.h"
#include "wgamd_ext.h"
#include "wgamd_ops.h"
#include "wgamd_tensor.h"
#include "wgamd_types.h"
#endif
