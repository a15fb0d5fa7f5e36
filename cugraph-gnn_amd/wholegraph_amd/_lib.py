"""ctypes view of ``libwholegraph_amd.so`` — the C-ABI boundary declared in ``include/*.h``.

This is the role of the reference's Cython module
(/root/reference/python/pylibwholegraph/pylibwholegraph/binding/wholememory_binding.pyx:1-262):
declare the C structs/enums, load the shared library, turn return codes into exceptions.
There is NO fallback: if the HIP library is missing every op raises ``WholeGraphLibraryError``.
"""
import ctypes
import os
from ctypes import (CFUNCTYPE, POINTER, Structure, c_bool, c_float, c_int, c_int64, c_size_t,
                    c_ulonglong, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
# WGAMD_LIBRARY_PATH: another build of the same library (kernel tuning A/B runs); it must exist — there is no fallback
LIB_PATH = os.environ.get("WGAMD_LIBRARY_PATH") or os.path.normpath(os.path.join(_HERE, "..", "lib", "libwholegraph_amd.so"))

WHOLEMEMORY_MAX_TENSOR_DIM = 8

# wholememory_error_code_t (include/wgamd_types.h)
(WHOLEMEMORY_SUCCESS, WHOLEMEMORY_UNKNOW_ERROR, WHOLEMEMORY_NOT_IMPLEMENTED, WHOLEMEMORY_LOGIC_ERROR,
 WHOLEMEMORY_CUDA_ERROR, WHOLEMEMORY_COMMUNICATION_ERROR, WHOLEMEMORY_INVALID_INPUT,
 WHOLEMEMORY_INVALID_VALUE, WHOLEMEMORY_OUT_OF_MEMORY, WHOLEMEMORY_NOT_SUPPORTED,
 WHOLEMEMORY_SYSTEM_ERROR) = range(11)

_ERROR_NAMES = ["SUCCESS", "UNKNOW_ERROR", "NOT_IMPLEMENTED", "LOGIC_ERROR", "CUDA_ERROR",
                "COMMUNICATION_ERROR", "INVALID_INPUT", "INVALID_VALUE", "OUT_OF_MEMORY",
                "NOT_SUPPORTED", "SYSTEM_ERROR"]

# wholememory_dtype_t
(DT_UNKNOWN, DT_FLOAT, DT_HALF, DT_DOUBLE, DT_BF16, DT_INT, DT_INT64, DT_INT16, DT_INT8,
 DT_COUNT) = range(10)

IDS_BYTE_OFFSETS = 64   # WGAMD_IDS_BYTE_OFFSETS (include/wgamd_ext.h): src_ids = int64 byte offsets from x

# wholememory_memory_allocation_type_t
MA_NONE, MA_DEVICE, MA_HOST, MA_PINNED = range(4)


class WholeGraphLibraryError(RuntimeError):
    pass


class WholeMemoryError(RuntimeError):
    """Raised for a non-SUCCESS return code; mirrors check_wholememory_error_code
    (binding .pyx:241-262), which maps codes to Python exceptions."""

    def __init__(self, code, what):
        self.code = code
        super().__init__(f"{what} failed: WHOLEMEMORY_{_ERROR_NAMES[code] if 0 <= code < 11 else code}")


class TensorDescription(Structure):
    _fields_ = [("sizes", c_int64 * WHOLEMEMORY_MAX_TENSOR_DIM),
                ("strides", c_int64 * WHOLEMEMORY_MAX_TENSOR_DIM),
                ("storage_offset", c_int64),
                ("dim", c_int),
                ("dtype", c_int)]


CREATE_CTX_FN = CFUNCTYPE(None, POINTER(c_void_p), c_void_p)
DESTROY_CTX_FN = CFUNCTYPE(None, c_void_p, c_void_p)
MALLOC_FN = CFUNCTYPE(c_void_p, POINTER(TensorDescription), c_int, c_void_p, c_void_p)
FREE_FN = CFUNCTYPE(None, c_void_p, c_void_p)


class TempMemoryFns(Structure):
    _fields_ = [("create_memory_context_fn", CREATE_CTX_FN),
                ("destroy_memory_context_fn", DESTROY_CTX_FN),
                ("malloc_fn", MALLOC_FN),
                ("free_fn", FREE_FN),
                ("global_context", c_void_p)]


class OutputMemoryFns(Structure):
    _fields_ = [("malloc_fn", MALLOC_FN),
                ("free_fn", FREE_FN),
                ("global_context", c_void_p)]


class EnvFns(Structure):
    _fields_ = [("temporary_fns", TempMemoryFns), ("output_fns", OutputMemoryFns)]


class UniqueId(Structure):
    """wholememory_unique_id_t (include/wgamd_comm.h): the 128-byte RCCL bootstrap id, passed BY VALUE."""
    _fields_ = [("internal", ctypes.c_char * 128)]


class CliqueInfo(Structure):
    """clique_info_t (include/wgamd_comm.h; wholememory.h:106-113)."""
    _fields_ = [(n, c_int) for n in ("is_in_clique", "clique_first_rank", "clique_rank", "clique_rank_num", "clique_id",
                                     "clique_num")]


# wholememory_memory_type_t / wholememory_memory_location_t
MT_NONE, MT_CONTINUOUS, MT_CHUNKED, MT_DISTRIBUTED, MT_HIERARCHY = range(5)
ML_NONE, ML_DEVICE, ML_HOST = range(3)

# every symbol include/*.h declares: name -> (restype, argtypes)
_T = c_void_p  # wholememory_tensor_t
SYMBOLS = {
    # wgamd_types.h
    "wholememory_dtype_get_element_size": (c_size_t, [c_int]),
    "wholememory_dtype_is_floating_number": (c_bool, [c_int]),
    "wholememory_dtype_is_integer_number": (c_bool, [c_int]),
    "wholememory_create_array_desc": None,
    "wholememory_create_matrix_desc": None,
    "wholememory_initialize_tensor_desc": (None, [POINTER(TensorDescription)]),
    "wholememory_copy_array_desc_to_matrix": None,
    "wholememory_copy_array_desc_to_tensor": None,
    "wholememory_copy_matrix_desc_to_tensor": None,
    "wholememory_convert_tensor_desc_to_array": None,
    "wholememory_convert_tensor_desc_to_matrix": None,
    "wholememory_get_memory_element_count_from_array": None,
    "wholememory_get_memory_size_from_array": None,
    "wholememory_get_memory_element_count_from_matrix": None,
    "wholememory_get_memory_size_from_matrix": None,
    "wholememory_get_memory_element_count_from_tensor": (c_int64, [POINTER(TensorDescription)]),
    "wholememory_get_memory_size_from_tensor": (c_int64, [POINTER(TensorDescription)]),
    "wholememory_squeeze_tensor": (c_bool, [POINTER(TensorDescription), c_int]),
    "wholememory_unsqueeze_tensor": (c_bool, [POINTER(TensorDescription), c_int]),
    "wholememory_get_default_env_func": (POINTER(EnvFns), []),
    "wgamd_create_default_memory_context": (c_void_p, []),
    "wgamd_destroy_default_memory_context": (None, [c_void_p]),
    # wgamd_tensor.h
    "wholememory_make_tensor_from_pointer": (c_int, [POINTER(_T), c_void_p, POINTER(TensorDescription)]),
    "wholememory_destroy_tensor": (c_int, [_T]),
    "wholememory_tensor_has_handle": (c_bool, [_T]),
    "wholememory_tensor_get_memory_handle": (c_void_p, [_T]),
    "wholememory_tensor_get_tensor_description": (POINTER(TensorDescription), [_T]),
    "wholememory_tensor_get_data_pointer": (c_void_p, [_T]),
    "wholememory_tensor_get_subtensor": (c_int, [_T, POINTER(c_int64), POINTER(c_int64), POINTER(_T)]),
    "wholememory_tensor_get_root": (_T, [_T]),
    "get_wholememory_tensor_count": (c_int64, []),
    # wgamd_ops.h
    "wholegraph_csr_unweighted_sample_without_replacement":
        (c_int, [_T, _T, _T, c_int, _T, c_void_p, c_void_p, c_void_p, c_ulonglong, POINTER(EnvFns), c_void_p]),
    "wgamd_csr_uniform_sample_with_replacement":
        (c_int, [_T, _T, _T, c_int, _T, c_void_p, c_void_p, c_void_p, c_ulonglong, POINTER(EnvFns), c_void_p]),
    "wholegraph_csr_weighted_sample_without_replacement":
        (c_int, [_T, _T, _T, _T, c_int, _T, c_void_p, c_void_p, c_void_p, c_ulonglong, POINTER(EnvFns), c_void_p]),
    "generate_random_positive_int_cpu": (c_int, [c_int64, c_int64, _T]),
    "generate_exponential_distribution_negative_float_cpu": (c_int, [c_int64, c_int64, _T]),
    "graph_append_unique": (c_int, [_T, _T, c_void_p, _T, POINTER(EnvFns), c_void_p]),
    "csr_add_self_loop": (c_int, [_T, _T, _T, _T, c_void_p]),
    "wholememory_gather": (c_int, [_T, _T, _T, POINTER(EnvFns), c_void_p, c_int]),
    "wholememory_scatter": (c_int, [_T, _T, _T, POINTER(EnvFns), c_void_p, c_int]),
    "wholememory_env_test_op": (c_int, [_T, _T, c_void_p, c_void_p, c_void_p, c_int64, POINTER(EnvFns), c_void_p]),
    # wgamd_comm.h
    "wholememory_init": (c_int, [ctypes.c_uint, c_int]),
    "wholememory_finalize": (c_int, []),
    "wholememory_create_unique_id": (c_int, [POINTER(UniqueId)]),
    "wholememory_create_communicator": (c_int, [POINTER(c_void_p), UniqueId, c_int, c_int]),
    "wholememory_destroy_communicator": (c_int, [c_void_p]),
    "wholememory_communicator_support_type_location": (c_int, [c_void_p, c_int, c_int]),
    "wholememory_communicator_get_rank": (c_int, [POINTER(c_int), c_void_p]),
    "wholememory_communicator_get_size": (c_int, [POINTER(c_int), c_void_p]),
    "wholememory_communicator_barrier": (c_int, [c_void_p]),
    "wgamd_communicator_rccl_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "wgamd_get_peer_pointers": (c_int, [POINTER(c_void_p), c_void_p]),
    "wgamd_mapped_row_offsets": (c_int, [_T, c_void_p, c_int, c_int64, c_void_p, POINTER(c_void_p), c_void_p]),
    "wgamd_ipc_export": (c_int, [c_void_p, c_void_p]),
    "wgamd_ipc_open": (c_int, [c_void_p, POINTER(c_void_p)]),
    "wgamd_ipc_close": (c_int, [c_void_p]),
    "wholememory_malloc": (c_int, [POINTER(c_void_p), c_size_t, c_void_p, c_int, c_int, c_size_t, POINTER(c_size_t)]),
    "wholememory_free": (c_int, [c_void_p]),
    "wholememory_get_communicator": (c_int, [POINTER(c_void_p), c_void_p]),
    "wholememory_get_memory_type": (c_int, [c_void_p]),
    "wholememory_get_memory_location": (c_int, [c_void_p]),
    "wholememory_get_total_size": (c_size_t, [c_void_p]),
    "wholememory_get_data_granularity": (c_size_t, [c_void_p]),
    "wholememory_get_local_memory": (c_int, [POINTER(c_void_p), POINTER(c_size_t), POINTER(c_size_t), c_void_p]),
    "wholememory_equal_entry_partition_plan": (c_int, [POINTER(c_size_t), c_size_t, c_int]),
    "wholememory_get_rank_partition_sizes": (c_int, [POINTER(c_size_t), c_void_p]),
    "wholememory_get_rank_partition_offsets": (c_int, [POINTER(c_size_t), c_void_p]),
    "wholememory_create_tensor": (c_int, [POINTER(_T), POINTER(TensorDescription), c_void_p, c_int, c_int,
                                          POINTER(c_size_t)]),
    "wholememory_make_tensor_from_handle": (c_int, [POINTER(_T), c_void_p, POINTER(TensorDescription)]),
    "wholememory_tensor_get_local_entry_count": (c_int, [POINTER(c_size_t), _T]),
    "wholememory_tensor_get_local_entry_start": (c_int, [POINTER(c_size_t), _T]),
    "wholememory_tensor_map_local_tensor": (c_int, [_T, POINTER(_T)]),
    "wholememory_load_from_file": (c_int, [c_void_p, c_size_t, c_size_t, c_size_t, POINTER(ctypes.c_char_p), c_int,
                                           c_int]),
    "wholememory_store_to_file": (c_int, [c_void_p, c_size_t, c_size_t, c_size_t, ctypes.c_char_p]),
    "wholememory_split_communicator": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_int]),
    "wholememory_communicator_get_local_size": (c_int, [POINTER(c_int), c_void_p]),
    "wholememory_communicator_get_clique_info": (c_int, [POINTER(CliqueInfo), c_void_p]),
    "wholememory_communicator_is_bind_to_nvshmem": (c_bool, [c_void_p]),
    "wholememory_communicator_set_distributed_backend": (c_int, [c_void_p, c_int]),
    "wholememory_communicator_get_distributed_backend": (c_int, [c_void_p]),
    "wholememory_is_intranode_communicator": (c_bool, [c_void_p]),
    "wholememory_is_intra_mnnvl_communicator": (c_bool, [c_void_p]),
    "wholememory_is_build_with_nvshmem": (c_bool, []),
    "fork_get_device_count": (c_int, []),
    "wholememory_get_local_communicator": (c_int, [POINTER(c_void_p), c_void_p]),
    "wholememory_get_cross_communicator": (c_int, [POINTER(c_void_p), c_void_p]),
    "wholememory_get_distributed_backend": (c_int, [c_void_p]),
    "wholememory_get_local_size": (c_int, [POINTER(c_size_t), c_void_p]),
    "wholememory_get_local_offset": (c_int, [POINTER(c_size_t), c_void_p]),
    "wholememory_get_rank_memory": (c_int, [POINTER(c_void_p), POINTER(c_size_t), POINTER(c_size_t), c_int, c_void_p]),
    "wholememory_get_global_pointer": (c_int, [POINTER(c_void_p), c_void_p]),
    "wholememory_tensor_get_entry_offsets": (c_int, [POINTER(c_size_t), _T]),
    "wholememory_tensor_get_entry_partition_sizes": (c_int, [POINTER(c_size_t), _T]),
    # wgamd_embedding.h
    "wholememory_create_embedding_optimizer": (c_int, [POINTER(c_void_p), c_int]),
    "wholememory_optimizer_set_parameter": (c_int, [c_void_p, ctypes.c_char_p, c_void_p]),
    "wholememory_destroy_embedding_optimizer": (None, [c_void_p]),
    "wholememory_create_embedding_cache_policy": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_int, c_int,
                                                          ctypes.c_float]),
    "wholememory_destroy_embedding_cache_policy": (c_int, [c_void_p]),
    "wholememory_create_embedding": (c_int, [POINTER(c_void_p), POINTER(TensorDescription), c_void_p, c_int, c_int,
                                             c_void_p, POINTER(c_size_t), c_int, c_int]),
    "wholememory_destroy_embedding": (c_int, [c_void_p]),
    "wholememory_embedding_get_embedding_tensor": (_T, [c_void_p]),
    "wholememory_embedding_set_optimizer": (c_int, [c_void_p, c_void_p]),
    "wholememory_embedding_gather": (c_int, [c_void_p, _T, _T, c_bool, c_void_p, c_int64]),
    "wholememory_embedding_gather_gradient_apply": (c_int, [c_void_p, _T, _T, c_bool, ctypes.c_float, c_void_p,
                                                            c_int64]),
    "wholememory_embedding_get_optimizer_state_names": (POINTER(ctypes.c_char_p), [c_void_p]),
    "wholememory_embedding_get_optimizer_state": (_T, [c_void_p, ctypes.c_char_p]),
    "wholememory_embedding_writeback_cache": (c_int, [c_void_p, c_int64]),
    "wholememory_embedding_drop_all_cache": (c_int, [c_void_p, c_int64]),
    "wgamd_embedding_cache_stats": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    # wgamd_ext.h
    "wgamd_csr_transpose_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "wgamd_csr_transpose_i32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_size_t, c_void_p]),
    "wgamd_coo_to_csr_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "wgamd_coo_to_csr_i64": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                     c_void_p]),
    "wgamd_spmm_csr_f32":(c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int,
                                   c_int, c_void_p, c_int64, c_void_p]),
    "wgamd_sage_aggregate_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int,
                                         c_void_p, c_int64, c_void_p]),
    "wgamd_sage_aggregate_fetch_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int,
                                               c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    "wgamd_gat_csr_bwd_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_float,
                                      c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                      c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wgamd_gat_csr_bwd_f32_v2": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_float,
                                         c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                         c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "wgamd_gat_csr_bwd_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "wgamd_sage_layer_fused_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p, c_int,
                                           c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_int64,
                                           c_void_p]),
    "wgamd_sage_layer_fused_bf16x3": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p, c_int,
                                              c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    "wgamd_sage_layer_fused_bf16x3_train": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p, c_int,
                                                    c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64,
                                                    c_void_p, c_int64, c_void_p]),
    "wgamd_sage_wgrad_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "wgamd_sage_wgrad_bf16x3": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_int64, c_void_p,
                                        c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t,
                                        c_void_p]),
    "wgamd_sage_split_weight_bf16x3": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "wgamd_sage_weight_planes_bytes": (c_size_t, [c_int, c_int]),
    "wgamd_sage_layer_weight_planes": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                             c_int, c_void_p]),
    "wgamd_sage_layer_uses_half_tiles": (c_int, [c_int]),
    "wgamd_sage_layer_bf16x3_supported": (c_int, [c_int, c_int]),
    "wgamd_spmm_csr_bwd_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p,
                                       c_int64, c_void_p]),
    "wgamd_spmm_csr_segmented_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "wgamd_spmm_csr_segmented_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64,
                                             c_void_p, c_size_t, c_void_p]),
    "wgamd_gat_csr_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int,
                                  c_int, c_float, c_void_p, c_void_p, c_int64, c_void_p]),
    "wgamd_frontier_list": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wgamd_call_group_hop_rows": (c_int, [c_void_p] * 5 + [c_int64] + [c_void_p] * 8 + [c_void_p]),
    "wgamd_call_group_hop_rows_batched": (c_int, [c_void_p] * 4 + [c_int64, c_int] + [c_void_p] * 8 + [c_void_p]),
    "wgamd_call_group_layer_cols": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_int] + [c_void_p] * 5),
    "wgamd_call_group_stage_batch": (c_int, [c_int] + [c_void_p] * 5 + [c_int, c_void_p, c_int, c_void_p, c_void_p, c_int]
                                     + [c_void_p] * 10),
    "wgamd_softmax_xent_state_bytes": (c_size_t, [c_int64]),
    "wgamd_softmax_xent_forward_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                               c_void_p, c_void_p]),
    "wgamd_softmax_xent_backward_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_int64, c_void_p]),
    "wgamd_bias_act_rows_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "wgamd_gat_transform_heads_supported": (c_int, [c_int, c_int, c_int]),
    "wgamd_gat_transform_weight_bytes": (c_size_t, [c_int, c_int, c_int]),
    "wgamd_gat_transform_weight_tiles": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "wgamd_gat_transform_heads_bf16x3": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_int64,
                                                 c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "wgamd_gat_layer_fused_supported": (c_int, [c_int, c_int, c_int]),
    "wgamd_gat_layer_fused_bf16x3": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_int,
                                             c_float, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p,
                                             c_int64, c_void_p]),
    "wgamd_gat_aggregate_heads_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int,
                                              c_float, c_void_p, c_void_p, c_int64, c_void_p]),
    "wgamd_rows_terms_bwd_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "wgamd_gather_term_slabs_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "wgamd_gat_aggregate_heads_bwd_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int,
                                                  c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                                  c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "wgamd_gat_aggregate_heads_bwd_gx_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int, c_float,
                                                     c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "wgamd_gat_aggregate_heads_ids_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int,
                                                  c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_int64, c_void_p]),
    "wgamd_gat_layer_fused_ids_bf16x3": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int,
                                                 c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_int64,
                                                 c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "wgamd_gat_csr_rows_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_float,
                                       c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "wgamd_set_weighted_sampling_mode": (None, [c_int, c_int]),
    "wgamd_set_sample_locality_min": (None, [c_int64]),
    "wgamd_sample_hop_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "wgamd_sample_hop_weighted_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int64]),
    "wgamd_sample_hop_nosync": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int,
                                        c_ulonglong, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                        c_void_p, c_void_p, c_size_t, c_void_p]),
    "wgamd_sample_hop_pyg_nosync": (c_int, [c_void_p, c_void_p]),
    "wgamd_call_group_target_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "wgamd_sample_hop_batched_nosync": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                                c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                                c_int64, c_void_p]),
    "wgamd_gather_terms_supported": (c_int, [c_int, c_int]),
    "wgamd_gather_terms_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_void_p, c_int64,
                                       c_void_p, c_int64, c_int, c_void_p]),
    "wgamd_unique_bounded_workspace_bytes": (c_size_t, [c_int64]),
    "wgamd_unique_bounded": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_size_t, c_void_p]),
    "wgamd_unique_bounded_live": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_size_t, c_void_p]),
    "wgamd_sample_hop_batched_nosync_ex": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                                   c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                   c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                                   c_int64, ctypes.c_uint, c_void_p]),
}

_lib = None


def lib():
    """The loaded shared library (loaded once).  Raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WholeGraphLibraryError(
                f"{LIB_PATH} not found: build the HIP library first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C cugraph-gnn_amd/csrc). "
                "There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, sig in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError == a declared symbol is not exported
            if sig is not None:
                fn.restype, fn.argtypes = sig
        _lib = handle
    return _lib


def check(code, what):
    if code != WHOLEMEMORY_SUCCESS:
        raise WholeMemoryError(code, what)
