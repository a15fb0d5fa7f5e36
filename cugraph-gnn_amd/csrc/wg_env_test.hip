// wholememory_env_test_op — the allocator-callback self test of the boundary (include/wgamd_ops.h).
// Contract: /root/reference/cpp/include/wholememory/wholememory_op.h:49-68, behaviour
// cpp/src/wholememory_ops/wholememory_test_op.cu:53-140 (out[i, j] = (T)(float)i + input[j] into temporary memory, copied to
// the caller's fixed output and to one output per allocation type obtained through output_fns).  A binding calls it once to
// prove that its create/malloc/free/destroy callbacks and its DEVICE / PINNED / HOST output allocations work.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "wg_common.hpp"

namespace wgamd {
namespace {

template <typename T>
__device__ __forceinline__ T row_tag(int64_t i)
{
  return static_cast<T>(static_cast<float>(i));
}
template <>
__device__ __forceinline__ __half row_tag<__half>(int64_t i)
{
  return __float2half(static_cast<float>(i));
}
template <>
__device__ __forceinline__ __hip_bfloat16 row_tag<__hip_bfloat16>(int64_t i)
{
  return __float2bfloat16(static_cast<float>(i));
}

// one thread per element of the dense [entries, dim] scratch (row stride = dim)
template <typename T>
__global__ void __launch_bounds__(256) env_test_fill_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t dim,
                                                            int64_t total)
{
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / dim, j = e - i * dim;
    out[e]          = row_tag<T>(i) + in[j];
  }
}

template <typename T>
void fill(const void* in, void* out, int64_t dim, int64_t entries, hipStream_t stream)
{
  const int64_t total = dim * entries;
  if (total == 0) return;
  const int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 32);
  env_test_fill_kernel<T><<<grid, 256, 0, stream>>>(static_cast<const T*>(in), static_cast<T*>(out), dim, total);
  WG_HIP_CHECK(hipGetLastError());
}

void* output_of_type(wholememory_env_func_t* env, void* ctx, wholememory_tensor_description_t* desc,
                     wholememory_memory_allocation_type_t where)
{
  void* p = env->output_fns.malloc_fn(desc, where, ctx, env->output_fns.global_context);
  if (p == nullptr && wholememory_get_memory_element_count_from_tensor(desc) > 0) throw std::bad_alloc();
  return p;
}

}  // namespace
}  // namespace wgamd

extern "C" wholememory_error_code_t wholememory_env_test_op(wholememory_tensor_t input_tensor,
                                                            wholememory_tensor_t output_fixed_tensor,
                                                            void* output_variable_device_tensor_handle,
                                                            void* output_variable_pinned_tensor_handle,
                                                            void* output_variable_host_tensor_handle,
                                                            int64_t output_variable_entry_count,
                                                            wholememory_env_func_t* p_env_fns, void* stream)
{
  using namespace wgamd;
  return guarded("wholememory_env_test_op", [&] {
    WG_REQUIRE_INPUT(input_tensor && output_fixed_tensor && p_env_fns, "null argument");
    auto* in_d  = wholememory_tensor_get_tensor_description(input_tensor);
    auto* out_d = wholememory_tensor_get_tensor_description(output_fixed_tensor);
    WG_EXPECTS(in_d->dim == 1, "input must be 1-D");
    const int64_t dim = in_d->sizes[0], entries = output_variable_entry_count;
    WG_EXPECTS(out_d->dim == 2 && out_d->sizes[0] == entries && out_d->sizes[1] == dim, "output_fixed_tensor must be [entry_count, dim]");
    WG_EXPECTS(in_d->dtype == out_d->dtype, "dtype mismatch");
    WG_EXPECTS(entries >= 0, "negative entry count");
    auto s = static_cast<hipStream_t>(stream);
    const size_t es = dtype_size(in_d->dtype), bytes = (size_t)(entries * dim) * es;

    temp_buffer scratch(p_env_fns);
    void* tmp      = scratch.alloc(entries * dim, in_d->dtype);
    const void* in = tensor_data(input_tensor);
    switch (in_d->dtype) {
      case WHOLEMEMORY_DT_FLOAT: fill<float>(in, tmp, dim, entries, s); break;
      case WHOLEMEMORY_DT_DOUBLE: fill<double>(in, tmp, dim, entries, s); break;
      case WHOLEMEMORY_DT_HALF: fill<__half>(in, tmp, dim, entries, s); break;
      case WHOLEMEMORY_DT_BF16: fill<__hip_bfloat16>(in, tmp, dim, entries, s); break;
      case WHOLEMEMORY_DT_INT: fill<int32_t>(in, tmp, dim, entries, s); break;
      case WHOLEMEMORY_DT_INT64: fill<int64_t>(in, tmp, dim, entries, s); break;
      case WHOLEMEMORY_DT_INT16: fill<int16_t>(in, tmp, dim, entries, s); break;
      case WHOLEMEMORY_DT_INT8: fill<int8_t>(in, tmp, dim, entries, s); break;
      default: throw logic_error("unsupported dtype");
    }
    wholememory_tensor_description_t dense = *out_d;  // the variable outputs are dense [entries, dim] blocks
    dense.strides[0]     = dim;
    dense.strides[1]     = 1;
    dense.storage_offset = 0;
    void* dev = output_variable_device_tensor_handle
                  ? output_of_type(p_env_fns, output_variable_device_tensor_handle, &dense, WHOLEMEMORY_MA_DEVICE) : nullptr;
    void* pin = output_variable_pinned_tensor_handle
                  ? output_of_type(p_env_fns, output_variable_pinned_tensor_handle, &dense, WHOLEMEMORY_MA_PINNED) : nullptr;
    void* host = output_variable_host_tensor_handle
                   ? output_of_type(p_env_fns, output_variable_host_tensor_handle, &dense, WHOLEMEMORY_MA_HOST) : nullptr;
    if (bytes) {
      // the fixed output may have strided rows
      char* fixed = static_cast<char*>(tensor_data(output_fixed_tensor));
      WG_HIP_CHECK(hipMemcpy2DAsync(fixed, (size_t)out_d->strides[0] * es, tmp, (size_t)dim * es, (size_t)dim * es,
                                    (size_t)entries, hipMemcpyDefault, s));
      if (dev) WG_HIP_CHECK(hipMemcpyAsync(dev, tmp, bytes, hipMemcpyDefault, s));
      if (pin) WG_HIP_CHECK(hipMemcpyAsync(pin, tmp, bytes, hipMemcpyDefault, s));
      if (host) WG_HIP_CHECK(hipMemcpyAsync(host, tmp, bytes, hipMemcpyDefault, s));
    }
    WG_HIP_CHECK(hipStreamSynchronize(s));  // the scratch goes back to the caller's allocator on return
  });
}
