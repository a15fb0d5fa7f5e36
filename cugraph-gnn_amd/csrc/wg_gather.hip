// Row gather / scatter of a feature (embedding) table by index — the feature fetch of a mini-batch.
//
// Replaces wholememory_gather / wholememory_scatter
// (/root/reference/cpp/include/wholememory/wholememory_op.h:25-47; reference kernels
// cpp/src/wholememory_ops/functions/gather_scatter_func.cuh:242-365,508-650) for tables that wrap
// a device pointer.  DISTRIBUTED tables (wholememory_malloc / wholememory_create_tensor with a communicator)
// go through the RCCL all-to-all pipeline of wg_comm.hip built on top of these local kernels; the Python host
// layer has the same pipeline over torch.distributed (cugraph-gnn_amd/wholegraph_amd/dist.py).
//
// gfx950 design: the op is pure HBM traffic (random 100 B – 1 KiB row reads, streaming writes).
//   * same dtype in/out  -> dtype-agnostic byte-row copy with the widest aligned vector
//     (16 B/lane), a power-of-two lane group per row (F=100 fp32: 25 of 32 lanes, two rows per
//     wave64; F=128: 32 lanes; F=256: the whole wave), and 4 rows in flight per group so each
//     wave keeps >= 4 independent 16 B loads outstanding per lane before the first store.
//   * converting gathers -> one templated element kernel (load InT, convert through fp32 for the
//     16-bit types, store OutT), lanes along the row for coalescing.
// A negative index leaves its row untouched (gather_scatter_func.cuh:285).
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <cstdlib>

#include "wg_common.hpp"

namespace wgamd {
namespace {

template <int V>
struct vec_of;
template <>
struct vec_of<16> {
  using type = uint4;
};
template <>
struct vec_of<8> {
  using type = uint2;
};
template <>
struct vec_of<4> {
  using type = uint32_t;
};
template <>
struct vec_of<2> {
  using type = uint16_t;
};
template <>
struct vec_of<1> {
  using type = uint8_t;
};

constexpr int kRowsInFlight = 4;

// MODE 0 (gather) : dst row i       <- src row idx[i]
// MODE 1 (scatter): dst row idx[i]  <- src row i
// MODE 2 (permute): dst row idx2[i] <- src row idx[i]   (the owner-is-me share of a DISTRIBUTED gather / scatter)
template <int V, typename IdxT, int MODE>
__global__ void __launch_bounds__(256) row_copy_kernel(const char* __restrict__ src,
                                                       int64_t src_stride,  // bytes
                                                       const IdxT* __restrict__ idx,
                                                       const IdxT* __restrict__ idx2,
                                                       int64_t n,
                                                       int row_bytes,
                                                       char* __restrict__ dst,
                                                       int64_t dst_stride,  // bytes
                                                       int log2_lanes)
{
  using vec_t         = typename vec_of<V>::type;
  const int lanes     = 1 << log2_lanes;
  const int64_t tid   = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t group = tid >> log2_lanes;
  const int sub       = (int)(tid & (lanes - 1));
  const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> log2_lanes;
  const int step      = lanes * V;

  const int iters = (row_bytes + step - 1) / step;  // same trip count for every lane (row_bytes >= V)
  for (int64_t row0 = group * kRowsInFlight; row0 < n; row0 += ngroups * kRowsInFlight) {
    int64_t r[kRowsInFlight], r2[kRowsInFlight];
    bool all_ok = true;
#pragma unroll
    for (int k = 0; k < kRowsInFlight; k++) {
      const int64_t ri = row0 + k < n ? row0 + k : n - 1;  // unconditional index load
      r[k]             = (int64_t)idx[ri];
      r2[k]            = MODE == 2 ? (int64_t)idx2[ri] : 0;
      if (row0 + k >= n || r2[k] < 0) r[k] = -1;
      all_ok = all_ok && r[k] >= 0;
    }
    if (__all(all_ok)) {
      // Fast path (every row of every lane group of the wave is live; wave-uniform branch): NOTHING below is under a
      // per-lane branch — a load or store behind one makes hipcc wait vmcnt(0) at every join, which put the
      // kRowsInFlight fetches in series.  Lanes past the end of the row re-copy its last V bytes (same value twice).
      for (int it = 0; it < iters; it++) {
        int off = sub * V + it * step;
        off     = off + V <= row_bytes ? off : row_bytes - V;
        vec_t v[kRowsInFlight];
#pragma unroll
        for (int k = 0; k < kRowsInFlight; k++) {
          const char* p = MODE == 1 ? src + (row0 + k) * src_stride + off : src + r[k] * src_stride + off;
          v[k]          = *reinterpret_cast<const vec_t*>(p);
        }
#pragma unroll
        for (int k = 0; k < kRowsInFlight; k++) {
          char* q = dst + (MODE == 0 ? row0 + k : MODE == 1 ? r[k] : r2[k]) * dst_stride + off;
          // gathered rows are written once and read by a later kernel: a streaming store keeps them from pushing table rows
          // (re-read by the next mini-batches of the call group) out of the caches — gather 1.47 -> 1.44 ms, products group
          if constexpr (MODE == 0 && V == 16) {
            using nt_t = __attribute__((ext_vector_type(4))) unsigned int;
            __builtin_nontemporal_store(__builtin_bit_cast(nt_t, v[k]), reinterpret_cast<nt_t*>(q));
          } else {
            *reinterpret_cast<vec_t*>(q) = v[k];
          }
        }
      }
      continue;
    }
    for (int off = sub * V; off < row_bytes; off += step) {
      vec_t v[kRowsInFlight];
#pragma unroll
      for (int k = 0; k < kRowsInFlight; k++) {
        if (r[k] >= 0) {
          const char* p = MODE == 1 ? src + (row0 + k) * src_stride + off : src + r[k] * src_stride + off;
          v[k]          = *reinterpret_cast<const vec_t*>(p);
        }
      }
#pragma unroll
      for (int k = 0; k < kRowsInFlight; k++) {
        if (r[k] >= 0) {
          char* q = dst + (MODE == 0 ? row0 + k : MODE == 1 ? r[k] : r2[k]) * dst_stride + off;
          *reinterpret_cast<vec_t*>(q) = v[k];
        }
      }
    }
  }
}

template <typename T>
__device__ __forceinline__ float to_f32(T v)
{
  return (float)v;
}
template <>
__device__ __forceinline__ float to_f32<__half>(__half v)
{
  return __half2float(v);
}
template <>
__device__ __forceinline__ float to_f32<__hip_bfloat16>(__hip_bfloat16 v)
{
  return __bfloat162float(v);
}
template <typename T>
__device__ __forceinline__ T from_f32(float v)
{
  return (T)v;
}
template <>
__device__ __forceinline__ __half from_f32<__half>(float v)
{
  return __float2half(v);
}
template <>
__device__ __forceinline__ __hip_bfloat16 from_f32<__hip_bfloat16>(float v)
{
  return __float2bfloat16(v);
}

template <typename InT, typename OutT>
__device__ __forceinline__ OutT convert(InT v)
{
  if constexpr (std::is_same<InT, OutT>::value) {
    return v;
  } else if constexpr (std::is_integral<InT>::value && std::is_integral<OutT>::value) {
    return (OutT)v;
  } else if constexpr (std::is_same<InT, double>::value && std::is_same<OutT, float>::value) {
    return (float)v;
  } else if constexpr (std::is_same<InT, float>::value && std::is_same<OutT, double>::value) {
    return (double)v;
  } else if constexpr (std::is_same<OutT, double>::value) {
    return (double)to_f32<InT>(v);
  } else if constexpr (std::is_same<InT, double>::value) {
    return from_f32<OutT>((float)v);
  } else {
    return from_f32<OutT>(to_f32<InT>(v));
  }
}

template <typename T, int N>
struct alignas(sizeof(T) * N) packed_t {
  T v[N];
};

// VEC elements per lane and step (4 when rows, strides and pointers allow 4-element accesses: 8-B loads / 16-B stores for
// the fp16 -> fp32 fetch of a half-precision table; the one-element-per-lane form ran at 0.30 of the HBM peak)
template <typename InT, typename OutT, typename IdxT, bool SCATTER, int VEC>
__global__ void __launch_bounds__(256) row_convert_kernel(const InT* __restrict__ src,
                                                          int64_t src_stride,  // elements
                                                          const IdxT* __restrict__ idx,
                                                          int64_t n,
                                                          int row_elems,
                                                          OutT* __restrict__ dst,
                                                          int64_t dst_stride,
                                                          int log2_lanes)
{
  const int lanes       = 1 << log2_lanes;
  const int64_t tid     = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t group   = tid >> log2_lanes;
  const int sub         = (int)(tid & (lanes - 1));
  const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> log2_lanes;
  for (int64_t row = group; row < n; row += ngroups) {
    int64_t r = (int64_t)idx[row];
    if (r < 0) continue;
    const InT* p = SCATTER ? src + row * src_stride : src + r * src_stride;
    OutT* q      = SCATTER ? dst + r * dst_stride : dst + row * dst_stride;
    if constexpr (VEC == 1) {
      for (int e = sub; e < row_elems; e += lanes) q[e] = convert<InT, OutT>(p[e]);
    } else {
      for (int e = sub * VEC; e < row_elems; e += lanes * VEC) {
        const packed_t<InT, VEC> a = *reinterpret_cast<const packed_t<InT, VEC>*>(p + e);
        packed_t<OutT, VEC> b;
#pragma unroll
        for (int k = 0; k < VEC; k++) b.v[k] = convert<InT, OutT>(a.v[k]);
        *reinterpret_cast<packed_t<OutT, VEC>*>(q + e) = b;
      }
    }
  }
}

inline int log2_ceil_lanes(int64_t units)
{
  int l = 0;
  while ((1 << l) < units && l < 6) l++;
  return l;
}

inline int grid_for(int64_t n_rows, int log2_lanes, int rows_per_group)
{
  int64_t groups_per_block = 256 >> log2_lanes;
  int64_t blocks           = (n_rows + groups_per_block * rows_per_group - 1) / (groups_per_block * rows_per_group);
  // memory-bound: grid-stride beyond a cap.  The cap is generous (128 workgroups per CU, i.e. 16 rounds of resident
  // workgroups) on purpose: with 8 — a fully persistent grid — the kernel itself runs just as fast, but its workgroups hold
  // every wave slot until the launch ends and a kernel of another stream (the sampling walk of the next call group) cannot
  // get a foot in; with short-lived workgroups the two interleave: +3 % end to end (3.40 -> 3.52 G edges/s, A/B x 3)
  static const int per_cu = getenv("WGAMD_GATHER_WG_PER_CU") ? atoi(getenv("WGAMD_GATHER_WG_PER_CU")) : 128;
  if (blocks > 256 * per_cu) blocks = 256 * per_cu;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

template <typename IdxT, int MODE>
void launch_copy(const char* src, int64_t src_stride, const IdxT* idx, const IdxT* idx2, int64_t n, int row_bytes, char* dst,
                 int64_t dst_stride, hipStream_t stream)
{
  int V = 16;
  while (V > 1 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)src_stride |
                    (uintptr_t)dst_stride | (uintptr_t)row_bytes) %
                   V) != 0)
    V >>= 1;
  int l2   = log2_ceil_lanes((row_bytes + V - 1) / V);
  int grid = grid_for(n, l2, kRowsInFlight);
#define WG_LAUNCH(VV)                                                                                                  \
  row_copy_kernel<VV, IdxT, MODE><<<grid, 256, 0, stream>>>(src, src_stride, idx, idx2, n, row_bytes, dst, dst_stride, l2)
  switch (V) {
    case 16: WG_LAUNCH(16); break;
    case 8: WG_LAUNCH(8); break;
    case 4: WG_LAUNCH(4); break;
    case 2: WG_LAUNCH(2); break;
    default: WG_LAUNCH(1); break;
  }
#undef WG_LAUNCH
}

template <typename InT, typename OutT, typename IdxT, bool SCATTER>
void launch_convert(const void* src, int64_t src_stride, const IdxT* idx, int64_t n, int row_elems, void* dst,
                    int64_t dst_stride, hipStream_t stream)
{
  const bool v4 = row_elems % 4 == 0 && src_stride % 4 == 0 && dst_stride % 4 == 0 &&
                  reinterpret_cast<uintptr_t>(src) % (4 * sizeof(InT)) == 0 && reinterpret_cast<uintptr_t>(dst) % (4 * sizeof(OutT)) == 0;
  int l2   = log2_ceil_lanes(v4 ? row_elems / 4 : row_elems);
  int grid = grid_for(n, l2, 1);
  if (v4)
    row_convert_kernel<InT, OutT, IdxT, SCATTER, 4><<<grid, 256, 0, stream>>>(
      static_cast<const InT*>(src), src_stride, idx, n, row_elems, static_cast<OutT*>(dst), dst_stride, l2);
  else
    row_convert_kernel<InT, OutT, IdxT, SCATTER, 1><<<grid, 256, 0, stream>>>(
      static_cast<const InT*>(src), src_stride, idx, n, row_elems, static_cast<OutT*>(dst), dst_stride, l2);
}

template <typename InT, typename IdxT, bool SCATTER>
void convert_out(wholememory_dtype_t out_dt, const void* src, int64_t ss, const IdxT* idx, int64_t n, int f, void* dst,
                 int64_t ds, hipStream_t st)
{
  if constexpr (std::is_integral<InT>::value) {
    switch (out_dt) {
      case WHOLEMEMORY_DT_INT8: return launch_convert<InT, int8_t, IdxT, SCATTER>(src, ss, idx, n, f, dst, ds, st);
      case WHOLEMEMORY_DT_INT16: return launch_convert<InT, int16_t, IdxT, SCATTER>(src, ss, idx, n, f, dst, ds, st);
      case WHOLEMEMORY_DT_INT: return launch_convert<InT, int32_t, IdxT, SCATTER>(src, ss, idx, n, f, dst, ds, st);
      case WHOLEMEMORY_DT_INT64: return launch_convert<InT, int64_t, IdxT, SCATTER>(src, ss, idx, n, f, dst, ds, st);
      default: break;
    }
  } else {
    switch (out_dt) {
      case WHOLEMEMORY_DT_HALF: return launch_convert<InT, __half, IdxT, SCATTER>(src, ss, idx, n, f, dst, ds, st);
      case WHOLEMEMORY_DT_BF16: return launch_convert<InT, __hip_bfloat16, IdxT, SCATTER>(src, ss, idx, n, f, dst, ds, st);
      case WHOLEMEMORY_DT_FLOAT: return launch_convert<InT, float, IdxT, SCATTER>(src, ss, idx, n, f, dst, ds, st);
      case WHOLEMEMORY_DT_DOUBLE: return launch_convert<InT, double, IdxT, SCATTER>(src, ss, idx, n, f, dst, ds, st);
      default: break;
    }
  }
  throw logic_error("unsupported output dtype");
}

template <typename IdxT, bool SCATTER>
void convert_in(wholememory_dtype_t in_dt, wholememory_dtype_t out_dt, const void* src, int64_t ss, const IdxT* idx,
                int64_t n, int f, void* dst, int64_t ds, hipStream_t st)
{
  switch (in_dt) {
    case WHOLEMEMORY_DT_INT8: return convert_out<int8_t, IdxT, SCATTER>(out_dt, src, ss, idx, n, f, dst, ds, st);
    case WHOLEMEMORY_DT_INT16: return convert_out<int16_t, IdxT, SCATTER>(out_dt, src, ss, idx, n, f, dst, ds, st);
    case WHOLEMEMORY_DT_INT: return convert_out<int32_t, IdxT, SCATTER>(out_dt, src, ss, idx, n, f, dst, ds, st);
    case WHOLEMEMORY_DT_INT64: return convert_out<int64_t, IdxT, SCATTER>(out_dt, src, ss, idx, n, f, dst, ds, st);
    case WHOLEMEMORY_DT_HALF: return convert_out<__half, IdxT, SCATTER>(out_dt, src, ss, idx, n, f, dst, ds, st);
    case WHOLEMEMORY_DT_BF16: return convert_out<__hip_bfloat16, IdxT, SCATTER>(out_dt, src, ss, idx, n, f, dst, ds, st);
    case WHOLEMEMORY_DT_FLOAT: return convert_out<float, IdxT, SCATTER>(out_dt, src, ss, idx, n, f, dst, ds, st);
    case WHOLEMEMORY_DT_DOUBLE: return convert_out<double, IdxT, SCATTER>(out_dt, src, ss, idx, n, f, dst, ds, st);
    default: throw logic_error("unsupported input dtype");
  }
}

template <bool SCATTER>
void rows_launch(const char* src, wholememory_matrix_description_t sm, const void* idx, wholememory_dtype_t idx_dtype,
                 int64_t n, char* dst, wholememory_matrix_description_t dm, hipStream_t stream)
{
  if (n == 0) return;
  const size_t ses = dtype_size(sm.dtype), des = dtype_size(dm.dtype);
  const int F      = (int)sm.sizes[1];
  if (sm.dtype == dm.dtype) {
    if (idx_dtype == WHOLEMEMORY_DT_INT)
      launch_copy<int32_t, SCATTER ? 1 : 0>(src, sm.stride * (int64_t)ses, static_cast<const int32_t*>(idx), nullptr, n,
                                            F * (int)ses, dst, dm.stride * (int64_t)des, stream);
    else
      launch_copy<int64_t, SCATTER ? 1 : 0>(src, sm.stride * (int64_t)ses, static_cast<const int64_t*>(idx), nullptr, n,
                                            F * (int)ses, dst, dm.stride * (int64_t)des, stream);
  } else {
    if (idx_dtype == WHOLEMEMORY_DT_INT)
      convert_in<int32_t, SCATTER>(sm.dtype, dm.dtype, src, sm.stride, static_cast<const int32_t*>(idx), n, F, dst,
                                   dm.stride, stream);
    else
      convert_in<int64_t, SCATTER>(sm.dtype, dm.dtype, src, sm.stride, static_cast<const int64_t*>(idx), n, F, dst,
                                   dm.stride, stream);
  }
  WG_HIP_CHECK(hipGetLastError());
}

// src_t rows are read, dst_t rows are written; gather: src is the (possibly DISTRIBUTED) table
template <bool SCATTER>
void rows_op(const char* op, wholememory_tensor_t src_t, wholememory_tensor_t idx_t, wholememory_tensor_t dst_t,
             wholememory_env_func_t* env, hipStream_t stream)
{
  WG_REQUIRE_INPUT(src_t && idx_t && dst_t, "null tensor");
  wholememory_tensor_t table = SCATTER ? dst_t : src_t;
  wholememory_tensor_description_t sd = src_t->desc, dd = dst_t->desc;
  WG_REQUIRE_INPUT(sd.dim == 1 || sd.dim == 2, "table / input should be 1D or 2D tensor");
  WG_REQUIRE_INPUT(dd.dim == sd.dim, "output tensor should be same dim as input tensor");
  WG_REQUIRE_INPUT(idx_t->desc.dim == 1, "indices tensor should be 1D tensor");
  if (sd.dim == 1) {
    WG_EXPECTS(wholememory_unsqueeze_tensor(&sd, 1) && wholememory_unsqueeze_tensor(&dd, 1), "unsqueeze failed");
  }
  wholememory_matrix_description_t sm, dm;
  wholememory_array_description_t id;
  WG_REQUIRE_INPUT(wholememory_convert_tensor_desc_to_matrix(&sm, &sd), "cannot view input as a matrix");
  WG_REQUIRE_INPUT(wholememory_convert_tensor_desc_to_matrix(&dm, &dd), "cannot view output as a matrix");
  WG_REQUIRE_INPUT(wholememory_convert_tensor_desc_to_array(&id, &idx_t->desc), "indices must be contiguous 1-D");
  WG_REQUIRE_INPUT(id.dtype == WHOLEMEMORY_DT_INT || id.dtype == WHOLEMEMORY_DT_INT64, "indices must be INT|INT64");
  WG_REQUIRE_INPUT(sm.sizes[1] == dm.sizes[1], "embedding dims differ: %ld vs %ld", (long)sm.sizes[1], (long)dm.sizes[1]);
  const int64_t n = id.size;
  WG_REQUIRE_INPUT((SCATTER ? sm.sizes[0] : dm.sizes[0]) >= n, "fewer dense rows than indices");
  bool sf = wholememory_dtype_is_floating_number(sm.dtype), df = wholememory_dtype_is_floating_number(dm.dtype);
  WG_EXPECTS(sf == df, "embedding and output should be same number type, e.g. floating number or integer number.");
  const void* idx = tensor_data(idx_t);
  if (table->handle != nullptr) {
    // DISTRIBUTED table: RCCL all-to-all pipeline (wg_comm.hip); every rank must make this call
    wholememory_tensor_t dense = SCATTER ? src_t : dst_t;
    WG_REQUIRE_INPUT(env != nullptr, "p_env_fns is required for a DISTRIBUTED table");
    WG_REQUIRE_INPUT(n == 0 || (dense->storage_ptr && idx), "null data pointer");
    const size_t des = dtype_size((SCATTER ? sm : dm).dtype);
    char* dense_ptr  = static_cast<char*>(dense->storage_ptr) + (SCATTER ? sm : dm).storage_offset * (int64_t)des;
    distributed_rows_op(SCATTER, table->handle, SCATTER ? dm : sm, idx, id.dtype, n, dense_ptr, SCATTER ? sm : dm, env,
                        stream);
    return;
  }
  if (n == 0) return;
  const size_t ses = dtype_size(sm.dtype), des = dtype_size(dm.dtype);
  const char* src  = static_cast<const char*>(src_t->storage_ptr) + sm.storage_offset * (int64_t)ses;
  char* dst        = static_cast<char*>(dst_t->storage_ptr) + dm.storage_offset * (int64_t)des;
  WG_REQUIRE_INPUT(src_t->storage_ptr && dst_t->storage_ptr && idx, "null data pointer");
  rows_launch<SCATTER>(src, sm, idx, id.dtype, n, dst, dm, stream);
}

}  // namespace

void local_rows_gather(const char* table, wholememory_matrix_description_t tm, const void* idx,
                       wholememory_dtype_t idx_dtype, int64_t n, char* out, wholememory_matrix_description_t om,
                       hipStream_t stream)
{
  rows_launch<false>(table, tm, idx, idx_dtype, n, out, om, stream);
}

void local_rows_scatter(const char* in, wholememory_matrix_description_t im, const void* idx,
                        wholememory_dtype_t idx_dtype, int64_t n, char* table, wholememory_matrix_description_t tm,
                        hipStream_t stream)
{
  rows_launch<true>(in, im, idx, idx_dtype, n, table, tm, stream);
}

void local_rows_permute(const char* src, wholememory_matrix_description_t sm, const int64_t* src_idx, const int64_t* dst_idx,
                        int64_t n, char* dst, wholememory_matrix_description_t dm, hipStream_t stream)
{
  WG_EXPECTS(sm.dtype == dm.dtype, "row permute does not convert");
  if (n == 0) return;
  const int64_t es = (int64_t)dtype_size(sm.dtype);
  launch_copy<int64_t, 2>(src, sm.stride * es, src_idx, dst_idx, n, (int)(sm.sizes[1] * es), dst, dm.stride * es, stream);
  WG_HIP_CHECK(hipGetLastError());
}

}  // namespace wgamd

extern "C" {

wholememory_error_code_t wholememory_gather(wholememory_tensor_t wholememory_tensor, wholememory_tensor_t indices_tensor,
                                            wholememory_tensor_t output_tensor, wholememory_env_func_t* p_env_fns,
                                            void* stream, int /*gather_sms*/)
{
  return wgamd::guarded("wholememory_gather", [&] {
    wgamd::rows_op<false>("wholememory_gather", wholememory_tensor, indices_tensor, output_tensor, p_env_fns,
                          static_cast<hipStream_t>(stream));
  });
}

wholememory_error_code_t wholememory_scatter(wholememory_tensor_t input_tensor, wholememory_tensor_t indices_tensor,
                                             wholememory_tensor_t wholememory_tensor, wholememory_env_func_t* p_env_fns,
                                             void* stream, int /*scatter_sms*/)
{
  return wgamd::guarded("wholememory_scatter", [&] {
    wgamd::rows_op<true>("wholememory_scatter", input_tensor, indices_tensor, wholememory_tensor, p_env_fns,
                         static_cast<hipStream_t>(stream));
  });
}

}  // extern "C"
