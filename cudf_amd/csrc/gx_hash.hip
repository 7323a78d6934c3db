// gx_hash.hip -- MurmurHash3_x86_32 column hashing and hash partitioning.
//
// gx_murmur3_32 is bit-compatible with cudf::hashing::detail::MurmurHash3_x86_32<T>
// (cpp/include/cudf/hashing/detail/murmurhash3_x86_32.cuh:22-67: element bytes, seed, floats
// normalised, bool as one byte) so that partitions interoperate with the reference's shuffles.
// gx_hash_partition_map is cudf::hash_partition (cpp/src/partitioning/partitioning.cu:568-660)
// in index form: partition id = hash % P, rows keep their relative order -- implemented as ONE
// stable radix pass over 1/2/4-byte partition ids with an iota payload (gx_sort_pairs).
#include "gx_common.hpp"

#include <type_traits>

namespace gx {

template <typename T>
struct Hasher;

template <>
struct Hasher<uint8_t> {
  static __device__ __forceinline__ uint32_t run(uint8_t v, uint32_t seed) { return murmur3_tail(v, 1u, seed); }
};
template <>
struct Hasher<uint16_t> {
  static __device__ __forceinline__ uint32_t run(uint16_t v, uint32_t seed) { return murmur3_tail(v, 2u, seed); }
};
template <>
struct Hasher<uint32_t> {
  static __device__ __forceinline__ uint32_t run(uint32_t v, uint32_t seed) { return murmur3_u32(v, seed); }
};
template <>
struct Hasher<uint64_t> {
  static __device__ __forceinline__ uint32_t run(uint64_t v, uint32_t seed) { return murmur3_u64(v, seed); }
};

// MODE 0: integer bytes as is; 1: bool (non-zero -> 1); 2: float normalisation (NaN -> quiet NaN,
// +-0 -> +0: cudf/hashing/detail/hash_functions.cuh normalize_nans_and_zeros)
template <typename U, int MODE>
__global__ void __launch_bounds__(256) k_murmur3(const U* __restrict__ in, const uint32_t* __restrict__ valid,
                                                 int64_t n, uint32_t seed, int combine, uint32_t* __restrict__ out)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint32_t h;
    if (valid && !bit_is_set(valid, i)) {
      h = 0xFFFFFFFFu;  // null element (primitive_row_operators.cuh:232-236)
    } else {
      U v = in[i];
      if (MODE == 1) v = v ? U(1) : U(0);
      if (MODE == 2) {
        constexpr U SIGN = U(1) << (sizeof(U) * 8 - 1);
        constexpr U EXP  = (sizeof(U) == 8) ? U(0x7FF0000000000000ull) : U(0x7F800000u);
        constexpr U QNAN = (sizeof(U) == 8) ? U(0x7FF8000000000000ull) : U(0x7FC00000u);
        const U mag      = v & U(~SIGN);
        if (mag > EXP) v = QNAN;
        else if (mag == 0) v = 0;
      }
      h = Hasher<U>::run(v, seed);
    }
    out[i] = combine ? hash_combine32(out[i], h) : h;
  }
}

template <typename U, int MODE>
int murmur_launch(const void* in, const uint32_t* valid, int64_t n, uint32_t seed, int combine, uint32_t* out,
                  hipStream_t s)
{
  int64_t blocks = div_up(n, 256 * 8);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((k_murmur3<U, MODE>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const U*>(in), valid,
                     n, seed, combine, out);
  GX_LAUNCH_CHECK();
  return 0;
}

// IdentityHash<T> of the reference (cpp/src/partitioning/partitioning.cu:852-872): the element cast to uint32 --
// static_cast<uint32_t>(key) for every arithmetic T.  Integers: sign- or zero-extended, then the low 32 bits.  bool: 0 / 1.
// Floating point: truncated toward zero; what C++ leaves undefined (NaN, values outside [0, 2^32)) is given the value the device's
// conversion instruction gives it -- and the reference's: cvt.rzi.u32.f{32,64} and v_cvt_u32_f{32,64} both saturate, NaN -> 0 --
// spelled out here so that the oracle can restate it.  Null element -> UINT32_MAX, column fold as in k_murmur3.
template <typename T>
__device__ __forceinline__ uint32_t identity_u32(T v)
{
  if constexpr (std::is_floating_point<T>::value) {
    if (!(v > T(-1))) return 0u;  // NaN, and everything that truncates below zero
    if (v >= T(4294967296.0)) return 0xFFFFFFFFu;
    return (uint32_t)v;
  } else if constexpr (std::is_same<T, bool>::value) {
    return v ? 1u : 0u;
  } else {
    return (uint32_t)v;
  }
}
template <typename T, bool IS_BOOL>
__global__ void __launch_bounds__(256) k_identity_hash(const T* __restrict__ in, const uint32_t* __restrict__ valid, int64_t n, int combine,
                                                       uint32_t* __restrict__ out)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint32_t h;
    if (valid && !bit_is_set(valid, i)) h = 0xFFFFFFFFu;
    else h = IS_BOOL ? (in[i] ? 1u : 0u) : identity_u32<T>(in[i]);
    out[i] = combine ? hash_combine32(out[i], h) : h;
  }
}
template <typename T, bool IS_BOOL = false>
int identity_launch(const void* in, const uint32_t* valid, int64_t n, int combine, uint32_t* out, hipStream_t s)
{
  int64_t blocks = div_up(n, 256 * 8);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((k_identity_hash<T, IS_BOOL>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const T*>(in), valid, n, combine, out);
  GX_LAUNCH_CHECK();
  return 0;
}

template <typename K>
__global__ void __launch_bounds__(256) k_partition_id(const uint32_t* __restrict__ hash, int64_t n, uint32_t nparts,
                                                      K* __restrict__ pid)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool pow2      = (nparts & (nparts - 1)) == 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t h = hash[i];
    pid[i]           = (K)(pow2 ? (h & (nparts - 1)) : (h % nparts));  // partitioning.cu:53-92
  }
}

template <typename K>
__global__ void __launch_bounds__(256) k_partition_offsets(const K* __restrict__ sorted_pid, int64_t n, int nparts,
                                                           int32_t* __restrict__ offsets)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > nparts) return;
  int64_t lo = 0, hi = n;  // first position with pid >= p
  while (lo < hi) {
    const int64_t mid = lo + (hi - lo) / 2;
    if ((int64_t)sorted_pid[mid] < (int64_t)p) lo = mid + 1; else hi = mid;
  }
  offsets[p] = (int32_t)lo;
}

template <typename K>
int partition_impl(int key_dtype, const uint32_t* row_hash, int64_t n, int nparts, int32_t* out_map,
                   int32_t* out_offsets, void* tmp, size_t* tmp_bytes, hipStream_t s)
{
  size_t sort_bytes = 0;
  int rc = gx_sort_pairs(key_dtype, nullptr, nullptr, nullptr, nullptr, n, 0, nullptr, &sort_bytes, s);
  if (rc) return rc;
  Carver c(tmp);
  K* pid         = c.take<K>((size_t)n);
  K* pid_sorted  = c.take<K>((size_t)n);
  char* sort_tmp = c.take<char>(sort_bytes);
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  if (n > 0) {
    int64_t blocks = div_up(n, 256 * 8);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((k_partition_id<K>), dim3((unsigned)blocks), dim3(256), 0, s, row_hash, n, (uint32_t)nparts,
                       pid);
    size_t sb = sort_bytes;
    rc        = gx_sort_pairs(key_dtype, pid, pid_sorted, nullptr, out_map, n, 0, sort_tmp, &sb, s);
    if (rc) return rc;
  }
  hipLaunchKernelGGL((k_partition_offsets<K>), dim3((unsigned)div_up(nparts + 1, 256)), dim3(256), 0, s, pid_sorted,
                     n, nparts, out_offsets);
  GX_LAUNCH_CHECK();
  return 0;
}

}  // namespace gx

extern "C" {

int gx_murmur3_32(int dtype, const void* in, const uint32_t* valid, int64_t n, uint32_t seed, int combine,
                  uint32_t* out, gx_stream_t s)
{
  if (n < 0) return GX_EINVAL;
  if (n == 0) return 0;
  if (!in || !out) return GX_EINVAL;
  switch (dtype) {
    case GX_BOOL8: return gx::murmur_launch<uint8_t, 1>(in, valid, n, seed, combine, out, s);
    case GX_INT8:
    case GX_UINT8: return gx::murmur_launch<uint8_t, 0>(in, valid, n, seed, combine, out, s);
    case GX_INT16:
    case GX_UINT16: return gx::murmur_launch<uint16_t, 0>(in, valid, n, seed, combine, out, s);
    case GX_INT32:
    case GX_UINT32: return gx::murmur_launch<uint32_t, 0>(in, valid, n, seed, combine, out, s);
    case GX_FLOAT32: return gx::murmur_launch<uint32_t, 2>(in, valid, n, seed, combine, out, s);
    case GX_INT64:
    case GX_UINT64: return gx::murmur_launch<uint64_t, 0>(in, valid, n, seed, combine, out, s);
    case GX_FLOAT64: return gx::murmur_launch<uint64_t, 2>(in, valid, n, seed, combine, out, s);
    default: return GX_EDTYPE;
  }
}

int gx_identity_hash_32(int dtype, const void* in, const uint32_t* valid, int64_t n, int combine, uint32_t* out, gx_stream_t s)
{
  if (n < 0) return GX_EINVAL;
  if (n == 0) return 0;
  if (!in || !out) return GX_EINVAL;
  switch (dtype) {
    case GX_BOOL8: return gx::identity_launch<uint8_t, true>(in, valid, n, combine, out, s);
    case GX_INT8: return gx::identity_launch<int8_t>(in, valid, n, combine, out, s);
    case GX_UINT8: return gx::identity_launch<uint8_t>(in, valid, n, combine, out, s);
    case GX_INT16: return gx::identity_launch<int16_t>(in, valid, n, combine, out, s);
    case GX_UINT16: return gx::identity_launch<uint16_t>(in, valid, n, combine, out, s);
    case GX_INT32: return gx::identity_launch<int32_t>(in, valid, n, combine, out, s);
    case GX_UINT32: return gx::identity_launch<uint32_t>(in, valid, n, combine, out, s);
    case GX_INT64: return gx::identity_launch<int64_t>(in, valid, n, combine, out, s);
    case GX_UINT64: return gx::identity_launch<uint64_t>(in, valid, n, combine, out, s);
    case GX_FLOAT32: return gx::identity_launch<float>(in, valid, n, combine, out, s);
    case GX_FLOAT64: return gx::identity_launch<double>(in, valid, n, combine, out, s);
    default: return GX_EDTYPE;
  }
}

int gx_hash_partition_map(const uint32_t* row_hash, int64_t n, int num_partitions, int32_t* out_map,
                          int32_t* out_offsets, void* tmp, size_t* tmp_bytes, gx_stream_t s)
{
  if (n < 0 || num_partitions < 1 || !tmp_bytes) return GX_EINVAL;
  if (tmp && ((n > 0 && (!row_hash || !out_map)) || !out_offsets)) return GX_EINVAL;
  if (num_partitions <= 256)
    return gx::partition_impl<uint8_t>(GX_UINT8, row_hash, n, num_partitions, out_map, out_offsets, tmp, tmp_bytes, s);
  if (num_partitions <= 65536)
    return gx::partition_impl<uint16_t>(GX_UINT16, row_hash, n, num_partitions, out_map, out_offsets, tmp, tmp_bytes, s);
  return gx::partition_impl<uint32_t>(GX_UINT32, row_hash, n, num_partitions, out_map, out_offsets, tmp, tmp_bytes, s);
}

}  // extern "C"
