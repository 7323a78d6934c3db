// gx_util.hip -- version, dtype table, synthetic data, checksums (bench/test support kernels).
#include "gx_common.hpp"

extern "C" {

const char* gx_version(void) { return "cudf_amd 0.1.0 (gfx950)"; }

int gx_dtype_size(int dtype)
{
  switch (dtype) {
    case GX_INT8:
    case GX_UINT8:
    case GX_BOOL8: return 1;
    case GX_INT16:
    case GX_UINT16: return 2;
    case GX_INT32:
    case GX_UINT32:
    case GX_FLOAT32: return 4;
    case GX_INT64:
    case GX_UINT64:
    case GX_FLOAT64: return 8;
    default: return 0;
  }
}

}  // extern "C"

namespace gx {

template <typename T>
__global__ void __launch_bounds__(256) k_fill_random(T* out, int64_t n, uint64_t seed, int64_t lo, uint64_t range,
                                                     int is_float)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t r = splitmix64(seed + (uint64_t)i);
    if (is_float) {
      out[i] = (T)((double)(r >> 11) * (1.0 / 9007199254740992.0));  // uniform [0,1), 53 bits
    } else if (range) {
      out[i] = (T)(lo + (int64_t)(r % range));
    } else {
      out[i] = (T)r;
    }
  }
}

__global__ void __launch_bounds__(256) k_mix64(uint64_t* data, int64_t n)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint64_t x = data[i];
    x          = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x          = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    data[i]    = x ^ (x >> 31);
  }
}

__global__ void __launch_bounds__(256) k_sequence_i32(int32_t* out, int64_t n, int32_t start)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = start + (int32_t)i;
}

// total order used by the sortedness check (same rule as gx_sort's to_sortable, ascending)
template <typename U, int KIND>
__device__ __forceinline__ U sortable(U bits)
{
  constexpr U SIGN = U(1) << (sizeof(U) * 8 - 1);
  if (KIND == 1) return bits ^ SIGN;
  if (KIND == 2) {
    constexpr U EXP = (sizeof(U) == 8) ? U(0x7FF0000000000000ull) : U(0x7F800000u);
    const U mag     = bits & U(~SIGN);
    if (mag > EXP) return U(~U(0));
    if (mag == 0) bits = 0;
    return bits ^ ((bits & SIGN) ? U(~U(0)) : SIGN);
  }
  return bits;
}

template <typename U, int KIND>
__global__ void __launch_bounds__(256) k_checksum(const U* in, int64_t n, int descending, unsigned long long* res)
{
  unsigned long long sum = 0, x = 0, bad = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const U v                  = in[i];
    const unsigned long long h = splitmix64((uint64_t)v);
    sum += h;
    x ^= h;
    if (i + 1 < n) {
      const U a = sortable<U, KIND>(v), b = sortable<U, KIND>(in[i + 1]);
      if (descending ? (a < b) : (b < a)) ++bad;
    }
  }
  sum = wave_reduce(sum, SumOp());
  bad = wave_reduce(bad, SumOp());
#pragma unroll
  for (int d = GX_WAVE / 2; d >= 1; d >>= 1) x ^= shfl_xor(x, d);
  if (lane_id() == 0) {
    atomicAdd(&res[0], sum);
    atomicXor(&res[1], x);
    atomicAdd(&res[2], bad);
  }
}

template <typename U, int KIND>
int checksum_launch(const void* in, int64_t n, int descending, uint64_t* res, hipStream_t s)
{
  GX_HIP_TRY(hipMemsetAsync(res, 0, 3 * sizeof(uint64_t), s));
  if (n == 0) return 0;
  int64_t blocks = div_up(n, 256 * 8);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((k_checksum<U, KIND>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const U*>(in), n,
                     descending, reinterpret_cast<unsigned long long*>(res));
  GX_LAUNCH_CHECK();
  return 0;
}

template <typename T>
int fill_launch(void* out, int64_t n, uint64_t seed, int64_t lo, int64_t hi, int is_float, hipStream_t s)
{
  if (n == 0) return 0;
  int64_t blocks = div_up(n, 256 * 8);
  if (blocks > 4096) blocks = 4096;
  const uint64_t range = (hi > lo) ? (uint64_t)(hi - lo) : 0;
  hipLaunchKernelGGL((k_fill_random<T>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<T*>(out), n, seed, lo,
                     range, is_float);
  GX_LAUNCH_CHECK();
  return 0;
}

}  // namespace gx

extern "C" {

int gx_mix64_inplace(uint64_t* data, int64_t n, gx_stream_t s)
{
  if (n < 0 || (n > 0 && !data)) return GX_EINVAL;
  if (n == 0) return 0;
  int64_t blocks = (n + 256 * 8 - 1) / (256 * 8);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gx::k_mix64, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, data, n);
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_fill_random(int dtype, void* out, int64_t n, uint64_t seed, int64_t lo, int64_t hi, gx_stream_t s)
{
  if (n < 0 || (n > 0 && !out)) return GX_EINVAL;
  switch (dtype) {
    case GX_INT8:
    case GX_UINT8:
    case GX_BOOL8: return gx::fill_launch<uint8_t>(out, n, seed, lo, hi, 0, s);
    case GX_INT16:
    case GX_UINT16: return gx::fill_launch<uint16_t>(out, n, seed, lo, hi, 0, s);
    case GX_INT32:
    case GX_UINT32: return gx::fill_launch<uint32_t>(out, n, seed, lo, hi, 0, s);
    case GX_INT64:
    case GX_UINT64: return gx::fill_launch<uint64_t>(out, n, seed, lo, hi, 0, s);
    case GX_FLOAT32: return gx::fill_launch<float>(out, n, seed, lo, hi, 1, s);
    case GX_FLOAT64: return gx::fill_launch<double>(out, n, seed, lo, hi, 1, s);
    default: return GX_EDTYPE;
  }
}

/* see gx.h */
namespace gx {
__global__ void __launch_bounds__(256) k_copy_bytes(const char* __restrict__ src, char* __restrict__ dst, size_t bytes)
{
  // 16-byte lanes over the aligned body, bytes at the ragged ends (src and dst share their alignment in the callers' use)
  const size_t head = (16 - (reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u;
  const bool same   = ((reinterpret_cast<uintptr_t>(src) ^ reinterpret_cast<uintptr_t>(dst)) & 15u) == 0;
  const size_t stride = (size_t)gridDim.x * 256;
  const size_t tid    = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (!same || bytes < 64) {
    for (size_t i = tid; i < bytes; i += stride) dst[i] = src[i];
    return;
  }
  const size_t h = head < bytes ? head : bytes;
  for (size_t i = tid; i < h; i += stride) dst[i] = src[i];
  const size_t nvec = (bytes - h) / 16;
  const uint4* s4   = reinterpret_cast<const uint4*>(src + h);
  uint4* d4         = reinterpret_cast<uint4*>(dst + h);
  for (size_t i = tid; i < nvec; i += stride) d4[i] = s4[i];
  for (size_t i = h + nvec * 16 + tid; i < bytes; i += stride) dst[i] = src[i];
}
}  // namespace gx
int gx_copy_bytes(const void* src, void* dst, size_t bytes, gx_stream_t s)
{
  if (bytes == 0) return 0;
  if (!src || !dst) return GX_EINVAL;
  size_t blocks = (bytes / 16 + 256 * 4 - 1) / (256 * 4);
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(gx::k_copy_bytes, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const char*>(src), static_cast<char*>(dst), bytes);
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_sequence_i32(int32_t* out, int64_t n, int32_t start, gx_stream_t s)
{
  if (n < 0 || (n > 0 && !out)) return GX_EINVAL;
  if (n == 0) return 0;
  int64_t blocks = gx::div_up(n, 256 * 8);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gx::k_sequence_i32, dim3((unsigned)blocks), dim3(256), 0, s, out, n, start);
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_checksum(int dtype, const void* in, int64_t n, int descending, uint64_t* res_dev, gx_stream_t s)
{
  if (n < 0 || !res_dev || (n > 0 && !in)) return GX_EINVAL;
  switch (dtype) {
    case GX_INT8: return gx::checksum_launch<uint8_t, 1>(in, n, descending, res_dev, s);
    case GX_UINT8:
    case GX_BOOL8: return gx::checksum_launch<uint8_t, 0>(in, n, descending, res_dev, s);
    case GX_INT16: return gx::checksum_launch<uint16_t, 1>(in, n, descending, res_dev, s);
    case GX_UINT16: return gx::checksum_launch<uint16_t, 0>(in, n, descending, res_dev, s);
    case GX_INT32: return gx::checksum_launch<uint32_t, 1>(in, n, descending, res_dev, s);
    case GX_UINT32: return gx::checksum_launch<uint32_t, 0>(in, n, descending, res_dev, s);
    case GX_FLOAT32: return gx::checksum_launch<uint32_t, 2>(in, n, descending, res_dev, s);
    case GX_INT64: return gx::checksum_launch<uint64_t, 1>(in, n, descending, res_dev, s);
    case GX_UINT64: return gx::checksum_launch<uint64_t, 0>(in, n, descending, res_dev, s);
    case GX_FLOAT64: return gx::checksum_launch<uint64_t, 2>(in, n, descending, res_dev, s);
    default: return GX_EDTYPE;
  }
}

}  // extern "C"
