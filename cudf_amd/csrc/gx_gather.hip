// gx_gather.hip -- gather of fixed-width columns (+ validity) and validity-bitmap utilities.
//
// gather replaces cudf::detail::gather (include/cudf/detail/gather.cuh:108-131 for data,
// :506-577 gather_bitmask kernel).  HBM-bound: reads 4 B of map + one random element, writes one
// coalesced element; the output bitmap is assembled with one wave64 ballot per 64 rows.
#include "gx_common.hpp"

namespace gx {

constexpr int GATHER_BT = 256;

template <typename T, bool HAS_VALID>
__global__ void __launch_bounds__(GATHER_BT) k_gather(const T* __restrict__ src,
                                                      const uint32_t* __restrict__ src_valid, int64_t src_rows,
                                                      const int32_t* __restrict__ map, int64_t n, int nullify_oob,
                                                      T* __restrict__ out, uint32_t* __restrict__ out_valid)
{
  // n rounded up to a multiple of 64 so every wave covers an aligned 64-row span of the bitmap
  const int64_t n64    = (n + 63) & ~int64_t(63);
  const int64_t stride = (int64_t)gridDim.x * GATHER_BT;
  for (int64_t i = (int64_t)blockIdx.x * GATHER_BT + threadIdx.x; i < n64; i += stride) {
    bool ok = false;
    if (i < n) {
      const int64_t m = map[i];
      const bool inb  = !nullify_oob || (m >= 0 && m < src_rows);
      T v             = T(0);
      if (inb) {
        v  = src[m];
        ok = !HAS_VALID || src_valid == nullptr || bit_is_set(src_valid, m);
      }
      out[i] = v;
    }
    if (HAS_VALID) {
      const uint64_t b = ballot(ok);
      if (lane_id() == 0) {
        const int64_t w = i >> 5;  // i is a multiple of 64 here
        out_valid[w]     = (uint32_t)b;
        if (w + 1 < ((n + 31) >> 5)) out_valid[w + 1] = (uint32_t)(b >> 32);
      }
    }
  }
}

template <typename T>
int gather_launch(const void* src, const uint32_t* src_valid, int64_t src_rows, const int32_t* map, int64_t n,
                  int nullify_oob, void* out, uint32_t* out_valid, hipStream_t s)
{
  int64_t blocks = div_up(n, GATHER_BT * 4);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  if (out_valid)
    hipLaunchKernelGGL((k_gather<T, true>), dim3((unsigned)blocks), dim3(GATHER_BT), 0, s,
                       static_cast<const T*>(src), src_valid, src_rows, map, n, nullify_oob, static_cast<T*>(out),
                       out_valid);
  else
    hipLaunchKernelGGL((k_gather<T, false>), dim3((unsigned)blocks), dim3(GATHER_BT), 0, s,
                       static_cast<const T*>(src), src_valid, src_rows, map, n, nullify_oob, static_cast<T*>(out),
                       out_valid);
  GX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- bitmask kernels
__device__ __forceinline__ uint32_t word_range_mask(int64_t w, int64_t begin_bit, int64_t end_bit)
{
  const int64_t lo = w * 32, hi = lo + 32;
  if (hi <= begin_bit || lo >= end_bit) return 0u;
  uint32_t m = 0xFFFFFFFFu;
  if (begin_bit > lo) m &= 0xFFFFFFFFu << (begin_bit - lo);
  if (end_bit < hi) m &= 0xFFFFFFFFu >> (hi - end_bit);
  return m;
}

__global__ void __launch_bounds__(256) k_bitmask_set(uint32_t* mask, int64_t begin_bit, int64_t end_bit, int valid)
{
  const int64_t w0 = begin_bit >> 5, w1 = (end_bit + 31) >> 5;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t w = w0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < w1; w += stride) {
    const uint32_t m = word_range_mask(w, begin_bit, end_bit);
    if (m == 0xFFFFFFFFu)
      mask[w] = valid ? 0xFFFFFFFFu : 0u;
    else if (m)
      mask[w] = valid ? (mask[w] | m) : (mask[w] & ~m);
  }
}

struct MaskList {
  const uint32_t* m[16];
  int count;
};

// out = AND of masks (null entries skipped); optional popcount of [0, nbits)
__global__ void __launch_bounds__(256) k_bitmask_and_count(MaskList ml, int64_t begin_bit, int64_t end_bit,
                                                           uint32_t* out, unsigned long long* count)
{
  const int64_t w0 = begin_bit >> 5, w1 = (end_bit + 31) >> 5;
  const int64_t stride    = (int64_t)gridDim.x * blockDim.x;
  unsigned long long pops = 0;
  for (int64_t w = w0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < w1; w += stride) {
    uint32_t v = 0xFFFFFFFFu;
    for (int k = 0; k < ml.count; ++k)
      if (ml.m[k]) v &= ml.m[k][w];
    if (out) out[w] = v;
    pops += __builtin_popcount(v & word_range_mask(w, begin_bit, end_bit));
  }
  if (count) {
    pops = wave_reduce(pops, SumOp());
    if (lane_id() == 0 && pops) atomicAdd(count, pops);
  }
}

__global__ void k_set_u64(unsigned long long* p, unsigned long long v) { *p = v; }

__global__ void __launch_bounds__(256) k_first_unset(const uint32_t* mask, int64_t nbits, unsigned long long* pos)
{
  const int64_t nw     = (nbits + 31) >> 5;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long best = (unsigned long long)nbits;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += stride) {
    const uint32_t inv = ~mask[w] & word_range_mask(w, 0, nbits);
    if (inv) {
      const unsigned long long p = (unsigned long long)(w * 32 + __builtin_ctz(inv));
      if (p < best) best = p;
      break;  // later words of this thread are larger
    }
  }
  best = wave_reduce(best, MinOp());
  if (lane_id() == 0 && best < (unsigned long long)nbits) atomicMin(pos, best);
}

// dst bits [dst_begin, dst_begin + nbits) = src bits [src_begin, src_begin + nbits) (src == NULL: all 1).
// One thread per destination word; fully covered words are plain stores, the (at most two) edge words
// merge under a mask with atomics so that neighbouring copies into the same word compose.
__global__ void __launch_bounds__(256) k_bitmask_copy(uint32_t* __restrict__ dst, int64_t dst_begin,
                                                      const uint32_t* __restrict__ src, int64_t src_begin, int64_t nbits)
{
  const int64_t w0     = dst_begin >> 5;
  const int64_t w1     = (dst_begin + nbits + 31) >> 5;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t w = w0 + (int64_t)blockIdx.x * 256 + threadIdx.x; w < w1; w += stride) {
    const int64_t lo = (w << 5) > dst_begin ? (w << 5) : dst_begin;                                  // first dst bit of this word
    const int64_t hi = ((w + 1) << 5) < dst_begin + nbits ? ((w + 1) << 5) : dst_begin + nbits;      // one past the last
    const uint32_t m = (hi - lo == 32) ? 0xFFFFFFFFu : (((1u << (hi - lo)) - 1u) << (lo & 31));
    uint32_t v       = 0xFFFFFFFFu;
    if (src) {
      const int64_t sb  = src_begin + (lo - dst_begin);  // source bit that lands on dst bit `lo`
      const int64_t sw  = sb >> 5;
      const unsigned sh = (unsigned)(sb & 31);
      const int64_t send = (src_begin + nbits + 31) >> 5;
      uint64_t two = src[sw];
      if (sh && sw + 1 < send) two |= (uint64_t)src[sw + 1] << 32;
      v = (uint32_t)(two >> sh) << (lo & 31);
    }
    if (m == 0xFFFFFFFFu) {
      dst[w] = v;
    } else {
      atomicAnd(&dst[w], ~m);
      atomicOr(&dst[w], v & m);
    }
  }
}

static inline unsigned word_grid(int64_t nwords)
{
  int64_t b = div_up(nwords, 256 * 4);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace gx

namespace gx {
// out[j] = rows[idx[j]] + base[segment of idx[j]]: the received (int32 local row) of a sharded join's pair turned into
// a global int64 row id in one gather -- segment s of the receive buffer came from rank s, whose shard starts at base[s]
struct SegBases {
  long long start[17];  // segment s = positions [start[s], start[s + 1])
  long long base[16];
  int nseg;
};
__global__ void __launch_bounds__(256) k_gather_global_rows(const int32_t* __restrict__ rows, const int32_t* __restrict__ idx, int64_t n,
                                                            SegBases sb, long long* __restrict__ out)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) {
    const long long i = idx[j];
    int s             = 0;
    for (int k = 1; k < sb.nseg; ++k) s += (i >= sb.start[k]) ? 1 : 0;
    out[j] = (long long)rows[i] + sb.base[s];
  }
}
// the same with the segment table in device memory (any number of segments; binary search): tab = [nseg + 1] starts | [nseg] bases
__global__ void __launch_bounds__(256) k_gather_global_rows_dev(const int32_t* __restrict__ rows, const int32_t* __restrict__ idx, int64_t n,
                                                                int nseg, const long long* __restrict__ tab, long long* __restrict__ out)
{
  const long long* start = tab;
  const long long* base  = tab + nseg + 1;
  const int64_t stride   = (int64_t)gridDim.x * 256;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) {
    const long long i = idx[j];
    int lo = 0, hi = nseg;  // the segment s with start[s] <= i < start[s + 1] (empty segments are skipped by the search)
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (start[mid] <= i) lo = mid; else hi = mid;
    }
    out[j] = (long long)rows[i] + base[lo];
  }
}
// Encoded global rows of the sharded join -> int64: enc = (source rank << shift) | row.  With `snap` (pair positions at which
// the probe of each chunk started, nchunks + 1 entries) the row counts inside the SENDER's chunk: global = base[src] +
// chunk * chunk_rows[src] + row; without it, global = base[src] + row.  Streaming: no gather, no random access.
// (round 6: the chunk positions -- <= 1024 -- staged in LDS, two pairs per lane per access; one element per trip with the binary
// search's loads from global memory behind each other took 0.8 - 1.7 ms per 3e8 pairs, twice per probe call)
constexpr int DEC_MAXCH = 1024;
__global__ void __launch_bounds__(256) k_decode_global_rows(const int32_t* __restrict__ enc, int64_t n, int shift,
                                                            const long long* __restrict__ bases, const long long* __restrict__ chunk_rows,
                                                            const long long* __restrict__ snap, int nchunks, long long* __restrict__ out)
{
  __shared__ long long s_snap[DEC_MAXCH + 1];
  if (snap)
    for (int i = threadIdx.x; i <= nchunks; i += 256) s_snap[i] = snap[i];
  __syncthreads();
  const uint32_t mask = (1u << shift) - 1u;
  auto chunk_of = [&](int64_t j) -> int {  // the chunk c with snap[c] <= j < snap[c + 1]
    int lo = 0, hi = nchunks;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_snap[mid] <= j) lo = mid; else hi = mid;
    }
    return lo;
  };
  auto decode = [&](uint32_t e, int c) -> long long {
    const uint32_t s = e >> shift;
    return bases[s] + (long long)(e & mask) + (snap ? (long long)c * chunk_rows[s] : 0ll);  // (<= 16 ranks: the tables stay in the L1)
  };
  // two pairs per lane per access: an 8-byte load and ONE 16-byte store, contiguous across the wave (four per lane as two 16-byte stores
  // left every 64-byte line to two instructions: 2.0 ms per 3e8 pairs against 0.8), two such accesses in flight per trip
  const int64_t nh     = n / 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  typedef int32_t i32x2 __attribute__((ext_vector_type(2)));
  typedef long long i64x2 __attribute__((ext_vector_type(2)));
  const bool aligned = ((reinterpret_cast<uintptr_t>(enc) & 7u) | (reinterpret_cast<uintptr_t>(out) & 15u)) == 0;
  auto two = [&](int64_t h, i32x2 e) {
    const int64_t j = h * 2;
    int c0 = 0, c1 = 0;
    if (snap) {
      c0 = chunk_of(j);
      c1 = j + 1 < s_snap[c0 + 1] ? c0 : chunk_of(j + 1);  // (c0 + 1 <= nchunks; snap[nchunks] = the number of pairs)
    }
    *reinterpret_cast<i64x2*>(out + j) = i64x2{decode((uint32_t)e.x, c0), decode((uint32_t)e.y, c1)};
  };
  if (aligned) {
    for (int64_t h = (int64_t)blockIdx.x * 256 + threadIdx.x; h < nh; h += 2 * stride) {
      const int64_t h2 = h + stride;
      const i32x2 ea   = *reinterpret_cast<const i32x2*>(enc + h * 2);
      const i32x2 eb   = h2 < nh ? *reinterpret_cast<const i32x2*>(enc + h2 * 2) : i32x2{0, 0};
      two(h, ea);
      if (h2 < nh) two(h2, eb);
    }
  }
  // the last pair of an odd count (or everything, for buffers that are not aligned)
  for (int64_t j = (aligned ? nh * 2 : 0) + (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += stride)
    out[j] = decode((uint32_t)enc[j], snap ? chunk_of(j) : 0);
}
__global__ void __launch_bounds__(256) k_widen_i32_i64(const int32_t* __restrict__ in, int64_t n, long long* __restrict__ out)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) out[j] = (long long)in[j];
}
}  // namespace gx

extern "C" {

int gx_gather_global_rows_dev(const int32_t* rows, int64_t nrows, const int32_t* idx, int64_t n, int nseg, const int64_t* segtab_dev,
                              int64_t* out, gx_stream_t s)
{
  if (n < 0 || nrows < 0 || nseg < 1 || !segtab_dev) return GX_EINVAL;
  if (n == 0) return 0;
  if (!rows || !idx || !out) return GX_EINVAL;
  int64_t blocks = gx::div_up(n, (int64_t)256 * 4);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(gx::k_gather_global_rows_dev, dim3((unsigned)blocks), dim3(256), 0, s, rows, idx, n, nseg,
                     reinterpret_cast<const long long*>(segtab_dev), reinterpret_cast<long long*>(out));
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_decode_global_rows(const int32_t* enc, int64_t n, int shift, const int64_t* bases_dev, const int64_t* chunk_rows_dev,
                          const int64_t* snap_dev, int nchunks, int64_t* out, gx_stream_t s)
{
  if (n < 0 || shift < 1 || shift > 31 || !bases_dev || (snap_dev && (!chunk_rows_dev || nchunks < 1 || nchunks > gx::DEC_MAXCH))) return GX_EINVAL;
  if (n == 0) return 0;
  if (!enc || !out) return GX_EINVAL;
  int64_t blocks = gx::div_up(n, (int64_t)256 * 8);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gx::k_decode_global_rows, dim3((unsigned)blocks), dim3(256), 0, s, enc, n, shift, reinterpret_cast<const long long*>(bases_dev),
                     reinterpret_cast<const long long*>(chunk_rows_dev), reinterpret_cast<const long long*>(snap_dev), nchunks,
                     reinterpret_cast<long long*>(out));
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_widen_i32_i64(const int32_t* in, int64_t n, int64_t* out, gx_stream_t s)
{
  if (n < 0 || (n > 0 && (!in || !out))) return GX_EINVAL;
  if (n == 0) return 0;
  int64_t blocks = gx::div_up(n, (int64_t)256 * 8);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gx::k_widen_i32_i64, dim3((unsigned)blocks), dim3(256), 0, s, in, n, reinterpret_cast<long long*>(out));
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_gather_global_rows(const int32_t* rows, int64_t nrows, const int32_t* idx, int64_t n, int nseg, const int64_t* seg_counts_host,
                          const int64_t* seg_bases_host, int64_t* out, gx_stream_t s)
{
  if (n < 0 || nrows < 0 || nseg < 1 || nseg > 16 || !seg_counts_host || !seg_bases_host) return GX_EINVAL;
  if (n == 0) return 0;
  if (!rows || !idx || !out) return GX_EINVAL;
  gx::SegBases sb{};
  long long run = 0;
  for (int k = 0; k < nseg; ++k) {
    sb.start[k] = run;
    sb.base[k]  = seg_bases_host[k];
    run += seg_counts_host[k];
  }
  sb.start[nseg] = run;
  sb.nseg        = nseg;
  if (run != nrows) return GX_EINVAL;
  int64_t blocks = gx::div_up(n, (int64_t)256 * 4);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(gx::k_gather_global_rows, dim3((unsigned)blocks), dim3(256), 0, s, rows, idx, n, sb, reinterpret_cast<long long*>(out));
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_gather(int elem_size, const void* src, const uint32_t* src_valid, int64_t src_rows, const int32_t* map,
              int64_t n, int nullify_oob, void* out, uint32_t* out_valid, gx_stream_t s)
{
  if (n < 0 || src_rows < 0) return GX_EINVAL;
  if (n == 0) return 0;
  if (!map || !out || (!src && src_rows > 0)) return GX_EINVAL;
  switch (elem_size) {
    case 1: return gx::gather_launch<uint8_t>(src, src_valid, src_rows, map, n, nullify_oob, out, out_valid, s);
    case 2: return gx::gather_launch<uint16_t>(src, src_valid, src_rows, map, n, nullify_oob, out, out_valid, s);
    case 4: return gx::gather_launch<uint32_t>(src, src_valid, src_rows, map, n, nullify_oob, out, out_valid, s);
    case 8: return gx::gather_launch<uint64_t>(src, src_valid, src_rows, map, n, nullify_oob, out, out_valid, s);
    default: return GX_EDTYPE;
  }
}

int gx_bitmask_set(uint32_t* mask, int64_t begin_bit, int64_t end_bit, int valid, gx_stream_t s)
{
  if (begin_bit < 0 || end_bit < begin_bit) return GX_EINVAL;
  if (end_bit == begin_bit) return 0;
  if (!mask) return GX_EINVAL;
  const int64_t nw = ((end_bit + 31) >> 5) - (begin_bit >> 5);
  hipLaunchKernelGGL(gx::k_bitmask_set, dim3(gx::word_grid(nw)), dim3(256), 0, s, mask, begin_bit, end_bit, valid);
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_bitmask_count(const uint32_t* mask, int64_t begin_bit, int64_t end_bit, int64_t* count_dev, gx_stream_t s)
{
  if (begin_bit < 0 || end_bit < begin_bit || !count_dev) return GX_EINVAL;
  GX_HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(int64_t), s));
  if (end_bit == begin_bit) return 0;
  if (!mask) return GX_EINVAL;
  gx::MaskList ml{};
  ml.m[0]  = mask;
  ml.count = 1;
  const int64_t nw = ((end_bit + 31) >> 5) - (begin_bit >> 5);
  hipLaunchKernelGGL(gx::k_bitmask_and_count, dim3(gx::word_grid(nw)), dim3(256), 0, s, ml, begin_bit, end_bit,
                     (uint32_t*)nullptr, reinterpret_cast<unsigned long long*>(count_dev));
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_bitmask_and(const uint32_t* const* masks_host, int nmasks, int64_t nbits, uint32_t* out, int64_t* count_dev,
                   gx_stream_t s)
{
  if (nmasks < 0 || nmasks > 16 || nbits < 0 || (nmasks > 0 && !masks_host)) return GX_EINVAL;
  if (count_dev) GX_HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(int64_t), s));
  if (nbits == 0) return 0;
  if (!out && !count_dev) return GX_EINVAL;
  gx::MaskList ml{};
  ml.count = nmasks;
  for (int k = 0; k < nmasks; ++k) ml.m[k] = masks_host[k];
  const int64_t nw = (nbits + 31) >> 5;
  hipLaunchKernelGGL(gx::k_bitmask_and_count, dim3(gx::word_grid(nw)), dim3(256), 0, s, ml, (int64_t)0, nbits, out,
                     reinterpret_cast<unsigned long long*>(count_dev));
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_bitmask_copy(uint32_t* dst, int64_t dst_begin_bit, const uint32_t* src, int64_t src_begin_bit, int64_t nbits,
                    gx_stream_t s)
{
  if (nbits < 0 || dst_begin_bit < 0 || src_begin_bit < 0 || (nbits > 0 && !dst)) return GX_EINVAL;
  if (nbits == 0) return 0;
  const int64_t nw = ((dst_begin_bit + nbits + 31) >> 5) - (dst_begin_bit >> 5);
  hipLaunchKernelGGL(gx::k_bitmask_copy, dim3(gx::word_grid(nw)), dim3(256), 0, s, dst, dst_begin_bit, src, src_begin_bit,
                     nbits);
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_bitmask_first_unset(const uint32_t* mask, int64_t nbits, int64_t* pos_dev, gx_stream_t s)
{
  if (nbits < 0 || !pos_dev) return GX_EINVAL;
  hipLaunchKernelGGL(gx::k_set_u64, dim3(1), dim3(1), 0, s, reinterpret_cast<unsigned long long*>(pos_dev),
                     (unsigned long long)nbits);
  if (nbits == 0 || !mask) return 0;
  const int64_t nw = (nbits + 31) >> 5;
  hipLaunchKernelGGL(gx::k_first_unset, dim3(gx::word_grid(nw)), dim3(256), 0, s, mask, nbits,
                     reinterpret_cast<unsigned long long*>(pos_dev));
  GX_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
