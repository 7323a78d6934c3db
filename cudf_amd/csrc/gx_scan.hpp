// gx_scan.hpp -- device-wide scan / reduce building blocks (reduce-then-scan, three launches).
//
// Fixed association order (lane order inside a row, row order, wave order, chunk order), so
// floating-point results are bit-reproducible run to run -- the reason the f64 paths use this
// form rather than a decoupled look-back chain, whose association depends on timing.
// Traffic: reduce pass reads N, apply pass reads N and writes N  (3 x sizeof(T) per element).
#pragma once

#include "gx_common.hpp"

namespace gx {
namespace scan {

constexpr int SCAN_BT  = 256;
constexpr int SCAN_IPT = 16;
constexpr int SCAN_CHUNK = SCAN_BT * SCAN_IPT;

// Loader: element i -> value of the scan's accumulator type (handles nulls / casts).
template <typename InT, typename AccT>
struct PlainLoader {
  const InT* in;
  const uint32_t* valid;
  AccT identity;
  __device__ __forceinline__ AccT operator()(int64_t i) const
  {
    if (valid && !bit_is_set(valid, i)) return identity;
    return static_cast<AccT>(in[i]);
  }
};

// Access pattern: wave w of a chunk owns the contiguous run [w, w+1) * 64 * IPT of it and walks it
// in IPT rows of 64 consecutive items, lane l taking item row * 64 + l -- every load and store is a
// fully coalesced 64-lane access (the earlier thread-contiguous layout had each lane on its own
// cache line: 0.6 TB/s).  Order is kept for non-commutative operators: rows are combined in
// order, lanes inside a row by an inclusive wave scan, waves in order.
// ORDERED = false (commutative operators: cudf::reduce): every lane folds its own column of the rows,
// then one ordered wave fold -- IPT + 6 operator applications per lane instead of 7 * IPT.
template <typename AccT, typename Op, typename Loader, bool ORDERED = true>
__global__ void __launch_bounds__(SCAN_BT) k_chunk_reduce(Loader load, int64_t n, AccT identity, Op op,
                                                          AccT* partials, const int* skip)
{
  if (skip && *skip) return;
  constexpr int NWV = SCAN_BT / GX_WAVE;
  __shared__ AccT s_w[NWV];
  const unsigned l    = lane_id();
  const unsigned w    = threadIdx.x / GX_WAVE;
  const int64_t chunk = blockIdx.x;
  const int64_t base  = chunk * SCAN_CHUNK + (int64_t)w * (GX_WAVE * SCAN_IPT) + l;
  AccT carry          = identity;
  if (ORDERED) {
#pragma unroll 4
    for (int k = 0; k < SCAN_IPT; ++k) {
      const int64_t i = base + (int64_t)k * GX_WAVE;
      const AccT v    = (i < n) ? load(i) : identity;
      const AccT inc  = wave_inclusive_scan(v, op);
      carry           = op(carry, shfl(inc, GX_WAVE - 1));
    }
  } else {
    AccT v[SCAN_IPT];
#pragma unroll
    for (int k = 0; k < SCAN_IPT; ++k) {
      const int64_t i = base + (int64_t)k * GX_WAVE;
      v[k]            = (i < n) ? load(i) : identity;
    }
#pragma unroll
    for (int k = 0; k < SCAN_IPT; ++k) carry = op(carry, v[k]);
    carry = shfl(wave_inclusive_scan(carry, op), GX_WAVE - 1);
  }
  if (l == 0) s_w[w] = carry;
  __syncthreads();
  if (threadIdx.x == 0) {
    AccT t = s_w[0];
#pragma unroll
    for (int k = 1; k < NWV; ++k) t = op(t, s_w[k]);
    partials[chunk] = t;
  }
}

// single block: exclusive scan of the chunk partials in place; partials[np] = grand total
template <typename AccT, typename Op>
__global__ void __launch_bounds__(1024) k_partials_scan(AccT* partials, int64_t np, AccT identity, Op op,
                                                        const int* skip)
{
  if (skip && *skip) return;
  __shared__ AccT s_tmp[1024 / GX_WAVE + 1];
  __shared__ AccT s_carry;
  if (threadIdx.x == 0) s_carry = identity;
  __syncthreads();
  for (int64_t base = 0; base < np; base += 1024) {
    const int64_t i = base + threadIdx.x;
    AccT v          = (i < np) ? partials[i] : identity;
    AccT total;
    AccT exc   = block_exclusive_scan<1024>(v, identity, op, s_tmp, &total);
    AccT carry = s_carry;
    if (i < np) partials[i] = op(carry, exc);
    __syncthreads();
    if (threadIdx.x == 0) s_carry = op(carry, total);
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[np] = s_carry;
}

template <typename AccT, typename OutT, typename Op, typename Loader, bool INCLUSIVE>
__global__ void __launch_bounds__(SCAN_BT) k_chunk_scan(Loader load, int64_t n, AccT identity, Op op,
                                                        const AccT* partials, OutT* out, const int* skip)
{
  if (skip && *skip) return;
  constexpr int NWV = SCAN_BT / GX_WAVE;
  __shared__ AccT s_w[NWV];
  const unsigned l    = lane_id();
  const unsigned w    = threadIdx.x / GX_WAVE;
  const int64_t chunk = blockIdx.x;
  const int64_t base  = chunk * SCAN_CHUNK + (int64_t)w * (GX_WAVE * SCAN_IPT) + l;
  AccT inc[SCAN_IPT];  // inclusive scan of each row inside the wave
  AccT carry = identity;
#pragma unroll
  for (int k = 0; k < SCAN_IPT; ++k) {
    const int64_t i = base + (int64_t)k * GX_WAVE;
    const AccT v    = (i < n) ? load(i) : identity;
    inc[k]          = wave_inclusive_scan(v, op);
    carry           = op(carry, shfl(inc[k], GX_WAVE - 1));
  }
  if (l == 0) s_w[w] = carry;
  __syncthreads();
  AccT run = partials[chunk];  // everything before this chunk, then the waves before this one
  for (unsigned k = 0; k < w; ++k) run = op(run, s_w[k]);
#pragma unroll
  for (int k = 0; k < SCAN_IPT; ++k) {
    const int64_t i = base + (int64_t)k * GX_WAVE;
    if (INCLUSIVE) {
      if (i < n) out[i] = static_cast<OutT>(op(run, inc[k]));
    } else {
      AccT ex = shfl_up(inc[k], 1);
      if (l == 0) ex = identity;
      if (i < n) out[i] = static_cast<OutT>(op(run, ex));
    }
    run = op(run, shfl(inc[k], GX_WAVE - 1));
  }
}

// ---- single-pass scan with decoupled look-back, for EXACT commutative operators (integer sum / product, any
// min / max): 16 B/row instead of the 24 B/row of reduce-then-scan.  Floating-point sums keep the three-launch
// form above: their result must not depend on which predecessor had published what when a tile looked back.
// Tiles are handed out by ticket (every predecessor of a running tile is running or done); a tile publishes its
// aggregate, then folds its predecessors' -- 64 of them per round, one per lane of wave 0 -- up to the nearest
// inclusive prefix, and publishes its own.  An 8-byte accumulator travels as two 8-byte granules
// {flag:2 | 32 value bits}: equal non-zero flags in both halves mean both come from the same publication.
struct LbStatus {
  unsigned long long lo, hi;
};
template <typename AccT>
__device__ __forceinline__ void lb_publish(LbStatus* st, unsigned flag, AccT v)
{
  static_assert(sizeof(AccT) == 8, "look-back scan: 8-byte accumulators");
  unsigned long long bits;
  __builtin_memcpy(&bits, &v, 8);
  store_agent_u64(&st->lo, ((unsigned long long)flag << 62) | (bits & 0xFFFFFFFFull));
  store_agent_u64(&st->hi, ((unsigned long long)flag << 62) | (bits >> 32));
}
template <typename AccT>
__device__ __forceinline__ unsigned lb_read(const LbStatus* st, AccT& v)
{
  unsigned long long l, h;
  unsigned spins = 0;
  for (;;) {
    l = load_agent_u64(&st->lo);
    h = load_agent_u64(&st->hi);
    if ((l >> 62) != 0 && (l >> 62) == (h >> 62)) break;
    // A chain cannot break by construction (tickets: every predecessor is running or done).  If one ever does, fail
    // LOUDLY: the trap aborts the kernel and the error surfaces at the caller's next synchronisation, instead of
    // a silently wrong scan or a hang.
    if (++spins > (1u << 24)) __builtin_trap();
    __builtin_amdgcn_s_sleep(2);
  }
  const unsigned long long bits = (l & 0xFFFFFFFFull) | (h << 32);
  __builtin_memcpy(&v, &bits, 8);
  return (unsigned)(l >> 62);
}

// Tile = 1024 threads x 16 elements: measured on 1e9 uint64 (scripts/xp/xp_scan.hip, profiles/r2_xp_scan.txt) the tile
// COUNT is what bounds this kernel -- 4096-element tiles 4.5 ms, 8192 3.5 ms, 16384 3.3 ms against a 3.0 ms copy;
// the ticket atomic, the wave-scan flavour and 16-byte loads each moved it by < 0.1 ms.  The look-back window is 16
// predecessors per round, not 64: a round waits for the slowest tile in its window (3.8 ms against 4.4 ms at 4096).
constexpr int LB_BT    = 1024;
constexpr int LB_CHUNK = LB_BT * SCAN_IPT;
constexpr int LB_WIN   = 16;

template <typename AccT, typename OutT, typename Op, typename Loader, bool INCLUSIVE>
__global__ void __launch_bounds__(LB_BT) k_lookback_scan(Loader load, int64_t n, AccT identity, Op op, LbStatus* status,
                                                         unsigned int* ticket, OutT* out)
{
  constexpr int NWV = LB_BT / GX_WAVE;
  __shared__ AccT s_w[NWV];
  __shared__ unsigned int s_tile;
  const unsigned l = lane_id();
  const unsigned w = threadIdx.x / GX_WAVE;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  __syncthreads();
  const int64_t tile = s_tile;
  const int64_t base = tile * LB_CHUNK + (int64_t)w * (GX_WAVE * SCAN_IPT) + l;
  AccT inc[SCAN_IPT];  // inclusive scan of each row inside the wave
  AccT carry = identity;
#pragma unroll
  for (int k = 0; k < SCAN_IPT; ++k) {
    const int64_t i = base + (int64_t)k * GX_WAVE;
    inc[k]          = (i < n) ? load(i) : identity;
  }
#pragma unroll
  for (int k = 0; k < SCAN_IPT; ++k) {
    inc[k] = wave_inclusive_scan(inc[k], op);
    carry  = op(carry, read_lane<GX_WAVE - 1>(inc[k]));
  }
  if (l == 0) s_w[w] = carry;
  __syncthreads();
  if (w == 0) {
    // aggregate of the tile: the 16 wave carries, scanned so that every wave finds its own prefix in s_w afterwards
    AccT wv = l < NWV ? s_w[l] : identity;
    wv      = wave_inclusive_scan(wv, op);
    const AccT agg = read_lane<NWV - 1>(wv);
    AccT ex  = identity;
    if (tile == 0) {
      if (l == 0) lb_publish<AccT>(&status[0], 2u, agg);
    } else {
      if (l == 0) lb_publish<AccT>(&status[tile], 1u, agg);
      int64_t pos = tile - 1;
      for (;;) {
        const int64_t idx = pos - (int64_t)l;
        AccT v            = identity;
        unsigned flag     = 2u;  // before the first tile: an (empty) inclusive prefix
        if (l >= (unsigned)LB_WIN) flag = 1u;  // outside the window: an empty aggregate
        else if (idx >= 0) flag = lb_read<AccT>(&status[idx], v);
        const uint64_t m2 = ballot(flag == 2u);
        if (m2 != 0) {  // nearest inclusive prefix: fold it and every aggregate nearer than it
          const unsigned first = (unsigned)__builtin_ctzll(m2);
          const AccT part      = wave_reduce(l <= first ? v : identity, op);
          ex                   = op(part, ex);
          break;
        }
        ex = op(wave_reduce(v, op), ex);
        pos -= LB_WIN;
      }
      if (l == 0) lb_publish<AccT>(&status[tile], 2u, op(ex, agg));
    }
    // s_w[k] <- everything before wave k of this tile
    AccT before = shfl_up(wv, 1);
    if (l == 0) before = identity;
    if (l < NWV) s_w[l] = op(ex, before);
  }
  __syncthreads();
  AccT run = s_w[w];
#pragma unroll
  for (int k = 0; k < SCAN_IPT; ++k) {
    const int64_t i = base + (int64_t)k * GX_WAVE;
    if (INCLUSIVE) {
      if (i < n) out[i] = static_cast<OutT>(op(run, inc[k]));
    } else {
      AccT exl = dpp_take<0x138, 0xF>(inc[k]);  // wave_shr:1
      if (l == 0) exl = identity;
      if (i < n) out[i] = static_cast<OutT>(op(run, exl));
    }
    run = op(run, read_lane<GX_WAVE - 1>(inc[k]));
  }
}

static inline int64_t lb_tiles(int64_t n) { return n > 0 ? div_up(n, (int64_t)LB_CHUNK) : 0; }
static inline int64_t num_chunks(int64_t n) { return n > 0 ? div_up(n, SCAN_CHUNK) : 0; }
// scratch bytes of the single-pass scan over n elements
static inline size_t lookback_bytes(int64_t n) { return (size_t)(lb_tiles(n) + 1) * sizeof(LbStatus) + 256; }

// out may alias the loader's input (each tile is fully loaded before it is written).
template <typename AccT, typename OutT, typename Op, typename Loader>
int device_scan_lookback(Loader load, int64_t n, AccT identity, Op op, bool inclusive, OutT* out, void* scratch, hipStream_t stream)
{
  if (n <= 0) return 0;
  const int64_t nc = lb_tiles(n);
  GX_HIP_TRY(hipMemsetAsync(scratch, 0, lookback_bytes(n), stream));
  unsigned int* ticket = static_cast<unsigned int*>(scratch);
  LbStatus* status     = reinterpret_cast<LbStatus*>(static_cast<char*>(scratch) + 256);
  if (inclusive)
    hipLaunchKernelGGL((k_lookback_scan<AccT, OutT, Op, Loader, true>), dim3((unsigned)nc), dim3(LB_BT), 0, stream, load, n, identity,
                       op, status, ticket, out);
  else
    hipLaunchKernelGGL((k_lookback_scan<AccT, OutT, Op, Loader, false>), dim3((unsigned)nc), dim3(LB_BT), 0, stream, load, n,
                       identity, op, status, ticket, out);
  GX_LAUNCH_CHECK();
  return 0;
}
// scratch elements of AccT needed for a scan over n elements
static inline size_t partials_count(int64_t n) { return (size_t)num_chunks(n) + 1; }

// out may alias the loader's input (each chunk is fully loaded before it is written).
template <typename AccT, typename OutT, typename Op, typename Loader>
int device_scan(Loader load, int64_t n, AccT identity, Op op, bool inclusive, OutT* out, AccT* partials,
                hipStream_t stream, const int* skip = nullptr)
{
  if (n <= 0) return 0;
  const int64_t nc = num_chunks(n);
  hipLaunchKernelGGL((k_chunk_reduce<AccT, Op, Loader>), dim3((unsigned)nc), dim3(SCAN_BT), 0, stream, load, n,
                     identity, op, partials, skip);
  hipLaunchKernelGGL((k_partials_scan<AccT, Op>), dim3(1), dim3(1024), 0, stream, partials, nc, identity, op,
                     skip);
  if (inclusive)
    hipLaunchKernelGGL((k_chunk_scan<AccT, OutT, Op, Loader, true>), dim3((unsigned)nc), dim3(SCAN_BT), 0,
                       stream, load, n, identity, op, partials, out, skip);
  else
    hipLaunchKernelGGL((k_chunk_scan<AccT, OutT, Op, Loader, false>), dim3((unsigned)nc), dim3(SCAN_BT), 0,
                       stream, load, n, identity, op, partials, out, skip);
  GX_LAUNCH_CHECK();
  return 0;
}

// Streaming reduce: a capped grid of workgroups walks the column with a grid stride, every lane keeping its own
// accumulator over 8 independent loads per trip (64 B per lane in flight), then one ordered fold per workgroup.
// The earlier form launched one 4096-element workgroup per chunk (244 k workgroups at 1e9 rows, each ending in a
// barrier and a serial fold): 1.5 TB/s; a pure 8 B/row read should run near the copy ceiling.
// The association order depends on n only (fixed grid, fixed lane / wave / workgroup order): results stay
// bit-reproducible run to run.  Commutative operators only (cudf::reduce).
constexpr int RED_MAX_BLOCKS = 2048;
static inline int64_t reduce_blocks(int64_t n)
{
  const int64_t nc = num_chunks(n);
  return nc < RED_MAX_BLOCKS ? nc : RED_MAX_BLOCKS;
}
template <typename AccT, typename Op, typename Loader>
__global__ void __launch_bounds__(SCAN_BT) k_stream_reduce(Loader load, int64_t n, AccT identity, Op op, AccT* partials)
{
  constexpr int U   = 8;
  constexpr int NWV = SCAN_BT / GX_WAVE;
  __shared__ AccT s_w[NWV];
  const int64_t stride = (int64_t)gridDim.x * SCAN_BT * U;
  AccT acc             = identity;
  for (int64_t i0 = (int64_t)blockIdx.x * SCAN_BT * U + threadIdx.x; i0 < n; i0 += stride) {
    AccT v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * SCAN_BT;
      v[u]            = (i < n) ? load(i) : identity;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc = op(acc, v[u]);
  }
  acc = shfl(wave_inclusive_scan(acc, op), GX_WAVE - 1);
  if (lane_id() == 0) s_w[threadIdx.x / GX_WAVE] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    AccT t = s_w[0];
#pragma unroll
    for (int k = 1; k < NWV; ++k) t = op(t, s_w[k]);
    partials[blockIdx.x] = t;
  }
}

// device-wide reduce: partials[reduce_blocks(n)] receives the result (as AccT)
template <typename AccT, typename Op, typename Loader>
int device_reduce(Loader load, int64_t n, AccT identity, Op op, AccT* partials, hipStream_t stream)
{
  const int64_t nb = reduce_blocks(n);
  if (nb > 0)
    hipLaunchKernelGGL((k_stream_reduce<AccT, Op, Loader>), dim3((unsigned)nb), dim3(SCAN_BT), 0, stream, load, n, identity, op,
                       partials);
  hipLaunchKernelGGL((k_partials_scan<AccT, Op>), dim3(1), dim3(1024), 0, stream, partials, nb, identity, op,
                     (const int*)nullptr);
  GX_LAUNCH_CHECK();
  return 0;
}

}  // namespace scan
}  // namespace gx
