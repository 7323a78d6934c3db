// gx_common.hpp -- device/host helpers shared by the gfx950 kernels of libcudf_amd.
// wave = 64 lanes everywhere (CDNA4); no rocPRIM/hipCUB/thrust in the product.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <stdint.h>

#include "../../include/cudf_amd/gx.h"

#define GX_WAVE 64

#define GX_HIP_TRY(expr)                           \
  do {                                             \
    hipError_t _e = (expr);                        \
    if (_e != hipSuccess) return (int)_e;          \
  } while (0)

#define GX_LAUNCH_CHECK()                          \
  do {                                             \
    hipError_t _e = hipGetLastError();             \
    if (_e != hipSuccess) return (int)_e;          \
  } while (0)

namespace gx {

__host__ __device__ static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
__host__ __device__ static inline int64_t div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

// carve helper for caller-provided scratch
struct Carver {
  char* base;
  size_t off;
  explicit Carver(void* p) : base(static_cast<char*>(p)), off(0) {}
  template <typename T>
  T* take(size_t count)
  {
    off       = align_up(off, 256);
    T* r      = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return r;
  }
  size_t total() const { return align_up(off, 256); }
};

// ---------------------------------------------------------------- device primitives
__device__ __forceinline__ unsigned lane_id()
{
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// cross-lane moves for any trivially copyable type (moved as 32-bit words)
template <typename T, typename F>
__device__ __forceinline__ T shfl_words(T v, F f)
{
  static_assert(sizeof(T) % 4 == 0, "shuffle payload must be a multiple of 4 bytes");
  uint32_t w[sizeof(T) / 4];
  __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
  for (unsigned k = 0; k < sizeof(T) / 4; ++k) w[k] = f(w[k]);
  T r;
  __builtin_memcpy(&r, w, sizeof(T));
  return r;
}
template <typename T>
__device__ __forceinline__ T shfl_up(T v, unsigned delta)
{
  return shfl_words(v, [delta](uint32_t x) { return (uint32_t)__shfl_up((int)x, delta, GX_WAVE); });
}
template <typename T>
__device__ __forceinline__ T shfl(T v, int src)
{
  return shfl_words(v, [src](uint32_t x) { return (uint32_t)__shfl((int)x, src, GX_WAVE); });
}
template <typename T>
__device__ __forceinline__ T shfl_xor(T v, int m)
{
  return shfl_words(v, [m](uint32_t x) { return (uint32_t)__shfl_xor((int)x, m, GX_WAVE); });
}

// Wave-wide match on an 8-bit digit: for every lane, the lanes holding the same digit.  Returns the
// number of such lanes below this one (the stable rank inside the wave) and their total count.
// One ballot per digit bit; the per-lane 64-bit mask update m &= (bit ? v : ~v) is ONE gfx950
// v_bitop3_b32 per half (truth table 0x90: a & ~(b ^ c) with c = 0 / ~0 from the lane's bit), which
// halves the VALU work of the and/xor/cndmask sequence the compiler emits for the generic form.
// `active` = ballot of the lanes that take part (others must pass live == false).
template <int NBITS>
__device__ __forceinline__ void match_rank(uint32_t d, bool live, uint64_t active, uint32_t& lower, uint32_t& count)
{
  uint32_t m_lo = (uint32_t)active, m_hi = (uint32_t)(active >> 32);
#pragma unroll
  for (int b = 0; b < NBITS; ++b) {
    const int beta   = __builtin_amdgcn_sbfe(d, b, 1);  // 0 or -1
    const uint64_t v = ballot(live && beta != 0);
    m_lo             = __builtin_amdgcn_bitop3_b32(m_lo, (uint32_t)v, (uint32_t)beta, 0x90);
    m_hi             = __builtin_amdgcn_bitop3_b32(m_hi, (uint32_t)(v >> 32), (uint32_t)beta, 0x90);
  }
  lower = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
  count = (uint32_t)__builtin_popcount(m_lo) + (uint32_t)__builtin_popcount(m_hi);
}

// Rank of a key among the keys of its tile that go to the same bin, for FEW bins (<= 1 << NBITS, NBITS <= 4): the lanes of a
// wave that share a bin are found with NBITS ballots, their leader reserves the whole group's ranks with ONE returning LDS
// atomic and hands the base to the others through ds_bpermute.  With 1..16 bins the plain per-lane atomic serialises 4 to 64
// lanes on one LDS address per instruction (the exchange partition of the sharded operators: 8 destination ranks).
template <int NBITS>
__device__ __forceinline__ uint32_t lds_rank_few(uint32_t* counters, uint32_t d, bool live)
{
  const uint64_t active = ballot(live);
  uint32_t m_lo = (uint32_t)active, m_hi = (uint32_t)(active >> 32);
#pragma unroll
  for (int b = 0; b < NBITS; ++b) {
    const int beta   = __builtin_amdgcn_sbfe(d, b, 1);  // 0 or -1
    const uint64_t v = ballot(live && beta != 0);
    m_lo             = __builtin_amdgcn_bitop3_b32(m_lo, (uint32_t)v, (uint32_t)beta, 0x90);
    m_hi             = __builtin_amdgcn_bitop3_b32(m_hi, (uint32_t)(v >> 32), (uint32_t)beta, 0x90);
  }
  const uint32_t lower = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
  const uint32_t count = (uint32_t)__builtin_popcount(m_lo) + (uint32_t)__builtin_popcount(m_hi);
  const uint64_t m     = ((uint64_t)m_hi << 32) | m_lo;
  const int leader     = m ? __builtin_ctzll(m) : (int)lane_id();
  uint32_t base        = 0;
  if (live && lower == 0) base = atomicAdd(&counters[d], count);
  base = (uint32_t)__builtin_amdgcn_ds_bpermute(leader << 2, (int)base);
  return live ? base + lower : 0u;
}

__device__ __forceinline__ void match_rank8(uint32_t d, bool live, uint64_t active, uint32_t& lower, uint32_t& count)
{
  match_rank<8>(d, live, active, lower, count);
}

// ---- in-wave bitonic sort of 64-bit keys (one or two keys per lane), ascending over
// element index e = r * 64 + lane.  "Flip" form of the network: every compare-exchange keeps the
// minimum in the lower element, so the only per-step state is a constant lane mask.  Lane moves:
// xor 1/2/3 = DPP quad_perm, xor 7 / 15 = DPP row_half_mirror / row_mirror (VALU, no LDS traffic);
// larger distances go through ds_bpermute.
template <int M>
__device__ __forceinline__ uint32_t lane_xor32(uint32_t v)
{
  if constexpr (M == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
  else if constexpr (M == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);
  else if constexpr (M == 3) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x1B, 0xF, 0xF, false);
  else if constexpr (M == 7) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);
  else if constexpr (M == 15) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);
  else return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane_id() ^ (unsigned)M) << 2), (int)v);
}
template <int M>
__device__ __forceinline__ uint64_t lane_xor64(uint64_t v)
{
  return ((uint64_t)lane_xor32<M>((uint32_t)(v >> 32)) << 32) | lane_xor32<M>((uint32_t)v);
}
// compare-exchange with the lane at distance xor M; LOWBIT: lanes with (lane & LOWBIT) == 0 keep the minimum
template <int M, int LOWBIT>
__device__ __forceinline__ void cmpx(uint64_t& k)
{
  const uint64_t o   = lane_xor64<M>(k);
  const bool keepmin = (lane_id() & (unsigned)LOWBIT) == 0;
  k                  = ((k < o) == keepmin) ? k : o;
}
__device__ __forceinline__ void wave_bitonic64(uint64_t& k)
{
  cmpx<1, 1>(k);
  cmpx<3, 2>(k); cmpx<1, 1>(k);
  cmpx<7, 4>(k); cmpx<2, 2>(k); cmpx<1, 1>(k);
  cmpx<15, 8>(k); cmpx<4, 4>(k); cmpx<2, 2>(k); cmpx<1, 1>(k);
  cmpx<31, 16>(k); cmpx<8, 8>(k); cmpx<4, 4>(k); cmpx<2, 2>(k); cmpx<1, 1>(k);
  cmpx<63, 32>(k); cmpx<16, 16>(k); cmpx<8, 8>(k); cmpx<4, 4>(k); cmpx<2, 2>(k); cmpx<1, 1>(k);
}
__device__ __forceinline__ void wave_halfclean64(uint64_t& k)
{
  cmpx<32, 32>(k); cmpx<16, 16>(k); cmpx<8, 8>(k); cmpx<4, 4>(k); cmpx<2, 2>(k); cmpx<1, 1>(k);
}
// 128 keys: k0 = elements 0..63, k1 = elements 64..127
__device__ __forceinline__ void wave_bitonic128(uint64_t& k0, uint64_t& k1)
{
  wave_bitonic64(k0);
  wave_bitonic64(k1);
  const uint64_t o1 = lane_xor64<63>(k1), o0 = lane_xor64<63>(k0);  // element e pairs with e ^ 127
  k0 = k0 < o1 ? k0 : o1;
  k1 = k1 < o0 ? o0 : k1;
  wave_halfclean64(k0);
  wave_halfclean64(k1);
}

// Unstable rank of each live lane inside its LDS bin: a returning LDS atomic per lane, except when the
// whole wave hits ONE bin (sorted or low-cardinality input), where 64 same-address atomics would
// serialise: then one lane reserves the run and the lanes number themselves.
// (round 4) ... and when a good part of the wave shares the first live lane's bin (a hot value: 10 % / 96 % of a tile's keys are
// one key): that group is numbered through one atomic, the other lanes take theirs as before.
__device__ __forceinline__ uint32_t lds_rank(uint32_t* counters, uint32_t d, bool live)
{
  const uint64_t act = ballot(live);
  if (act == 0) return 0;
  const int leader    = __builtin_ctzll(act);
  const uint32_t dl   = shfl(d, leader);
  const uint64_t same = ballot(live && d == dl);
  if (same == act || __builtin_popcountll(same) >= 8) {
    uint32_t base = 0;
    if ((int)lane_id() == leader) base = atomicAdd(&counters[dl], (uint32_t)__builtin_popcountll(same));
    base = shfl(base, leader);
    if (live && d == dl) return base + (uint32_t)__builtin_popcountll(same & lanemask_lt());
    return live ? atomicAdd(&counters[d], 1u) : 0u;  // (only when same != act)
  }
  return live ? atomicAdd(&counters[d], 1u) : 0u;
}

struct SumOp {
  template <typename T>
  __device__ __forceinline__ T operator()(T a, T b) const { return a + b; }
};
struct MinOp {
  template <typename T>
  __device__ __forceinline__ T operator()(T a, T b) const { return b < a ? b : a; }
};
struct MaxOp {
  template <typename T>
  __device__ __forceinline__ T operator()(T a, T b) const { return a < b ? b : a; }
};
struct ProdOp {
  template <typename T>
  __device__ __forceinline__ T operator()(T a, T b) const { return a * b; }
};

// DPP lane movement (gfx9 family): every 32-bit word of v moved by the DPP pattern CTRL under row mask RM; lanes the
// pattern leaves without a source keep their own value.
template <int CTRL, int RM, typename T>
__device__ __forceinline__ T dpp_take(T v)
{
  return shfl_words(v, [](uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, RM, 0xF, false); });
}
// value of lane `src` (a compile-time lane) in every lane: v_readlane per word, no LDS crossbar trip
template <int SRC, typename T>
__device__ __forceinline__ T read_lane(T v)
{
  return shfl_words(v, [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)x, SRC); });
}
// Inclusive scan across the 64 lanes of a wave: row_shr:1/2/4/8 inside each row of 16 lanes, row_bcast:15 into
// rows 1 and 3, row_bcast:31 into rows 2 and 3 -- six VALU steps per word.  (The earlier __shfl_up form cost one
// ds_bpermute per word and step on the LDS pipe all four SIMDs of a CU share.)  Lower lanes are always the LEFT
// operand, so any associative operator keeps its order.
template <typename T, typename Op>
__device__ __forceinline__ T wave_inclusive_scan(T v, Op op)
{
  const unsigned l = lane_id();
  { const T o = dpp_take<0x111, 0xF>(v); if ((l & 15u) >= 1u) v = op(o, v); }
  { const T o = dpp_take<0x112, 0xF>(v); if ((l & 15u) >= 2u) v = op(o, v); }
  { const T o = dpp_take<0x114, 0xF>(v); if ((l & 15u) >= 4u) v = op(o, v); }
  { const T o = dpp_take<0x118, 0xF>(v); if ((l & 15u) >= 8u) v = op(o, v); }
  { const T o = dpp_take<0x142, 0xA>(v); if (l & 16u) v = op(o, v); }
  { const T o = dpp_take<0x143, 0xC>(v); if (l >= 32u) v = op(o, v); }
  return v;
}

template <typename T, typename Op>
__device__ __forceinline__ T wave_reduce(T v, Op op)
{
#pragma unroll
  for (int d = GX_WAVE / 2; d >= 1; d >>= 1) v = op(v, shfl_xor(v, d));
  return v;
}

// Block-wide exclusive scan of one value per thread.  `lds` needs BT/64 + 1 elements of T.
// Every thread of the block must call it.  Returns the exclusive prefix; *total (optional) the
// block aggregate.  Association order is fixed (lane order inside a wave, wave order across), so
// floating-point results are deterministic.
template <int BT, typename T, typename Op>
__device__ __forceinline__ T block_exclusive_scan(T v, T identity, Op op, T* lds, T* total)
{
  constexpr int NW = BT / GX_WAVE;
  const unsigned l = lane_id();
  const unsigned w = threadIdx.x / GX_WAVE;
  T inc            = wave_inclusive_scan(v, op);
  if (l == GX_WAVE - 1) lds[w] = inc;
  __syncthreads();
  if (w == 0) {
    T x = (l < NW) ? lds[l] : identity;
    T s = wave_inclusive_scan(x, op);
    if (l < NW) lds[l] = s;  // inclusive over waves
  }
  __syncthreads();
  T wave_prefix = (w == 0) ? identity : lds[w - 1];
  T exc         = shfl_up(inc, 1);
  if (l == 0) exc = identity;
  if (total) *total = lds[NW - 1];
  T r = op(wave_prefix, exc);
  __syncthreads();  // lds may be reused by the caller
  return r;
}

// ---- sub-bucket sort without a sorting network.  `cnt` <= 128 keys that agree on every bit above
// `sh + 8`: (1) wave-private 256-bin counting split on byte [sh, sh+8) (one returning LDS atomic per
// key, an exclusive scan of the 256 counters held four per lane), which leaves only keys with the
// same byte adjacent and possibly out of order; (2) when no bin holds more than four keys -- the
// common case: cnt / 256 keys per bin on average -- four phases of odd-even transposition, two
// consecutive elements per lane (even phase in-lane, odd phase through DPP wave shifts), finish the
// job.  ~75 VALU per sub-bucket against ~250 (64 keys) to ~650 (128 keys) for the bitonic network.
// Returns false (keys untouched) when a bin is too full; the caller falls back to the network.
__device__ __forceinline__ uint64_t dpp_wave_shl1_u64(uint64_t v, uint64_t fill)
{  // lane i receives lane i+1; lane 63 receives `fill`
  const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)fill, (int)(uint32_t)v, 0x130, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(fill >> 32), (int)(uint32_t)(v >> 32), 0x130, 0xF, 0xF, false);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ uint64_t dpp_wave_shr1_u64(uint64_t v, uint64_t fill)
{  // lane i receives lane i-1; lane 0 receives `fill`
  const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)fill, (int)(uint32_t)v, 0x138, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(fill >> 32), (int)(uint32_t)(v >> 32), 0x138, 0xF, 0xF, false);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
// On success the sorted keys are returned in registers, lane l holding elements 2l (k0) and 2l+1 (k1);
// the caller stores them (to LDS or straight to global memory).
__device__ __forceinline__ bool wave_split_sort(uint64_t* keys, uint32_t cnt, uint32_t* cw /* 256 wave-private counters, 16-B aligned */,
                                                int sh, uint64_t& k0, uint64_t& k1)
{
  const unsigned lane = lane_id();
  const uint32_t e0 = 2 * lane, e1 = 2 * lane + 1;
  k0 = e0 < cnt ? keys[e0] : ~0ull;
  k1 = e1 < cnt ? keys[e1] : ~0ull;
  uint4* cw4  = reinterpret_cast<uint4*>(cw);
  cw4[lane]   = make_uint4(0u, 0u, 0u, 0u);
  const uint32_t b0 = (uint32_t)(k0 >> sh) & 0xFFu, b1 = (uint32_t)(k1 >> sh) & 0xFFu;
  const uint32_t r0 = e0 < cnt ? atomicAdd(&cw[b0], 1u) : 0u;
  const uint32_t r1 = e1 < cnt ? atomicAdd(&cw[b1], 1u) : 0u;
  uint4 c = cw4[lane];
  uint32_t mx = c.x > c.y ? c.x : c.y;
  mx          = c.z > mx ? c.z : mx;
  mx          = c.w > mx ? c.w : mx;
  if (ballot(mx > 4u) != 0) return false;  // wave-uniform
  const uint32_t sum = c.x + c.y + c.z + c.w;
  const uint32_t inc = wave_inclusive_scan(sum, SumOp());
  uint32_t run       = inc - sum;
  uint4 st;
  st.x = run; run += c.x;
  st.y = run; run += c.y;
  st.z = run; run += c.z;
  st.w = run;
  cw4[lane] = st;
  if (e0 < cnt) keys[cw[b0] + r0] = k0;
  if (e1 < cnt) keys[cw[b1] + r1] = k1;
  k0 = e0 < cnt ? keys[e0] : ~0ull;
  k1 = e1 < cnt ? keys[e1] : ~0ull;
#pragma unroll
  for (int ph = 0; ph < 2; ++ph) {
    {  // even phase: (2l, 2l+1) inside the lane
      const bool sw     = k1 < k0;
      const uint64_t lo = sw ? k1 : k0, hi = sw ? k0 : k1;
      k0 = lo;
      k1 = hi;
    }
    {  // odd phase: (2l+1, 2l+2): lane l's k1 with lane l+1's k0
      const uint64_t nxt = dpp_wave_shl1_u64(k0, ~0ull);  // k0 of lane + 1
      const uint64_t prv = dpp_wave_shr1_u64(k1, 0ull);   // k1 of lane - 1
      k1                 = nxt < k1 ? nxt : k1;
      k0                 = prv > k0 ? prv : k0;
    }
  }
  return true;
}

// Inclusive +-scan of one uint32 per lane over the wave with DPP row shifts and row broadcasts (gfx9 family:
// row_shr:1/2/4/8 inside each row of 16 lanes, row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3):
// 6 VALU steps, no LDS crossbar round trips (the __shfl_up form costs one ds_bpermute + wait per step).
__device__ __forceinline__ uint32_t wave_inclusive_sum_dpp(uint32_t v)
{
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
  return v;
}

// wave_split_sort for TWO sub-buckets at once (A and B, each <= 128 keys): the two dependency chains -- zero the
// counters, rank with returning LDS atomics, scan, scatter, read back, odd-even clean-up -- are issued interleaved,
// so each LDS round trip of one hides behind the other's.  The 256 counters of a sub-bucket are 16-bit halves of
// 128 words (counts <= 128), which fits both arrays into the wave's 1 KiB counter row `cw`.  Returns bit 0 / bit 1
// = A / B sorted into (a0, a1) / (b0, b1), lane l holding elements 2l and 2l + 1; a sub-bucket whose bit is clear
// had a bin with more than four keys and is untouched in LDS (the caller falls back to the network).
// cntB == 0: only A is processed.
__device__ __forceinline__ unsigned wave_split_sort_x2(uint64_t* keysA, uint32_t cntA, uint64_t* keysB, uint32_t cntB,
                                                       uint32_t* cw /* 256 words, 16-B aligned */, int sh, uint64_t& a0,
                                                       uint64_t& a1, uint64_t& b0, uint64_t& b1)
{
  const unsigned lane = lane_id();
  const uint32_t e0 = 2 * lane, e1 = 2 * lane + 1;
  uint32_t* cwA = cw;
  uint32_t* cwB = cw + 128;
  a0 = e0 < cntA ? keysA[e0] : ~0ull;
  a1 = e1 < cntA ? keysA[e1] : ~0ull;
  b0 = e0 < cntB ? keysB[e0] : ~0ull;
  b1 = e1 < cntB ? keysB[e1] : ~0ull;
  reinterpret_cast<uint2*>(cwA)[lane] = make_uint2(0u, 0u);
  reinterpret_cast<uint2*>(cwB)[lane] = make_uint2(0u, 0u);
  const uint32_t ba0 = (uint32_t)(a0 >> sh) & 0xFFu, ba1 = (uint32_t)(a1 >> sh) & 0xFFu;
  const uint32_t bb0 = (uint32_t)(b0 >> sh) & 0xFFu, bb1 = (uint32_t)(b1 >> sh) & 0xFFu;
  // rank inside the bin: old value of the bin's 16-bit half
  uint32_t ra0 = 0, ra1 = 0, rb0 = 0, rb1 = 0;
  if (e0 < cntA) ra0 = atomicAdd(&cwA[ba0 >> 1], 1u << (16 * (ba0 & 1)));
  if (e0 < cntB) rb0 = atomicAdd(&cwB[bb0 >> 1], 1u << (16 * (bb0 & 1)));
  if (e1 < cntA) ra1 = atomicAdd(&cwA[ba1 >> 1], 1u << (16 * (ba1 & 1)));
  if (e1 < cntB) rb1 = atomicAdd(&cwB[bb1 >> 1], 1u << (16 * (bb1 & 1)));
  ra0 = (ra0 >> (16 * (ba0 & 1))) & 0xFFFFu;
  ra1 = (ra1 >> (16 * (ba1 & 1))) & 0xFFFFu;
  rb0 = (rb0 >> (16 * (bb0 & 1))) & 0xFFFFu;
  rb1 = (rb1 >> (16 * (bb1 & 1))) & 0xFFFFu;
  const uint2 ca = reinterpret_cast<uint2*>(cwA)[lane];  // bins 4l .. 4l + 3
  const uint2 cb = reinterpret_cast<uint2*>(cwB)[lane];
  const uint32_t ca0 = ca.x & 0xFFFFu, ca1 = ca.x >> 16, ca2 = ca.y & 0xFFFFu, ca3 = ca.y >> 16;
  const uint32_t cb0 = cb.x & 0xFFFFu, cb1 = cb.x >> 16, cb2 = cb.y & 0xFFFFu, cb3 = cb.y >> 16;
  uint32_t mxa = ca0 > ca1 ? ca0 : ca1, mxb = cb0 > cb1 ? cb0 : cb1;
  mxa = ca2 > mxa ? ca2 : mxa;
  mxb = cb2 > mxb ? cb2 : mxb;
  mxa = ca3 > mxa ? ca3 : mxa;
  mxb = cb3 > mxb ? cb3 : mxb;
  const bool okA = cntA > 0 && ballot(mxa > 4u) == 0;  // wave-uniform
  const bool okB = cntB > 0 && ballot(mxb > 4u) == 0;
  const uint32_t suma = ca0 + ca1 + ca2 + ca3, sumb = cb0 + cb1 + cb2 + cb3;
  uint32_t runa = wave_inclusive_sum_dpp(suma) - suma;
  uint32_t runb = wave_inclusive_sum_dpp(sumb) - sumb;
  uint2 sa, sb2;
  sa.x = runa | ((runa + ca0) << 16);
  runa += ca0 + ca1;
  sa.y = runa | ((runa + ca2) << 16);
  sb2.x = runb | ((runb + cb0) << 16);
  runb += cb0 + cb1;
  sb2.y = runb | ((runb + cb2) << 16);
  if (okA) reinterpret_cast<uint2*>(cwA)[lane] = sa;
  if (okB) reinterpret_cast<uint2*>(cwB)[lane] = sb2;
  if (okA) {
    if (e0 < cntA) keysA[((cwA[ba0 >> 1] >> (16 * (ba0 & 1))) & 0xFFFFu) + ra0] = a0;
    if (e1 < cntA) keysA[((cwA[ba1 >> 1] >> (16 * (ba1 & 1))) & 0xFFFFu) + ra1] = a1;
  }
  if (okB) {
    if (e0 < cntB) keysB[((cwB[bb0 >> 1] >> (16 * (bb0 & 1))) & 0xFFFFu) + rb0] = b0;
    if (e1 < cntB) keysB[((cwB[bb1 >> 1] >> (16 * (bb1 & 1))) & 0xFFFFu) + rb1] = b1;
  }
  if (okA) {
    a0 = e0 < cntA ? keysA[e0] : ~0ull;
    a1 = e1 < cntA ? keysA[e1] : ~0ull;
  }
  if (okB) {
    b0 = e0 < cntB ? keysB[e0] : ~0ull;
    b1 = e1 < cntB ? keysB[e1] : ~0ull;
  }
#pragma unroll
  for (int ph = 0; ph < 2; ++ph) {
    {  // even phase: (2l, 2l+1) inside the lane
      const bool swa     = a1 < a0;
      const uint64_t loa = swa ? a1 : a0, hia = swa ? a0 : a1;
      a0 = loa;
      a1 = hia;
      const bool swb     = b1 < b0;
      const uint64_t lob = swb ? b1 : b0, hib = swb ? b0 : b1;
      b0 = lob;
      b1 = hib;
    }
    {  // odd phase: (2l+1, 2l+2): lane l's k1 with lane l+1's k0
      const uint64_t nxa = dpp_wave_shl1_u64(a0, ~0ull), pva = dpp_wave_shr1_u64(a1, 0ull);
      const uint64_t nxb = dpp_wave_shl1_u64(b0, ~0ull), pvb = dpp_wave_shr1_u64(b1, 0ull);
      a1 = nxa < a1 ? nxa : a1;
      a0 = pva > a0 ? pva : a0;
      b1 = nxb < b1 ? nxb : b1;
      b0 = pvb > b0 ? pvb : b0;
    }
  }
  return (okA ? 1u : 0u) | (okB ? 2u : 0u);
}

// bits -> unsigned key whose unsigned order is the cudf order (KIND 0 unsigned, 1 signed, 2 float).
//  signed: sign flip.  float: -0.0 -> +0.0, NaN -> all ones (after +Inf; all NaNs equivalent),
//  then the IEEE total-order flip.  desc_mask (0 or ~0) reverses the order.  Equal sortable bits
//  <=> equivalent under the row comparator (NaN == NaN, -0 == +0).
//  K_FTOTAL (round 5): the IEEE total-order flip alone -- a BIJECTION (from_sortable below), which is the cudf order exactly on
//  columns that hold no NaN and no -0.0 ("clean" float columns: the sort's unordered levels check every key and fall back).
enum KeyKind { K_UNSIGNED = 0, K_SIGNED = 1, K_FLOAT = 2, K_FTOTAL = 3 };
template <typename U, int KIND>
__host__ __device__ __forceinline__ U to_sortable(U bits, U desc_mask)
{
  constexpr U SIGN = U(U(1) << (sizeof(U) * 8 - 1));
  if (KIND == K_SIGNED) {
    bits ^= SIGN;
  } else if (KIND == K_FTOTAL) {
    bits ^= (bits & SIGN) ? U(~U(0)) : SIGN;
  } else if (KIND == K_FLOAT) {
    constexpr U EXP = (sizeof(U) == 8) ? U(0x7FF0000000000000ull) : U(0x7F800000u);
    const U mag     = bits & U(~SIGN);
    if (mag > EXP) {
      bits = U(~U(0));
    } else {
      if (mag == 0) bits = 0;
      bits ^= (bits & SIGN) ? U(~U(0)) : SIGN;
    }
  }
  return bits ^ desc_mask;
}

// sortable form -> the key's own bits: the inverse of to_sortable for the kinds where it has one (integers: the transform is an
// involution; K_FTOTAL: the flip undone).  K_FLOAT has none (-0.0 and NaN payloads are merged): its paths carry the original bits.
template <typename U, int KIND>
__host__ __device__ __forceinline__ U from_sortable(U s, U desc_mask)
{
  static_assert(KIND != K_FLOAT, "K_FLOAT's sortable form is not invertible");
  constexpr U SIGN = U(U(1) << (sizeof(U) * 8 - 1));
  s ^= desc_mask;
  if (KIND == K_SIGNED) return U(s ^ SIGN);
  if (KIND == K_FTOTAL) return (s & SIGN) ? U(s ^ SIGN) : U(~s);
  return s;
}
// a float key that K_FTOTAL must not see: NaN (all of them sort last, in input order) or -0.0 (equivalent to +0.0)
template <typename U>
__host__ __device__ __forceinline__ bool float_unclean(U bits)
{
  constexpr U SIGN = U(U(1) << (sizeof(U) * 8 - 1));
  constexpr U EXP  = (sizeof(U) == 8) ? U(0x7FF0000000000000ull) : U(0x7F800000u);
  return (U)(bits & U(~SIGN)) > EXP || bits == SIGN;
}

// splitmix64: counter-based generator for synthetic data and checksums
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// validity bitmap: LSB-first bits in uint32 words, 1 = valid (cudf/utilities/bit.hpp:47-61)
__device__ __forceinline__ bool bit_is_set(const uint32_t* mask, int64_t i)
{
  return (mask[i >> 5] >> (i & 31)) & 1u;
}
__device__ __forceinline__ bool row_valid(const uint32_t* mask, int64_t i)
{
  return mask == nullptr || bit_is_set(mask, i);
}

// MurmurHash3_x86_32 finaliser / block (published algorithm; the reference delegates to
// cuco::murmurhash3_32: include/cudf/hashing/detail/murmurhash3_x86_32.cuh:16,45)
__host__ __device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h)
{
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}
__host__ __device__ __forceinline__ uint32_t mm3_block(uint32_t h, uint32_t k)
{
  k *= 0xcc9e2d51u;
  k = rotl32(k, 15);
  k *= 0x1b873593u;
  h ^= k;
  h = rotl32(h, 13);
  return h * 5u + 0xe6546b64u;
}
__host__ __device__ __forceinline__ uint32_t murmur3_u32(uint32_t v, uint32_t seed)
{
  return fmix32(mm3_block(seed, v) ^ 4u);
}
__host__ __device__ __forceinline__ uint32_t murmur3_u64(uint64_t v, uint32_t seed)
{
  uint32_t h = mm3_block(seed, (uint32_t)v);
  h          = mm3_block(h, (uint32_t)(v >> 32));
  return fmix32(h ^ 8u);
}
// tail-only variants for 1- and 2-byte elements
__host__ __device__ __forceinline__ uint32_t murmur3_tail(uint32_t k, uint32_t len, uint32_t seed)
{
  k *= 0xcc9e2d51u;
  k = rotl32(k, 15);
  k *= 0x1b873593u;
  return fmix32((seed ^ k) ^ len);
}
__host__ __device__ __forceinline__ uint32_t hash_combine32(uint32_t lhs, uint32_t rhs)
{
  return lhs ^ (rhs + 0x9e3779b9u + (lhs << 6) + (lhs >> 2));
}

// ---- per-THREAD sorting networks over 16 registers (the local sort's window passes, gx_sort.hip k_local_sort).
// sort16_regs: the 60-comparator, 10-layer network for 16 inputs (Green's; checked with the 0-1 principle over all 2^16
// inputs, tests/test_kernel_formulas.py).  merge16_regs: Batcher's odd-even merge of two ascending runs v[0..7] and
// v[8..15], 25 comparators.  Indices are compile-time constants, so the 16 keys never leave their registers.
#define GX_CE(a, b)                          \
  {                                          \
    const bool s_ = v[b] < v[a];             \
    const T lo_   = s_ ? v[b] : v[a];        \
    v[b]          = s_ ? v[a] : v[b];        \
    v[a]          = lo_;                     \
  }
template <typename T>
__device__ __forceinline__ void sort16_regs(T (&v)[16])
{
  GX_CE(0, 13); GX_CE(1, 12); GX_CE(2, 15); GX_CE(3, 14); GX_CE(4, 8); GX_CE(5, 6);
  GX_CE(7, 11); GX_CE(9, 10); GX_CE(0, 5); GX_CE(1, 7); GX_CE(2, 9); GX_CE(3, 4);
  GX_CE(6, 13); GX_CE(8, 14); GX_CE(10, 15); GX_CE(11, 12); GX_CE(0, 1); GX_CE(2, 3);
  GX_CE(4, 5); GX_CE(6, 8); GX_CE(7, 9); GX_CE(10, 11); GX_CE(12, 13); GX_CE(14, 15);
  GX_CE(0, 2); GX_CE(1, 3); GX_CE(4, 10); GX_CE(5, 11); GX_CE(6, 7); GX_CE(8, 9);
  GX_CE(12, 14); GX_CE(13, 15); GX_CE(1, 2); GX_CE(3, 12); GX_CE(4, 6); GX_CE(5, 7);
  GX_CE(8, 10); GX_CE(9, 11); GX_CE(13, 14); GX_CE(1, 4); GX_CE(2, 6); GX_CE(5, 8);
  GX_CE(7, 10); GX_CE(9, 13); GX_CE(11, 14); GX_CE(2, 4); GX_CE(3, 6); GX_CE(9, 12);
  GX_CE(11, 13); GX_CE(3, 5); GX_CE(6, 8); GX_CE(7, 9); GX_CE(10, 12); GX_CE(3, 4);
  GX_CE(5, 6); GX_CE(7, 8); GX_CE(9, 10); GX_CE(11, 12); GX_CE(6, 7); GX_CE(8, 9);
}
template <typename T>
__device__ __forceinline__ void merge16_regs(T (&v)[16])
{
  GX_CE(0, 8); GX_CE(1, 9); GX_CE(2, 10); GX_CE(3, 11); GX_CE(4, 12); GX_CE(5, 13);
  GX_CE(6, 14); GX_CE(7, 15); GX_CE(4, 8); GX_CE(5, 9); GX_CE(6, 10); GX_CE(7, 11);
  GX_CE(2, 4); GX_CE(3, 5); GX_CE(6, 8); GX_CE(7, 9); GX_CE(10, 12); GX_CE(11, 13);
  GX_CE(1, 2); GX_CE(3, 4); GX_CE(5, 6); GX_CE(7, 8); GX_CE(9, 10); GX_CE(11, 12);
  GX_CE(13, 14);
}
#undef GX_CE

// XCD-aware tile mapping: the dispatcher is observed to place block b on XCD b % 8
// (MI355X_MICROARCH.md, "Workgroup dispatch").  Give each XCD a contiguous range of tiles so
// neighbouring tiles share an L2 (speed only; correctness never depends on it).
__device__ __forceinline__ int64_t xcd_swizzle(int64_t bid, int64_t nblocks)
{
  constexpr int64_t NXCD = 8;
  const int64_t full     = nblocks / NXCD * NXCD;
  if (bid >= full) return bid;  // ragged tail keeps its index
  const int64_t per = full / NXCD;
  return (bid % NXCD) * per + bid / NXCD;
}

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

__device__ __forceinline__ void store_agent_u64(unsigned long long* p, unsigned long long v)
{
  __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long load_agent_u64(const unsigned long long* p)
{
  return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace gx
