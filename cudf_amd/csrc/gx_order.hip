// gx_order.hip -- cudf::sorted_order of ONE 64-bit column as a KEYS-ONLY sort of 64-bit words (round 6).
//
// Replaces cub::DeviceRadixSort::SortPairs over (key copy, iota) at cpp/src/sort/sorted_order_radix.cu:56-179 (stable: :81), whose cost
// does not depend on the value distribution.  Round 3-5's pairs path (gx_sort.hip: look-back levels on key BITS into fixed 8192 / 16384
// key cells) sorts uniform keys in 21 ms per 1e9 rows and DECLINES every uneven column -- bell-shaped, lognormal, Zipf-like integers,
// any real-valued float64 -- to 4-8 LSD pair passes: 65-93 ms.  The keys-only sort got its distribution insensitivity in round 5
// (sample-chosen splitters, warped cells, big-cell rescue: 11-15 ms on anything); this file puts the argsort ON it:
//
//   word(i) = rank34(key_i) << ib | i          ib = bits(n - 1),  rank34 = a MONOTONE 64 - ib bit image of the key
//
// sorted ascending as plain uint64 keys.  Rows come out ordered by rank, ties of the rank by row -- which IS the stable order wherever
// the rank separates distinct keys, and wherever it does not (two different keys, one rank) the rows sit in ONE run of equal ranks and a
// last pass puts that run right by (key, row).
//
// The rank has to do two things at once: keep DISTINCT keys apart (else runs grow), and leave the words EVENLY dense at every scale the
// word sort cuts on -- its cells are sized for 93 % full at 1e9 rows, and a first version that gave each of 4096 sample-quantile buckets
// the same share of the rank space (Gamma(4) noise: +-50 % in true mass) overflowed its cells and fell to eight LSD passes: 94.8 ms
// (profiles/r6_run8_*).  So:
//   * 4096 buckets cut at every 4th of 16384 sorted sample keys (k_om_plan: one workgroup, bitonic sort in LDS);
//   * their MASS is measured, not assumed: k_om_count looks every 32nd 64-key chunk up (3 % of the column, 0.25 B/row) and a bucket's
//     share of the rank space is its share of that count (+-1 %) -- k_om_plan2: base_b, alloc_b;
//   * inside a bucket of width W_b: alloc_b >= W_b ("lossless"): every key value owns F_b = alloc_b / W_b consecutive ranks and its rows
//     are spread over them BY ROW NUMBER, rank = base_b + (key - lo_b) F_b + (row F_b >> ib) -- monotone in (key, row), distinct keys in
//     disjoint intervals, and a value of 10^5 copies is 10^5 evenly spread words instead of one cell of the word sort (the reference
//     benchmark's own keys in [100, 10001): cpp/benchmarks/sort/sort.cpp:24-26); alloc_b < W_b ("lossy"): rank = base_b + (key - lo_b) >> sh_b,
//     distinct keys may share a rank -> the fix-up.  A value that fills several quantiles (consecutive equal splitters) gets a bucket of
//     its own (the empty one in front of its successor's) with W = 1.
// Uniform random 64-bit keys at 1e9 rows: every bucket is 2^52 wide against 2^22 ranks: lossy by 30 bits, 5.8 % of the rows share a rank
// with a neighbour and have their keys fetched again (a 2- or 3-row insertion sort per run).  Runs beyond 16 rows go to a list and are
// sorted per workgroup (LDS up to 4096 rows, a network in global scratch beyond: a density spike the samples cannot see -- slow, correct).
//
// Cost per 1e9 rows: plan 0.3 + map 16 B/row + word sort 48 B/row (11-15 ms) + finish 12 B/row + the runs' gathers; descending order and
// float64 (NaN last and equal to each other, -0.0 == 0.0: to_sortable<K_FLOAT>) go through the same sortable form, ties by row in both
// directions (sorted_order_radix.cu:81; sort_column_impl.cuh:35-57 for the NaN / zero rule).
#include "gx_common.hpp"

extern "C" size_t gx_sort_plan_bytes(void);  // gx_sort.hip: sizeof(SortPlan) -- the header at the start of a sort's scratch (status word inside)

namespace gx {
namespace order {

constexpr int OM_S     = 16384;  // sample keys
constexpr int OM_B     = 4096;   // buckets (one per 4 samples)
constexpr int OM_LUT  = 16384;  // cells of the bucket look-up table over the splitters' range: the binary search starts inside one cell
constexpr int OM_CSTRIDE = 32;  // k_om_count looks at every 32nd 64-key chunk
constexpr int OM_SMALL = 16;     // runs of equal ranks up to this length are put right by the thread that finds them
constexpr int OM_LDSRUN = 4096;  // ... up to this one by a workgroup in LDS
constexpr int OM_FIN_WGS = 8192; // workgroups of the finish passes (each owns a contiguous range of the sorted words and a segment of the run list)

struct alignas(256) OmPlan {
  uint64_t lo[OM_B];      // first sortable key of bucket b (lo[0] = 0; buckets 0 and OM_B - 1 are catch-alls beyond the sample's reach)
  uint64_t base[OM_B];    // first rank of bucket b
  uint64_t mul[OM_B];     // M_b = alloc_b 2^64 / (W_b << lz_b): umulhi(d << lz_b, M_b) = floor(d alloc_b / W_b), d = key - lo_b
  uint32_t meta[OM_B];    // lz_b (bits 0-5) | lossy (bit 8) | eq (bit 9: lo[b] is a repeated splitter -- rows with key == lo[b] belong to
                          // bucket b - 1, which is empty otherwise and holds that one VALUE)
  uint32_t cnt[OM_B];     // sampled rows per bucket (k_om_count)
  uint16_t lut[OM_LUT + 8];  // lut[c] = bucket of the first key of cell c of [lo[1], lo[OM_B - 1]] (cells of 2^gsh keys); lut[OM_LUT] = OM_B - 1
  int gsh;
  uint64_t invn;          // floor((2^64 - 1) / n): row * invn = the row's position in [0, 1) as a 64-bit fraction
  unsigned int nlong;     // entries of the long-run list
  unsigned int nwg;       // ... of which the workgroup pass has to sort (k_om_medium's list)
  unsigned long long nan_count;  // float64 keys: NaN rows (k_om_map) -- a descending order has them first, in REVERSE row order
  unsigned int long_overflow;
  int ib, rb;
  int nlossy;
};
struct LongRun {
  unsigned int start, len;
};

__device__ __forceinline__ int bitlen64(uint64_t x) { return x ? 64 - __builtin_clzll(x) : 0; }

// ---- 1. sample: OM_S keys at even strides, in sortable form
template <int KIND>
__global__ void __launch_bounds__(256) k_om_sample(const uint64_t* __restrict__ keys, int64_t n, uint64_t desc_mask, uint64_t* __restrict__ samp)
{
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= OM_S) return;
  const int64_t i = (int64_t)(((__int128)j * n) / OM_S);
  samp[j]         = to_sortable<uint64_t, KIND>(keys[i < n ? i : n - 1], desc_mask);
}

// ---- 2. plan: sort the sample (bitonic, 128 KiB of LDS), cut 4096 buckets, size their shifts
__global__ void __launch_bounds__(1024) k_om_plan(const uint64_t* __restrict__ samp, OmPlan* plan, int ib, int64_t n)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* s = reinterpret_cast<uint64_t*>(smem);
  const int t = threadIdx.x;
  for (int i = t; i < OM_S; i += 1024) s[i] = samp[i];
  __syncthreads();
  for (int k = 2; k <= OM_S; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < OM_S; i += 1024) {
        const int p = i ^ j;
        if (p > i) {
          const uint64_t a = s[i], b = s[p];
          const bool up    = (i & k) == 0;
          if ((a > b) == up) {
            s[i] = b;
            s[p] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  constexpr int SPB = OM_S / OM_B;  // samples per bucket
  // bucket 0 = [0, ext_min) and bucket OM_B - 1 = (ext_max, 2^64) are CATCH-ALLS for keys beyond the sample's reach (the sample minimum /
  // maximum pushed out by 16 times the extent of the end bucket): rare rows, one or two ranks, put right by the run pass.  Without them
  // the end buckets ran to the ends of the key SPACE and a smooth tail's quarter million rows shared a handful of ranks.
  auto lo_of = [&](int b) -> uint64_t {
    if (b == 0) return 0ull;
    if (b == 1) {
      const uint64_t w = s[2 * SPB] - s[0];
      const uint64_t e = w > (~0ull >> 4) ? ~0ull : w * 16;
      return s[0] > e ? s[0] - e : 0ull;
    }
    if (b == OM_B - 1) {
      const uint64_t top = s[OM_S - 1], w = top - s[(OM_B - 2) * SPB];
      const uint64_t e   = w > (~0ull >> 4) ? ~0ull : w * 16 + 1;
      return top >= ~0ull - e ? ~0ull : top + e + 1;
    }
    // a GAP bucket -- 64 times wider than its neighbours together: two clusters with nothing in between -- holds its rows at its two
    // ends; mapped linearly, each end was one rank: two runs of 10^5 distinct keys for the long-run pass (55 ms for the two-cluster
    // column of the robustness block).  Its boundaries move INWARD by 16 neighbour widths, so the neighbours take its rows and the
    // gap itself becomes a (nearly) empty bucket.  Boundary b moves up when bucket b is a gap, down when bucket b - 1 is.
    const uint64_t c0 = s[b * SPB];
    auto raw = [&](int q) -> uint64_t { return s[q * SPB]; };
    auto is_gap = [&](int q) -> bool {  // bucket q = [raw(q), raw(q + 1)), 2 <= q <= OM_B - 3
      if (q < 2 || q > OM_B - 3) return false;
      const uint64_t w = raw(q + 1) - raw(q), wl = raw(q) - raw(q - 1), wr = raw(q + 2) - raw(q + 1);
      return (w >> 6) > wl && (w >> 6) > wr && (w >> 6) > wl + wr;
    };
    const bool gap_here = is_gap(b), gap_before = is_gap(b - 1);
    if (gap_here && !gap_before) return c0 + 16 * (c0 - raw(b - 1)) + 1;      // (16 wl < w / 4: stays inside the bucket)
    if (gap_before && !gap_here) return c0 - 16 * (raw(b + 1) - c0);          // (16 wr < w / 4)
    return c0;
  };
  for (int b = t; b < OM_B; b += 1024) {
    const uint64_t lo = lo_of(b);
    plan->lo[b]   = lo;
    plan->cnt[b]  = 0;
    // a repeated splitter: the previous bucket [lo, lo) is empty -- it becomes the bucket of the VALUE lo (never a catch-all or its neighbour)
    plan->meta[b] = (b >= 3 && b <= OM_B - 2 && lo_of(b - 1) == lo) ? 512u : 0u;
  }
  // look-up table: the bucket of the first key of each of 16384 equal cells of [lo[1], lo[OM_B - 1]]
  {
    const uint64_t l1 = lo_of(1), span = lo_of(OM_B - 1) - l1;
    int gsh = bitlen64(span) - 14;
    if (gsh < 0) gsh = 0;
    for (int c = t; c <= OM_LUT; c += 1024) {
      int b = OM_B - 1;
      if (c < OM_LUT) {
        const uint64_t off = (uint64_t)c << gsh;
        if (off <= span) {
          const uint64_t key = l1 + off;
          b = 1;  // (key >= lo[1])
          for (int step = OM_B / 2; step > 0; step >>= 1) {
            const int q = b + step;
            if (q < OM_B && lo_of(q) <= key) b = q;
          }
        }
      }
      plan->lut[c] = (uint16_t)b;
    }
    if (t == 0) plan->gsh = gsh;
  }
  if (t == 0) {
    plan->invn          = ~0ull / (uint64_t)(n > 0 ? n : 1);
    plan->nlong         = 0;
    plan->nwg           = 0;
    plan->nan_count     = 0;
    plan->long_overflow = 0;
    plan->ib            = ib;
    plan->rb            = 64 - ib;
    plan->nlossy        = 0;
  }
}

// bucket of sortable key s: upper bound over lo[] -- the look-up table narrows it to the buckets that start inside the key's cell (one
// or two for evenly spread keys: 12 dependent LDS reads per key were 8.2 of the first version's 39 ms), a binary search finishes; a key
// that IS a repeated splitter goes to the (otherwise empty) bucket in front
__device__ __forceinline__ int om_bucket(const uint64_t* __restrict__ lo, const uint16_t* __restrict__ lut, const uint32_t* __restrict__ meta, int gsh, uint64_t s)
{
  const uint64_t l1 = lo[1];
  if (s < l1) return 0;
  uint64_t c = (s - l1) >> gsh;
  if (c > (uint64_t)(OM_LUT - 1)) c = OM_LUT - 1;
  int bl = lut[c], bh = lut[c + 1];  // lo[bl] <= first key of the cell <= s; every bucket beyond bh starts behind the cell
  if (c == (uint64_t)(OM_LUT - 1)) bh = OM_B - 1;
  while (bl < bh) {
    const int mid = (bl + bh + 1) >> 1;
    if (lo[mid] <= s) bl = mid; else bh = mid - 1;
  }
  if ((meta[bl] & 512u) && s == lo[bl]) --bl;
  return bl;
}

// ---- 2b. count: every 32nd 64-key chunk looked up -> sampled mass per bucket
template <int KIND>
__global__ void __launch_bounds__(256) k_om_count(const uint64_t* __restrict__ keys, int64_t n, uint64_t desc_mask, OmPlan* plan)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* s_lo  = reinterpret_cast<uint64_t*>(smem);
  uint32_t* s_eq  = reinterpret_cast<uint32_t*>(smem + OM_B * 8);
  uint32_t* s_cnt = s_eq + OM_B;
  uint16_t* s_lut = reinterpret_cast<uint16_t*>(s_cnt + OM_B);
  for (int i = threadIdx.x; i < OM_B; i += 256) {
    s_lo[i]  = plan->lo[i];
    s_eq[i]  = plan->meta[i];
    s_cnt[i] = 0;
  }
  for (int i = threadIdx.x; i <= OM_LUT; i += 256) s_lut[i] = plan->lut[i];
  __syncthreads();
  const int gsh = plan->gsh;
  const int64_t nchunks = (n + 63) / 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t c = ((int64_t)blockIdx.x * 4 + w) * OM_CSTRIDE; c < nchunks; c += (int64_t)gridDim.x * 4 * OM_CSTRIDE) {
    const int64_t i = c * 64 + lane;
    if (i < n) atomicAdd(&s_cnt[om_bucket(s_lo, s_lut, s_eq, gsh, to_sortable<uint64_t, KIND>(keys[i], desc_mask))], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < OM_B; i += 256)
    if (s_cnt[i]) atomicAdd(&plan->cnt[i], s_cnt[i]);
}

// ---- 2c. plan2: rank space per bucket in proportion to its sampled mass; per-bucket map
__global__ void __launch_bounds__(1024) k_om_plan2(OmPlan* plan)
{
  __shared__ unsigned long long s_part[1024];
  __shared__ unsigned int s_lossy;
  const int t  = threadIdx.x;
  const int rb = plan->rb;
  if (t == 0) s_lossy = 0;
  // total sampled mass
  unsigned long long loc = 0;
  for (int b = t * 4; b < t * 4 + 4; ++b) loc += plan->cnt[b];
  s_part[t] = loc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (t < o) s_part[t] += s_part[t + o];
    __syncthreads();
  }
  const unsigned long long total = s_part[0] ? s_part[0] : 1ull;
  __syncthreads();
  // every bucket: a floor of R / 2^20 ranks (a bucket the sample never hit may still hold rows), the rest in proportion
  const unsigned long long R    = rb >= 64 ? ~0ull : (1ull << rb);
  const unsigned long long rmin = R >> 20 ? R >> 20 : 1ull;
  const unsigned long long rest = R - rmin * OM_B - OM_B;  // (- OM_B: the divisions below round down, the shares never add up to more)
  unsigned long long al[4];
  loc = 0;
  for (int k = 0; k < 4; ++k) {
    const unsigned long long c = plan->cnt[t * 4 + k];
    al[k] = rmin + (unsigned long long)(((__uint128_t)c * rest) / total);
    loc += al[k];
  }
  s_part[t] = loc;
  __syncthreads();
  // exclusive scan over the 1024 thread sums (one wave of work: 4096 buckets once per sort)
  if (t == 0) {
    unsigned long long run = 0;
    for (int i = 0; i < 1024; ++i) {
      const unsigned long long v = s_part[i];
      s_part[i]                  = run;
      run += v;
    }
  }
  __syncthreads();
  unsigned long long base = s_part[t];
  for (int k = 0; k < 4; ++k) {
    const int b = t * 4 + k;
    const uint64_t lo = plan->lo[b];
    // width of the bucket's key range: [lo, next lo); the bucket of a repeated splitter's VALUE (eq of b + 1) holds one value
    uint64_t wm1;
    if (b + 1 < OM_B) {
      const uint64_t nx = plan->lo[b + 1];
      wm1               = ((plan->meta[b + 1] & 512u) || nx <= lo) ? 0ull : nx - lo - 1;
    } else {
      wm1 = ~0ull - lo;
    }
    // M = alloc 2^64 / (W << lz) with W << lz in [2^63, 2^64]: umulhi(d << lz, M) = floor(d alloc / W) (to within the floor of M)
    const int lz          = wm1 ? __builtin_clzll(wm1) : 63;
    const __uint128_t wal = ((__uint128_t)wm1 + 1) << lz;
    const uint64_t M      = (uint64_t)((((__uint128_t)al[k]) << 64) / wal);
    // lossless: every key value owns >= 1 rank (alloc >= 2 W keeps the floors apart); lossy: distinct keys may share a rank
    const bool lossy = al[k] < 2 * ((unsigned long long)wm1 + 1) || wm1 >= (1ull << 62) || b == 0 || b == OM_B - 1;  // (the end buckets clamp)
    if (lossy) atomicAdd(&s_lossy, 1u);
    plan->base[b] = base;
    plan->mul[b]  = M;
    plan->meta[b] = (plan->meta[b] & 512u) | (unsigned)lz | (lossy ? 256u : 0u);
    base += al[k];
  }
  __syncthreads();
  if (t == 0) plan->nlossy = (int)s_lossy;
}

// the rank of sortable key s in row `row`
__device__ __forceinline__ uint64_t om_rank(const uint64_t* __restrict__ lo, const uint16_t* __restrict__ lut, const uint32_t* __restrict__ meta,
                                            const uint64_t* __restrict__ base, const uint64_t* __restrict__ mul, int gsh, uint64_t s, uint64_t row, uint64_t invn)
{
  const int b       = om_bucket(lo, lut, meta, gsh, s);
  const uint32_t mt = meta[b];
  const int lz      = (int)(mt & 63u);
  const uint64_t d  = s - lo[b];
  const uint64_t M  = mul[b];
  const uint64_t A  = __umul64hi(d << lz, M);
  // The value d owns the ranks [A, B), B = floor((d + 1) alloc / W); its rows are spread over them by row / n -- NOT row >> ib: 1e9 rows
  // fill 93 % of 2^30, the rows of every value would crowd into the first 93 % of its ranks and the word sort's cells, sized for 93 %
  // full, would overflow by the thousand (measured: keys in [100, 10001), 476 M of 1e9 rows through the big-cell rescue, 78.6 ms).
  // Lossless bucket: B - A >= 2.  Lossy bucket: B - A is 0 or 1 where the bucket is wider than its rank space (the value shares rank A
  // with its neighbours: the run pass) -- the same formula, no branch.
  const uint64_t d1 = (d + 1) << lz;                  // ((W << lz) = 2^64 wraps to 0)
  const uint64_t B  = d1 ? __umul64hi(d1, M) : M;     // (d + 1 = W and W << lz = 2^64: floor(W alloc / W) = alloc = M)
  return base[b] + A + __umul64hi(row * invn, B - A);
}

// ---- 3. map: word(i) = rank << ib | i
template <int KIND>
__global__ void __launch_bounds__(1024) k_om_map(const uint64_t* __restrict__ keys, int64_t n, uint64_t desc_mask, OmPlan* __restrict__ plan,
                                                 uint64_t* __restrict__ words)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* s_lo   = reinterpret_cast<uint64_t*>(smem);
  uint64_t* s_base = s_lo + OM_B;
  uint64_t* s_mul  = s_base + OM_B;
  uint32_t* s_meta = reinterpret_cast<uint32_t*>(s_mul + OM_B);
  uint16_t* s_lut  = reinterpret_cast<uint16_t*>(s_meta + OM_B);
  for (int i = threadIdx.x; i < OM_B; i += 1024) {
    s_lo[i]   = plan->lo[i];
    s_base[i] = plan->base[i];
    s_mul[i]  = plan->mul[i];
    s_meta[i] = plan->meta[i];
  }
  for (int i = threadIdx.x; i <= OM_LUT; i += 1024) s_lut[i] = plan->lut[i];
  __syncthreads();
  const int gsh        = plan->gsh;
  const int ib         = plan->ib;
  const uint64_t invn  = plan->invn;
  constexpr int U      = 4;
  const int64_t stride = (int64_t)gridDim.x * 1024 * U;
  unsigned int nans    = 0;
  for (int64_t i0 = (int64_t)blockIdx.x * 1024 * U + threadIdx.x; i0 < n; i0 += stride) {
    uint64_t k[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * 1024;
      k[u]            = i < n ? __builtin_nontemporal_load(&keys[i]) : 0ull;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * 1024;
      if (i < n) {
        if (KIND == K_FLOAT && (k[u] & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull) ++nans;
        const uint64_t r = om_rank(s_lo, s_lut, s_meta, s_base, s_mul, gsh, to_sortable<uint64_t, KIND>(k[u], desc_mask), (uint64_t)i, invn);
        __builtin_nontemporal_store((r << ib) | (uint64_t)i, &words[i]);
      }
    }
  }
  if (KIND == K_FLOAT) {  // (wave-uniform branch: a column without NaN costs one ballot per wave)
    const unsigned int wsum = wave_reduce(nans, SumOp());
    if (wsum && lane_id() == 0) atomicAdd(&plan->nan_count, (unsigned long long)wsum);
  }
}

// float64, DESCENDING: the reference sorts the pair (isnan * (row + 1), value) downwards (cpp/src/sort/sorted_order_radix.cu:37-48), so the
// NaN rows come first in REVERSE row order; the word sort leaves them first in row order: turn the block around
__global__ void __launch_bounds__(256) k_om_reverse_nans(int32_t* __restrict__ out, const OmPlan* __restrict__ plan)
{
  const unsigned long long c = plan->nan_count;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < c / 2; i += (unsigned long long)gridDim.x * 256) {
    const int32_t a = out[i], b = out[c - 1 - i];
    out[i]         = b;
    out[c - 1 - i] = a;
  }
}

// ---- 5. finish: rows out; runs of equal ranks in lossy buckets are put right by (key, row).
// Two kernels.  A first version did both in one pass -- the thread that found a run's head fetched the run's keys and sorted them on the
// spot -- and took ~25 ms per 1e9 rows: a wave sat out two or three DEPENDENT memory round trips (the run's words, then its keys, ~2 us
// each under load) for the two or three heads among its 64 rows.  Now pass A only streams: every row's index goes out as it stands, the
// heads of lossy runs are appended to the workgroup's OWN segment of a run list (an LDS counter: no global atomic); pass B gives every
// listed run a thread of its own, so all lanes of all waves have a gather in flight.
struct RunSeg {  // per workgroup of pass A
  unsigned int count, pad;
};

// A listed head is its position; bit 31 = the run has EXACTLY two rows (seen in pass A whenever the word behind the run lies in the thread's
// or its right neighbour's quad): pass B then needs neither the sorted words nor the probe for the run's end -- it reads the two rows pass A
// wrote.  skip_lossless (one key column): runs inside a lossless bucket are not listed at all (one key value, already in row order).
constexpr unsigned int OM_HEAD_L2 = 1u << 31;
__global__ void __launch_bounds__(256) k_om_finish_a(const uint64_t* __restrict__ sorted, int64_t n, const OmPlan* __restrict__ plan, int32_t* __restrict__ out,
                                                     unsigned int* __restrict__ heads, RunSeg* __restrict__ segs, int64_t chunk, unsigned int seg_cap,
                                                     int skip_lossless)
{
  __shared__ uint64_t s_bl[OM_B];  // first rank of bucket b | lossy << 63
  // 0: every run is listed (no lossless bucket, or several key columns); 1: the head's bucket is looked up; 2: nothing is listed (every bucket lossless)
  const int nlossy = plan->nlossy;
  const int lmode  = !skip_lossless || nlossy == OM_B ? 0 : (nlossy == 0 ? 2 : 1);
  if (lmode == 1)
    for (int i = threadIdx.x; i < OM_B; i += 256) s_bl[i] = plan->base[i] | ((plan->meta[i] & 256u) ? (1ull << 63) : 0ull);
  // pure streaming: 8 bytes in and 4 bytes out per row, the neighbours' ranks through wave shuffles (the wave's edge lanes load
  // theirs), the heads of runs of equal ranks appended -- position only -- to the workgroup's own segment through an LDS counter
  __shared__ unsigned int s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const int ib         = plan->ib;
  const uint64_t imask = (1ull << ib) - 1;
  const int64_t p0 = (int64_t)blockIdx.x * chunk, p1 = p0 + chunk < n ? p0 + chunk : n;  // (chunk is a multiple of 1024: whole waves of quads)
  unsigned int* mine = heads + (size_t)blockIdx.x * seg_cap;
  const unsigned lane = threadIdx.x & 63u;
  typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
  typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
  // FOUR consecutive words per thread: two 16-byte loads, one 16-byte store (the sorted words start 256-byte aligned, p0 is a multiple of
  // 1024).  The edge lanes' neighbour words are requested together with the wave's own -- asked for where they are used, behind the
  // shuffles, they were a second dependent round trip in every iteration (2 words per thread and that order: 4.0 ms per 1e9 rows).
  for (int64_t pb = p0; pb < p1; pb += 1024) {
    const int64_t p = pb + 4 * (int64_t)threadIdx.x;
    uint64_t w[4]   = {0, 0, 0, 0};
    uint64_t ep = ~0ull, en = ~0ull;  // the word before the thread's first / behind its last (edge lanes only)
    const bool full = p + 3 < p1;
    if (full) {
      const u64x2 a = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(sorted + p));
      const u64x2 c = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(sorted + p + 2));
      w[0] = a.x;
      w[1] = a.y;
      w[2] = c.x;
      w[3] = c.y;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (p + k < p1) w[k] = sorted[p + k];
    }
    const bool edge_n = lane == 63 || p + 4 >= p1;
    if (lane == 0 && p < p1 && p > 0) ep = sorted[p - 1];
    if (edge_n && p < p1) {  // the word behind the thread's last live one
      const int64_t q = (p + 4 < p1 ? p + 4 : p1);
      if (q < n) en = sorted[q];
    }
    uint64_t r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = w[k] >> ib;
    uint64_t rp = __shfl_up(r[3], 1), rn = __shfl_down(r[0], 1);  // the previous lane's last word, the next lane's first
    const uint64_t rn2 = __shfl_down(r[1], 1);                       // ... and its second
    if (lane == 0) rp = ep == ~0ull ? ~0ull : ep >> ib;
    if (edge_n) rn = en == ~0ull ? ~0ull : en >> ib;
    if (p >= p1) continue;
    if (full) {
      __builtin_nontemporal_store(i32x4{(int32_t)(w[0] & imask), (int32_t)(w[1] & imask), (int32_t)(w[2] & imask), (int32_t)(w[3] & imask)},
                                  reinterpret_cast<i32x4*>(out + p));
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (p + k < p1) out[p + k] = (int32_t)(w[k] & imask);  // right unless the row sits in a run of a lossy bucket: pass B rewrites those
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (p + k >= p1) break;
      const uint64_t prev = k == 0 ? rp : r[k - 1];
      // the next rank: the thread's next live word, else the word behind the workgroup's range (en)
      const uint64_t next = (k < 3 && p + k + 1 < p1) ? r[k + 1] : rn;
      if (prev != r[k] && next == r[k]) {  // word k heads a run
        if (lmode == 2) continue;
        if (lmode == 1) {
          int b = 0;
#pragma unroll
          for (int step = OM_B / 2; step > 0; step >>= 1)
            if ((s_bl[b + step] & ~(1ull << 63)) <= r[k]) b += step;
          if (!(s_bl[b] >> 63)) continue;
        }
        // the rank two words on, where this thread knows it (inside the workgroup's range, not across the wave's edge)
        bool two = false;
        if (p + k + 2 < p1) {
          if (k <= 1) two = r[k + 2] != r[k];
          else if (lane != 63) two = (k == 2 ? rn : rn2) != r[k];
        }
        const unsigned int e = atomicAdd(&s_n, 1u);
        if (e < seg_cap) mine[e] = (unsigned int)(p + k) | (two ? OM_HEAD_L2 : 0u);  // (seg_cap = chunk / 2 + 1 runs of >= 2 rows: cannot overflow)
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) segs[blockIdx.x].count = s_n < seg_cap ? s_n : seg_cap;
}

// =====================================================================================================================================
// Several key columns (round 6).  cudf::sorted_order of a TABLE is a comparison sort under the lexicographic row
// comparator (cpp/src/sort/sort_impl.cuh:61-93, sort.cu:31-50); rounds 1-5 ran it as LSD over the columns -- per extra column one
// single-column argsort, one random 8-byte gather of the column through the order so far and one random 4-byte gather of that order:
// ~90 ms for 2 x int64 at 1e9 rows.  Here the tuple goes through ONE word sort.  Column j has its own bucket plan P_j (the same
// sample / count / plan kernels); the ranks nest from the last column to the first,
//
//   F_k = row / n                                   (a 64-bit fraction)
//   F_j = base_j(b) + A_j(d) + floor(F_{j+1} (B_j(d) - A_j(d)))      -- column j's value owns the ranks [A, B) of P_j's 2^64, and the
//                                                                       REST of the tuple says where inside them the row sits
//   word = F_0 (computed in 64 - ib bits) << ib | row
//
// so F_j is monotone in (c_j, ..., c_{k-1}, row) and rows of equal leading values are ordered -- as far as the bits reach -- by the
// columns behind.  What the bits do not resolve ends in runs of equal ranks, which the run pass puts right by comparing FULL tuples
// (column values fetched on demand, ties by row: a stable order, which both cudf::sorted_order and stable_sorted_order accept).
// One map pass per column (8 B/row fraction in, 8 B/row out, in place): 2 x int64 at 1e9 rows = the single-column cost + ~5 ms.
// NaN: equal to each other and greater than every number in both directions (the comparator rule: row_operator/common_utils.cuh:157-169
// -- NOT the reverse-row-order quirk of the single-column radix path).  Columns: 1-, 2-, 4- or 8-byte numerics without nulls.
constexpr int OMT_MAXCOLS = 8;
struct ColDesc {
  const void* data;
  uint64_t desc_mask;  // 0 or all ones at the column's width
  int kind;            // KeyKind
  int width;           // bytes
};
struct TableDesc {
  ColDesc c[OMT_MAXCOLS];
  int ncols;
};

template <typename U>
__device__ __forceinline__ uint64_t col_sortable_w(const ColDesc& c, int64_t i)
{
  const U b = static_cast<const U*>(c.data)[i];
  const U m = (U)c.desc_mask;
  if (c.kind == K_SIGNED) return (uint64_t)to_sortable<U, K_SIGNED>(b, m);
  return (uint64_t)to_sortable<U, K_UNSIGNED>(b, m);
}
// the column's value in row i as a 64-bit unsigned key (narrow types zero-extended: the order is what matters)
__device__ __forceinline__ uint64_t col_sortable(const ColDesc& c, int64_t i)
{
  switch (c.width) {
    case 8:
      if (c.kind == K_FLOAT) return to_sortable<uint64_t, K_FLOAT>(static_cast<const uint64_t*>(c.data)[i], c.desc_mask);
      return col_sortable_w<uint64_t>(c, i);
    case 4:
      if (c.kind == K_FLOAT) return (uint64_t)to_sortable<uint32_t, K_FLOAT>(static_cast<const uint32_t*>(c.data)[i], (uint32_t)c.desc_mask);
      return col_sortable_w<uint32_t>(c, i);
    case 2: return col_sortable_w<uint16_t>(c, i);
    default: return col_sortable_w<uint8_t>(c, i);
  }
}

__global__ void __launch_bounds__(256) k_omt_sample(ColDesc c, int64_t n, uint64_t* __restrict__ samp)
{
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= OM_S) return;
  const int64_t i = (int64_t)(((__int128)j * n) / OM_S);
  samp[j]         = col_sortable(c, i < n ? i : n - 1);
}

__global__ void __launch_bounds__(256) k_omt_count(ColDesc col, int64_t n, OmPlan* plan)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* s_lo  = reinterpret_cast<uint64_t*>(smem);
  uint32_t* s_eq  = reinterpret_cast<uint32_t*>(smem + OM_B * 8);
  uint32_t* s_cnt = s_eq + OM_B;
  uint16_t* s_lut = reinterpret_cast<uint16_t*>(s_cnt + OM_B);
  for (int i = threadIdx.x; i < OM_B; i += 256) {
    s_lo[i]  = plan->lo[i];
    s_eq[i]  = plan->meta[i];
    s_cnt[i] = 0;
  }
  for (int i = threadIdx.x; i <= OM_LUT; i += 256) s_lut[i] = plan->lut[i];
  __syncthreads();
  const int gsh = plan->gsh;
  const int64_t nchunks = (n + 63) / 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // (a short column: every chunk -- the stride only thins out what is plentiful)
  const int64_t cstride = nchunks >= 64 * OM_CSTRIDE ? OM_CSTRIDE : 1;
  for (int64_t c = ((int64_t)blockIdx.x * 4 + w) * cstride; c < nchunks; c += (int64_t)gridDim.x * 4 * cstride) {
    const int64_t i = c * 64 + lane;
    if (i < n) atomicAdd(&s_cnt[om_bucket(s_lo, s_lut, s_eq, gsh, col_sortable(col, i))], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < OM_B; i += 256)
    if (s_cnt[i]) atomicAdd(&plan->cnt[i], s_cnt[i]);
}

// one column's level of the nested rank: io[i] = rank_j(c_j[i]; F = io[i] or row / n), the leading column's shifted up over the row
__global__ void __launch_bounds__(1024) k_omt_map(ColDesc col, int64_t n, const OmPlan* __restrict__ plan, uint64_t* __restrict__ io, int innermost, int leading)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* s_lo   = reinterpret_cast<uint64_t*>(smem);
  uint64_t* s_base = s_lo + OM_B;
  uint64_t* s_mul  = s_base + OM_B;
  uint32_t* s_meta = reinterpret_cast<uint32_t*>(s_mul + OM_B);
  uint16_t* s_lut  = reinterpret_cast<uint16_t*>(s_meta + OM_B);
  for (int i = threadIdx.x; i < OM_B; i += 1024) {
    s_lo[i]   = plan->lo[i];
    s_base[i] = plan->base[i];
    s_mul[i]  = plan->mul[i];
    s_meta[i] = plan->meta[i];
  }
  for (int i = threadIdx.x; i <= OM_LUT; i += 1024) s_lut[i] = plan->lut[i];
  __syncthreads();
  const int gsh        = plan->gsh;
  const int ib         = plan->ib;  // (0 unless this is the leading column's plan)
  const uint64_t invn  = plan->invn;
  constexpr int U      = 4;
  const int64_t stride = (int64_t)gridDim.x * 1024 * U;
  for (int64_t i0 = (int64_t)blockIdx.x * 1024 * U + threadIdx.x; i0 < n; i0 += stride) {
    uint64_t k[U], f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * 1024;
      k[u]            = i < n ? col_sortable(col, i) : 0ull;
      f[u]            = innermost ? (uint64_t)i * invn : (i < n ? __builtin_nontemporal_load(&io[i]) : 0ull);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * 1024;
      if (i < n) {
        const int b       = om_bucket(s_lo, s_lut, s_meta, gsh, k[u]);
        const int lz      = (int)(s_meta[b] & 63u);
        const uint64_t d  = k[u] - s_lo[b];
        const uint64_t M  = s_mul[b];
        const uint64_t A  = __umul64hi(d << lz, M);
        const uint64_t d1 = (d + 1) << lz;
        const uint64_t B  = d1 ? __umul64hi(d1, M) : M;
        const uint64_t r  = s_base[b] + A + __umul64hi(f[u], B - A);
        __builtin_nontemporal_store(leading ? ((r << ib) | (uint64_t)i) : r, &io[i]);
      }
    }
  }
}

// rows ra, rb agree in the leading column: which comes first?  the columns behind, then the row
__device__ __forceinline__ bool tuple_rest_less(const TableDesc& t, uint32_t ra, uint32_t rb)
{
  for (int c = 1; c < t.ncols; ++c) {
    const uint64_t a = col_sortable(t.c[c], ra), b = col_sortable(t.c[c], rb);
    if (a != b) return a < b;
  }
  return ra < rb;
}
// ---- the run passes, for one key column (OneCol) or a tuple (Tuple).  A policy supplies the leading key of a row in sortable form and
// the order of two rows whose leading keys agree.
template <int KIND>
struct OneCol {
  const uint64_t* keys;
  uint64_t desc_mask;
  static constexpr bool kSkipLossless = true;  // a run inside a lossless bucket holds ONE key value: already in row order
  __device__ __forceinline__ uint64_t key(uint32_t row) const { return to_sortable<uint64_t, KIND>(keys[row], desc_mask); }
  __device__ __forceinline__ bool rest_less(uint32_t ra, uint32_t rb) const { return ra < rb; }
};
struct Tuple {
  TableDesc t;
  static constexpr bool kSkipLossless = false;  // equal ranks say nothing about the columns behind the first
  __device__ __forceinline__ uint64_t key(uint32_t row) const { return col_sortable(t.c[0], row); }
  __device__ __forceinline__ bool rest_less(uint32_t ra, uint32_t rb) const { return tuple_rest_less(t, ra, rb); }
};
template <class P>
__device__ __forceinline__ bool kr_less(const P& pol, uint64_t ka, uint32_t ra, uint64_t kb, uint32_t rb)
{
  return ka < kb || (ka == kb && pol.rest_less(ra, rb));
}

constexpr unsigned int OM_WAVERUN = 128;  // listed runs up to this length belong to one wave (two rows per lane), longer ones to a workgroup

// pass B: one thread per run head that pass A listed
template <class P>
__global__ void __launch_bounds__(256) k_om_finish_b(const uint64_t* __restrict__ sorted, int64_t n, P pol, OmPlan* plan, int32_t* __restrict__ out,
                                                     const unsigned int* __restrict__ heads, const RunSeg* __restrict__ segs, unsigned int seg_cap,
                                                     LongRun* __restrict__ longlist, unsigned int long_cap)
{
  __shared__ uint64_t s_bl[P::kSkipLossless ? OM_B : 1];  // first rank of bucket b | lossy << 63
  const unsigned int cnt = segs[blockIdx.x].count;
  if (cnt == 0) return;
  if (P::kSkipLossless) {
    for (int i = threadIdx.x; i < OM_B; i += 256) s_bl[i] = plan->base[i] | ((plan->meta[i] & 256u) ? (1ull << 63) : 0ull);
    __syncthreads();
  }
  const int ib           = plan->ib;
  const uint64_t imask   = (1ull << ib) - 1;
  const uint64_t bmask   = ~(1ull << 63);
  const unsigned int* mine = heads + (size_t)blockIdx.x * seg_cap;
  // Rounds of 256 heads.  Step 1, a thread per head: runs of 2 - 4 rows are put right on the spot (all loads issued at once, a network in
  // registers), runs of 5 - 16 rows go to the round's list in LDS, longer ones to the global list.  Step 2, SIXTEEN LANES per listed run:
  // a lane per row fetches its key, counts the rows of the run that come before it (16 shuffled compares) and writes its row there.
  // (The first version kept step 2 in the thread -- an insertion sort over private arrays: dependent gathers one after the other and the
  // arrays in scratch memory; ids of a dozen rows each, 8e7 such runs per 1e9 rows, took 99 ms in this kernel.)
  __shared__ unsigned int s_mid[256];
  __shared__ unsigned char s_midl[256];
  __shared__ unsigned int s_nmid, s_nlong, s_lbase;
  __shared__ LongRun s_long[256];
  for (unsigned int base = 0; base < cnt; base += 256) {
    if (threadIdx.x == 0) {
      s_nmid  = 0;
      s_nlong = 0;
    }
    __syncthreads();
    const unsigned int i = base + threadIdx.x;
    const unsigned int hd = i < cnt ? mine[i] : 0u;
    if (i < cnt && (hd & OM_HEAD_L2)) {  // two rows, a lossy bucket (pass A saw both): the rows as pass A wrote them, their keys, one compare
      const unsigned int p = hd & ~OM_HEAD_L2;
      const uint32_t r0 = (uint32_t)out[p], r1 = (uint32_t)out[p + 1];
      const uint64_t k0 = pol.key(r0), k1 = pol.key(r1);
      if (kr_less(pol, k1, r1, k0, r0)) {  // (one column, equal keys: r0 < r1 already)
        out[p]     = (int32_t)r1;
        out[p + 1] = (int32_t)r0;
      }
    } else if (i < cnt) {
      const unsigned int p = hd;
      const uint64_t w0 = sorted[p], w1 = sorted[p + 1];
      const uint64_t r  = w0 >> ib;
      bool skip = false;
      if (P::kSkipLossless) {
        int b = 0;
#pragma unroll
        for (int step = OM_B / 2; step > 0; step >>= 1)
          if ((s_bl[b + step] & bmask) <= r) b += step;
        skip = !(s_bl[b] >> 63);
      }
      if (!skip) {
        int64_t q = (int64_t)p + 2;
        while (q < n && (sorted[q] >> ib) == r) ++q;
        const unsigned int L = (unsigned int)(q - p);
        if (L == 2) {  // the common case
          const uint32_t r0 = (uint32_t)(w0 & imask), r1 = (uint32_t)(w1 & imask);
          const uint64_t k0 = pol.key(r0), k1 = pol.key(r1);
          if (kr_less(pol, k1, r1, k0, r0)) {  // (one column, equal keys: r0 < r1 already)
            out[p]     = (int32_t)r1;
            out[p + 1] = (int32_t)r0;
          }
        } else if (L <= 4u) {
          const bool four = L == 4u;
          uint32_t rr[4] = {(uint32_t)(w0 & imask), (uint32_t)(w1 & imask), (uint32_t)(sorted[p + 2] & imask), four ? (uint32_t)(sorted[p + 3] & imask) : 0u};
          uint64_t kk[4] = {pol.key(rr[0]), pol.key(rr[1]), pol.key(rr[2]), four ? pol.key(rr[3]) : 0ull};
          auto cx = [&](int a, int b) {  // the smaller of (a, b) to a
            if (kr_less(pol, kk[b], rr[b], kk[a], rr[a])) {
              const uint64_t tk = kk[a];
              const uint32_t tr = rr[a];
              kk[a] = kk[b];
              rr[a] = rr[b];
              kk[b] = tk;
              rr[b] = tr;
            }
          };
          cx(0, 1);
          if (four) cx(2, 3);
          cx(0, 2);
          if (four) cx(1, 3);
          cx(1, 2);
          out[p]     = (int32_t)rr[0];
          out[p + 1] = (int32_t)rr[1];
          out[p + 2] = (int32_t)rr[2];
          if (four) out[p + 3] = (int32_t)rr[3];
        } else if (L <= (unsigned)OM_SMALL) {
          const unsigned int e = atomicAdd(&s_nmid, 1u);  // (at most 256 per round)
          s_mid[e]  = p;
          s_midl[e] = (unsigned char)L;
        } else {
          s_long[atomicAdd(&s_nlong, 1u)] = LongRun{p, L};  // (at most 256 per round)
        }
      }
    }
    __syncthreads();
    // the round's long runs: ONE reservation in the global list (a global atomic per run -- 1e7 runs of a hundred equal keys, all on one
    // address at ~88 atomics / us -- was 114 ms of this kernel)
    const unsigned int nlg = s_nlong;
    if (nlg) {
      if (threadIdx.x == 0) s_lbase = atomicAdd(&plan->nlong, nlg);
      __syncthreads();
      const unsigned int e = s_lbase + threadIdx.x;
      if (threadIdx.x < nlg) {
        if (e < long_cap) longlist[e] = s_long[threadIdx.x];
        else plan->long_overflow = 1u;  // cannot happen: long_cap = n / (OM_SMALL + 1) + 1 runs of more than OM_SMALL rows
      }
    }
    const unsigned int nmid = s_nmid;
    const unsigned int j = threadIdx.x & 15u;
    for (unsigned int e = threadIdx.x >> 4; e < nmid; e += 16) {
      const unsigned int p = s_mid[e], L = s_midl[e];
      const bool has     = j < L;
      const uint32_t row = has ? (uint32_t)(sorted[p + j] & imask) : 0u;
      const uint64_t key = has ? pol.key(row) : 0ull;
      unsigned int pos   = 0;
#pragma unroll
      for (int t = 0; t < OM_SMALL; ++t) {
        const uint64_t kt = __shfl(key, t, 16);
        const uint32_t rt = __shfl(row, t, 16);
        if (has && (unsigned)t < L && (unsigned)t != j && kr_less(pol, kt, rt, key, row)) ++pos;
      }
      if (has) out[p + pos] = (int32_t)row;
    }
    __syncthreads();  // (the next round resets the list)
  }
}

// ---- 6a. listed runs up to 128 rows: ONE WAVE looks at each -- two rows per lane, their keys fetched, neighbours compared through
// shuffles.  A column of WIDE keys with DUPLICATES (a hundred rows per 64-bit id: the buckets are lossy, every id is one run of a
// hundred equal keys) lists n / 100 runs that are all in order already; a workgroup and a sorting network for each took 0.5 s per 1e9
// rows.  Runs found in order are done (pass A wrote their rows); the others are ranked inside the wave; runs beyond 128 rows go to the workgroup pass's list.
template <class P>
__global__ void __launch_bounds__(256) k_om_medium(const uint64_t* __restrict__ sorted, P pol, OmPlan* __restrict__ plan, const LongRun* __restrict__ longlist,
                                                   unsigned int long_cap, LongRun* __restrict__ wglist, int32_t* __restrict__ out)
{
  const int ib         = plan->ib;
  const uint64_t imask = (1ull << ib) - 1;
  unsigned int nl      = plan->nlong;
  if (nl > long_cap) nl = long_cap;
  const unsigned int lane = threadIdx.x & 63u;
  const unsigned int wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = gridDim.x * 4u;
  for (unsigned int e = wave; e < nl; e += nwaves) {
    const LongRun run = longlist[e];
    const unsigned int p0 = run.start, L = run.len;
    if (L <= OM_WAVERUN) {  // (wave-uniform; longer runs: the workgroup pass)
      const bool h0 = lane < L, h1 = lane + 64u < L;
      const uint32_t r0 = h0 ? (uint32_t)(sorted[p0 + lane] & imask) : 0u, r1 = h1 ? (uint32_t)(sorted[p0 + 64u + lane] & imask) : 0u;
      const uint64_t k0 = h0 ? pol.key(r0) : 0ull, k1 = h1 ? pol.key(r1) : 0ull;
      uint64_t nk0 = __shfl_down(k0, 1), nk1 = __shfl_down(k1, 1);
      uint32_t nr0 = __shfl_down(r0, 1), nr1 = __shfl_down(r1, 1);
      const uint64_t fk1 = __shfl(k1, 0);
      const uint32_t fr1 = __shfl(r1, 0);
      if (lane == 63u) {
        nk0 = fk1;
        nr0 = fr1;
      }
      bool mine = false;
      if (lane + 1u < L) mine = kr_less(pol, nk0, nr0, k0, r0);                         // element lane + 1 before element lane?
      if (lane < 63u && lane + 65u < L) mine = mine || kr_less(pol, nk1, nr1, k1, r1);  // element lane + 65 before element lane + 64?
      if (ballot(mine) != 0ull) {
        // out of order: every element counts the elements of the run that come before it (L broadcasts) and goes there
        unsigned int pos0 = 0, pos1 = 0;
        for (unsigned int t = 0; t < L; ++t) {
          const unsigned int src = t & 63u;
          const uint64_t ka = __shfl(k0, src), kb = __shfl(k1, src);
          const uint32_t ra = __shfl(r0, src), rb = __shfl(r1, src);
          const uint64_t kt = t < 64u ? ka : kb;
          const uint32_t rt = t < 64u ? ra : rb;
          if (h0 && t != lane && kr_less(pol, kt, rt, k0, r0)) ++pos0;
          if (h1 && t != lane + 64u && kr_less(pol, kt, rt, k1, r1)) ++pos1;
        }
        if (h0) out[p0 + pos0] = (int32_t)r0;
        if (h1) out[p0 + pos1] = (int32_t)r1;
      }
    } else if (lane == 0u) {
      wglist[atomicAdd(&plan->nwg, 1u)] = run;  // (at most n / 129 entries)
    }
  }
}

// ---- 6b. the workgroup list: one workgroup per run; (key, row) sorted by a network whose compare-exchanges all put the
// smaller element at the lower index, so positions beyond the run's length behave as +infinity without being stored
template <class P>
__global__ void __launch_bounds__(1024) k_om_long(const uint64_t* __restrict__ sorted, P pol, const OmPlan* __restrict__ plan, int32_t* __restrict__ out,
                                                  const LongRun* __restrict__ longlist, unsigned int long_cap, uint64_t* __restrict__ gk, uint32_t* __restrict__ gr)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* s_k = reinterpret_cast<uint64_t*>(smem);                    // OM_LDSRUN keys
  uint32_t* s_r = reinterpret_cast<uint32_t*>(smem + OM_LDSRUN * 8);    // OM_LDSRUN rows
  const int ib  = plan->ib;
  const uint64_t imask = (1ull << ib) - 1;
  unsigned int nl = plan->nwg;
  if (nl > long_cap) nl = long_cap;
  for (unsigned int e = blockIdx.x; e < nl; e += gridDim.x) {
    const unsigned int p0 = longlist[e].start, L = longlist[e].len;
    unsigned int N = 1;
    while (N < L) N <<= 1;
    const bool in_lds = L <= (unsigned)OM_LDSRUN;
    uint64_t* K = in_lds ? s_k : gk + p0;  // (global scratch: the run's own slice)
    uint32_t* R = in_lds ? s_r : gr + p0;
    for (unsigned int i = threadIdx.x; i < L; i += 1024) {
      const uint32_t row = (uint32_t)(sorted[p0 + i] & imask);
      K[i] = pol.key(row);
      R[i] = row;
    }
    __syncthreads();
    // in order already (one value's rows, or rows whose keys rise with the row)?  then pass A's output stands
    int bad = 0;
    for (unsigned int i = threadIdx.x; i + 1 < L; i += 1024) bad |= kr_less(pol, K[i + 1], R[i + 1], K[i], R[i]) ? 1 : 0;
    if (__syncthreads_or(bad)) {
      for (unsigned int k = 2; k <= N; k <<= 1) {
        for (unsigned int j = k >> 1; j > 0; j >>= 1) {
          for (unsigned int i = threadIdx.x; i < N; i += 1024) {
            // first stage of a merge: mirror inside the block of k; later stages: distance j.  Always (lower index) <= (higher index).
            const unsigned int prt = (j == (k >> 1)) ? (i ^ (k - 1)) : (i ^ j);
            if (prt > i && prt < L) {  // (i < prt < L; a partner beyond L is +infinity: no exchange)
              const uint64_t ka = K[i], kb = K[prt];
              const uint32_t ra = R[i], rb = R[prt];
              if (kr_less(pol, kb, rb, ka, ra)) {
                K[i]   = kb;
                R[i]   = rb;
                K[prt] = ka;
                R[prt] = ra;
              }
            }
          }
          __syncthreads();
        }
      }
      for (unsigned int i = threadIdx.x; i < L; i += 1024) out[p0 + i] = (int32_t)R[i];
    }
    __syncthreads();
  }
}

static inline size_t round256(size_t b) { return (b + 255) / 256 * 256; }
static thread_local int g_order_map = 1;  // 1: 64-bit sorted_order from 2^25 rows goes through the word sort (default); 0: the round-3 pairs path

struct Layout {
  size_t inner_bytes;
  char* inner;
  OmPlan* plan;  // nplans of them: [0] is the plan of the (leading) key column -- the one gx_sort_order_map_info reads
  uint64_t* samp;
  uint64_t* words;
  uint64_t* sorted;
  size_t total;
};

// the finish passes' geometry: workgroups, rows per workgroup (whole waves of pairs), run-list entries per workgroup
struct FinGeo {
  unsigned int wgs, seg_cap;
  int64_t chunk;
  size_t long_cap;
};
static FinGeo fin_geometry(int64_t n)
{
  FinGeo g;
  const int64_t want = (n + 511) / 512;
  g.wgs      = (unsigned int)(want < OM_FIN_WGS ? (want > 0 ? want : 1) : OM_FIN_WGS);
  g.chunk    = ((n + g.wgs - 1) / g.wgs + 1023) / 1024 * 1024;
  g.seg_cap  = (unsigned int)(g.chunk / 2 + 1);
  g.long_cap = (size_t)n / (OM_SMALL + 1) + 1;
  return g;
}

static int layout(void* tmp, int64_t n, Layout& L, int nplans = 1)
{
  size_t inner = 0;
  int rc       = gx_sort_keys(GX_UINT64, nullptr, nullptr, n, 0, nullptr, &inner, nullptr);
  if (rc) return rc;
  // the runs' scratch lives in the word sort's scratch, which is free by then, behind the plan header: per-workgroup run segments,
  // their counts, the long-run list and its (key, row) slices
  const FinGeo g     = fin_geometry(n);
  const size_t lneed = round256(gx_sort_plan_bytes()) + round256((size_t)g.wgs * g.seg_cap * sizeof(unsigned int)) +
                       round256((size_t)g.wgs * sizeof(RunSeg)) + 2 * round256(g.long_cap * sizeof(LongRun)) + round256((size_t)n * 8) + round256((size_t)n * 4);
  if (lneed > inner) inner = lneed;
  Carver c(tmp);
  L.inner       = c.take<char>(inner);  // FIRST: the word sort's plan header -- its status word -- is what gx_sort_status(tmp) reads
  L.inner_bytes = inner;
  L.plan        = c.take<OmPlan>((size_t)nplans);
  L.samp        = c.take<uint64_t>(OM_S);
  L.words       = c.take<uint64_t>((size_t)n);
  L.sorted      = c.take<uint64_t>((size_t)n);
  L.total       = c.total();
  return 0;
}

// the run scratch inside the (dead) word-sort scratch
struct RunScratch {
  unsigned int* heads;
  RunSeg* segs;
  LongRun* longlist;
  LongRun* wglist;
  uint64_t* gk;
  uint32_t* gr;
};
static RunScratch run_scratch(const Layout& L, const FinGeo& g, int64_t n)
{
  RunScratch r;
  char* lbase = L.inner + round256(gx_sort_plan_bytes());
  r.heads     = reinterpret_cast<unsigned int*>(lbase);
  r.segs      = reinterpret_cast<RunSeg*>(lbase + round256((size_t)g.wgs * g.seg_cap * sizeof(unsigned int)));
  r.longlist  = reinterpret_cast<LongRun*>(reinterpret_cast<char*>(r.segs) + round256((size_t)g.wgs * sizeof(RunSeg)));
  r.wglist    = reinterpret_cast<LongRun*>(reinterpret_cast<char*>(r.longlist) + round256(g.long_cap * sizeof(LongRun)));
  r.gk        = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(r.wglist) + round256(g.long_cap * sizeof(LongRun)));
  r.gr        = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(r.gk) + round256((size_t)n * 8));
  return r;
}

template <int KIND>
static int sorted_order_words(const void* keys, int64_t n, int descending, int32_t* out, void* tmp, size_t* tmp_bytes, hipStream_t s)
{
  Layout L;
  int rc = layout(tmp, n, L);
  if (rc) return rc;
  if (!tmp) {
    *tmp_bytes = L.total;
    return 0;
  }
  if (*tmp_bytes < L.total) return GX_ETMP;
  if (!keys || !out) return GX_EINVAL;
  const uint64_t* k        = static_cast<const uint64_t*>(keys);
  const uint64_t desc_mask = descending ? ~0ull : 0ull;
  int ib                   = 1;
  while (((int64_t)1 << ib) < n) ++ib;
  static std::atomic<bool> attr_set{false};
  static int num_cus = 0;
  if (!attr_set) {
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_om_plan), hipFuncAttributeMaxDynamicSharedMemorySize, OM_S * 8));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_om_map<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, OM_B * 28 + OM_LUT * 2 + 64));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_om_count<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, OM_B * 16 + OM_LUT * 2 + 64));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_om_long<OneCol<KIND>>), hipFuncAttributeMaxDynamicSharedMemorySize, OM_LDSRUN * 12));
    int dev = 0;
    GX_HIP_TRY(hipGetDevice(&dev));
    GX_HIP_TRY(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev));
    attr_set = true;
  }
  const unsigned cus = (unsigned)(num_cus > 0 ? num_cus : 256);
  hipLaunchKernelGGL((k_om_sample<KIND>), dim3(OM_S / 256), dim3(256), 0, s, k, n, desc_mask, L.samp);
  hipLaunchKernelGGL(k_om_plan, dim3(1), dim3(1024), OM_S * 8, s, (const uint64_t*)L.samp, L.plan, ib, n);
  hipLaunchKernelGGL((k_om_count<KIND>), dim3(cus), dim3(256), OM_B * 16 + OM_LUT * 2 + 64, s, k, n, desc_mask, L.plan);
  hipLaunchKernelGGL(k_om_plan2, dim3(1), dim3(1024), 0, s, L.plan);
  hipLaunchKernelGGL((k_om_map<KIND>), dim3(cus), dim3(1024), OM_B * 28 + OM_LUT * 2 + 64, s, k, n, desc_mask, L.plan, L.words);
  size_t ib2 = L.inner_bytes;
  rc         = gx_sort_keys(GX_UINT64, L.words, L.sorted, n, 0, L.inner, &ib2, s);
  if (rc) return rc;
  // (behind the word sort its scratch is dead: the long-run list and slices alias it, past the plan header whose status word stays)
  const FinGeo g     = fin_geometry(n);
  const RunScratch R = run_scratch(L, g, n);
  hipLaunchKernelGGL(k_om_finish_a, dim3(g.wgs), dim3(256), 0, s, (const uint64_t*)L.sorted, n, (const OmPlan*)L.plan, out, R.heads, R.segs, g.chunk, g.seg_cap, 1);
  const OneCol<KIND> pol{k, desc_mask};
  hipLaunchKernelGGL((k_om_finish_b<OneCol<KIND>>), dim3(g.wgs), dim3(256), 0, s, (const uint64_t*)L.sorted, n, pol, L.plan, out, (const unsigned int*)R.heads,
                     (const RunSeg*)R.segs, g.seg_cap, R.longlist, (unsigned int)g.long_cap);
  hipLaunchKernelGGL((k_om_medium<OneCol<KIND>>), dim3(cus * 8), dim3(256), 0, s, (const uint64_t*)L.sorted, pol, L.plan, (const LongRun*)R.longlist, (unsigned int)g.long_cap, R.wglist, out);
  hipLaunchKernelGGL((k_om_long<OneCol<KIND>>), dim3(cus), dim3(1024), OM_LDSRUN * 12, s, (const uint64_t*)L.sorted, pol, (const OmPlan*)L.plan, out,
                     (const LongRun*)R.wglist, (unsigned int)g.long_cap, R.gk, R.gr);
  if (KIND == K_FLOAT && descending) hipLaunchKernelGGL(k_om_reverse_nans, dim3(cus), dim3(256), 0, s, out, (const OmPlan*)L.plan);
  GX_LAUNCH_CHECK();
  return 0;
}

static int col_desc(int dtype, const void* data, int descending, ColDesc& c)
{
  c.data = data;
  switch (dtype) {
    case GX_INT8: c.kind = K_SIGNED; c.width = 1; break;
    case GX_INT16: c.kind = K_SIGNED; c.width = 2; break;
    case GX_INT32: c.kind = K_SIGNED; c.width = 4; break;
    case GX_INT64: c.kind = K_SIGNED; c.width = 8; break;
    case GX_UINT8: case GX_BOOL8: c.kind = K_UNSIGNED; c.width = 1; break;
    case GX_UINT16: c.kind = K_UNSIGNED; c.width = 2; break;
    case GX_UINT32: c.kind = K_UNSIGNED; c.width = 4; break;
    case GX_UINT64: c.kind = K_UNSIGNED; c.width = 8; break;
    case GX_FLOAT32: c.kind = K_FLOAT; c.width = 4; break;
    case GX_FLOAT64: c.kind = K_FLOAT; c.width = 8; break;
    default: return GX_EDTYPE;
  }
  c.desc_mask = descending ? (c.width == 8 ? ~0ull : ((1ull << (8 * c.width)) - 1)) : 0ull;
  return 0;
}

static int sorted_order_table(int ncols, const int* dtypes, const void* const* cols, const int* descending, int64_t n, int32_t* out, void* tmp,
                              size_t* tmp_bytes, hipStream_t s)
{
  Layout L;
  int rc = layout(tmp, n, L, ncols);
  if (rc) return rc;
  if (!tmp) {
    *tmp_bytes = L.total;
    return 0;
  }
  if (*tmp_bytes < L.total) return GX_ETMP;
  if (!out || !dtypes || !cols) return GX_EINVAL;
  TableDesc t;
  t.ncols = ncols;
  for (int c = 0; c < ncols; ++c) {
    if (!cols[c]) return GX_EINVAL;
    rc = col_desc(dtypes[c], cols[c], descending ? descending[c] : 0, t.c[c]);
    if (rc) return rc;
  }
  int ib = 1;
  while (((int64_t)1 << ib) < n) ++ib;
  static std::atomic<bool> attr_set{false};
  static int num_cus = 0;
  if (!attr_set) {
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_om_plan), hipFuncAttributeMaxDynamicSharedMemorySize, OM_S * 8));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_omt_map), hipFuncAttributeMaxDynamicSharedMemorySize, OM_B * 28 + OM_LUT * 2 + 64));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_omt_count), hipFuncAttributeMaxDynamicSharedMemorySize, OM_B * 16 + OM_LUT * 2 + 64));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_om_long<Tuple>), hipFuncAttributeMaxDynamicSharedMemorySize, OM_LDSRUN * 12));
    int dev = 0;
    GX_HIP_TRY(hipGetDevice(&dev));
    GX_HIP_TRY(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev));
    attr_set = true;
  }
  const unsigned cus = (unsigned)(num_cus > 0 ? num_cus : 256);
  for (int c = ncols - 1; c >= 0; --c) {  // from the last column to the leading one: each level reads the fraction the level behind it left
    OmPlan* P = L.plan + c;
    hipLaunchKernelGGL(k_omt_sample, dim3(OM_S / 256), dim3(256), 0, s, t.c[c], n, L.samp);
    hipLaunchKernelGGL(k_om_plan, dim3(1), dim3(1024), OM_S * 8, s, (const uint64_t*)L.samp, P, c == 0 ? ib : 0, n);
    hipLaunchKernelGGL(k_omt_count, dim3(cus), dim3(256), OM_B * 16 + OM_LUT * 2 + 64, s, t.c[c], n, P);
    hipLaunchKernelGGL(k_om_plan2, dim3(1), dim3(1024), 0, s, P);
    hipLaunchKernelGGL(k_omt_map, dim3(cus), dim3(1024), OM_B * 28 + OM_LUT * 2 + 64, s, t.c[c], n, (const OmPlan*)P, L.words, c == ncols - 1 ? 1 : 0, c == 0 ? 1 : 0);
  }
  size_t ib2 = L.inner_bytes;
  rc         = gx_sort_keys(GX_UINT64, L.words, L.sorted, n, 0, L.inner, &ib2, s);
  if (rc) return rc;
  const FinGeo g     = fin_geometry(n);
  const RunScratch R = run_scratch(L, g, n);
  hipLaunchKernelGGL(k_om_finish_a, dim3(g.wgs), dim3(256), 0, s, (const uint64_t*)L.sorted, n, (const OmPlan*)L.plan, out, R.heads, R.segs, g.chunk, g.seg_cap, 0);
  const Tuple pol{t};
  hipLaunchKernelGGL((k_om_finish_b<Tuple>), dim3(g.wgs), dim3(256), 0, s, (const uint64_t*)L.sorted, n, pol, L.plan, out, (const unsigned int*)R.heads,
                     (const RunSeg*)R.segs, g.seg_cap, R.longlist, (unsigned int)g.long_cap);
  hipLaunchKernelGGL((k_om_medium<Tuple>), dim3(cus * 8), dim3(256), 0, s, (const uint64_t*)L.sorted, pol, L.plan, (const LongRun*)R.longlist, (unsigned int)g.long_cap, R.wglist, out);
  hipLaunchKernelGGL((k_om_long<Tuple>), dim3(cus), dim3(1024), OM_LDSRUN * 12, s, (const uint64_t*)L.sorted, pol, (const OmPlan*)L.plan, out, (const LongRun*)R.wglist,
                     (unsigned int)g.long_cap, R.gk, R.gr);
  GX_LAUNCH_CHECK();
  return 0;
}

}  // namespace order
}  // namespace gx

extern "C" {

void gx_sort_set_order_map(int mode) { gx::order::g_order_map = mode ? 1 : 0; }

// whether gx_sorted_order (no nulls) takes this path; the caller (gx_sort.hip) asks first
int gx_order_map_applies(int dtype, int64_t n)
{
  return gx::order::g_order_map && n >= (1ll << 25) && n <= 0x7FFFFFFFll && (dtype == GX_INT64 || dtype == GX_UINT64 || dtype == GX_FLOAT64);
}

int gx_sorted_order_words(int dtype, const void* keys, int64_t n, int descending, int32_t* out, void* tmp, size_t* tmp_bytes, gx_stream_t s)
{
  using namespace gx::order;
  if (n < 0 || !tmp_bytes) return GX_EINVAL;
  hipStream_t st = (hipStream_t)s;
  switch (dtype) {
    case GX_INT64: return sorted_order_words<gx::K_SIGNED>(keys, n, descending, out, tmp, tmp_bytes, st);
    case GX_UINT64: return sorted_order_words<gx::K_UNSIGNED>(keys, n, descending, out, tmp, tmp_bytes, st);
    case GX_FLOAT64: return sorted_order_words<gx::K_FLOAT>(keys, n, descending, out, tmp, tmp_bytes, st);
    default: return GX_EDTYPE;
  }
}

// cudf::sorted_order / stable_sorted_order of a table of 1 <= ncols <= 8 numeric columns without nulls (see "Several key columns" above)
int gx_sorted_order_table(int ncols, const int* dtypes, const void* const* cols, const int* descending, int64_t n, int32_t* out, void* tmp,
                          size_t* tmp_bytes, gx_stream_t s)
{
  if (n < 0 || n > 0x7FFFFFFFll || !tmp_bytes || ncols < 1 || ncols > gx::order::OMT_MAXCOLS) return GX_EINVAL;
  if (n == 0) {
    if (!tmp) *tmp_bytes = 256;
    return 0;
  }
  return gx::order::sorted_order_table(ncols, dtypes, cols, descending, n, out, tmp, tmp_bytes, (hipStream_t)s);
}

// state of the last run that used `tmp`: info[0] = runs that went to the long list, [1] = list overflow (never), [2] = row bits, [3] = rank
// bits, [4] = lossy buckets (rank space < key range: distinct keys may share a rank).  Synchronises.
int gx_sort_order_map_info(const void* tmp, int64_t n, int32_t* info5_host, gx_stream_t s)
{
  using namespace gx::order;
  if (!tmp || !info5_host) return GX_EINVAL;
  Layout L;
  int rc = layout(const_cast<void*>(tmp), n, L);
  if (rc) return rc;
  static thread_local OmPlan h;
  GX_HIP_TRY(hipMemcpyAsync(&h, L.plan, sizeof(OmPlan), hipMemcpyDeviceToHost, (hipStream_t)s));
  GX_HIP_TRY(hipStreamSynchronize((hipStream_t)s));
  info5_host[0] = (int32_t)h.nlong;
  info5_host[1] = (int32_t)h.long_overflow;
  info5_host[2] = h.ib;
  info5_host[3] = h.rb;
  info5_host[4] = h.nlossy;
  return 0;
}

}  // extern "C"
