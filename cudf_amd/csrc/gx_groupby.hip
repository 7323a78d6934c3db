// gx_groupby.hip -- hash groupby SUM / COUNT for gfx950 (single int32/int64 key column).
//
// Replaces cudf's hash groupby core (cpp/src/groupby/hash/compute_groupby.cu:50-155,
// compute_global_memory_aggs.cuh:123-157, single_pass_functors.cuh:85-157): the reference
// inserts ROW INDICES into a set sized by the number of rows and accumulates into N-sized sparse
// columns; here the table is sized by the number of GROUPS (caller's max_groups), slots hold the
// key itself, and the accumulators sit beside the slots (DESIGN.md "groupby").
//
// float SUM: every row does a RETURNING f64 atomic add; knowing the old value lets the thread
// compute the exact rounding error of that addition (two_sum) and add it to a second per-group
// compensation word, so sum + comp is the exact total up to second-order terms and the rounded
// result is within 1 ulp of the correctly rounded sum regardless of the order in which the
// atomics land (the reference's plain relaxed atomic add -- device_atomics.cuh:57-62 -- drifts by
// sqrt(rows per group) ulps).  integer SUM: 64-bit wrapping atomics (exact).
#include <type_traits>

#include "gx_common.hpp"
#include "gx_scan.hpp"

namespace gx {
namespace gb {

constexpr int GBT = 256;

struct GbState {
  unsigned long long special_used;  // rows whose key equals the reserved slot value exist
  unsigned long long overflow;      // table filled up: more distinct keys than max_groups
};

// stored key: key + 1 (mod 2^64) so that 0 can mean EMPTY; the key that maps to 0 (all ones)
// lives in the dedicated slot `capacity`.
template <typename K>
__device__ __forceinline__ unsigned long long stored_key(K k)
{
  return (unsigned long long)k + 1ull;  // K is uint32_t (never wraps to 0) or uint64_t
}

__device__ __forceinline__ uint64_t gb_hash(unsigned long long s, uint32_t log2cap)
{
  return (s * 0x9E3779B97F4A7C15ull) >> (64 - log2cap);
}

template <typename K>
__device__ __forceinline__ int64_t find_or_insert(unsigned long long* table, uint32_t log2cap, K key, GbState* st)
{
  const uint64_t cap = 1ull << log2cap, mask = cap - 1;
  unsigned long long s = stored_key<K>(key);
  if (s == 0ull) {
    st->special_used = 1ull;
    return (int64_t)cap;
  }
  // Once the table has overflowed the result is discarded (the caller retries with a larger table): stop probing.
  // A chain is cut after 2^16 slots: at the load factor the caller sizes for (<= 0.5) no legitimate chain comes
  // near that, and a full table would otherwise cost every new key a walk over all of its slots.
  if (*reinterpret_cast<volatile unsigned long long*>(&st->overflow) != 0ull) return -1;
  const uint64_t limit = cap < 65536ull ? cap : 65536ull;
  uint64_t h = gb_hash(s, log2cap);
  for (uint64_t probes = 0; probes < limit; ++probes) {
    unsigned long long cur = table[h];
    if (cur == 0ull) {
      cur = atomicCAS(&table[h], 0ull, s);
      if (cur == 0ull) return (int64_t)h;
    }
    if (cur == s) return (int64_t)h;
    h = (h + 1) & mask;
  }
  st->overflow = 1ull;
  return -1;
}

template <typename V, bool IS_FLOAT>
struct Acc;
template <typename V>
struct Acc<V, true> {
  static __device__ __forceinline__ void add(double* sum, double* comp, int64_t g, V v)
  {
    const double x   = (double)v;
    const double old = atomicAdd(&sum[g], x);  // returning add: old is what this x was added to
    const double s   = old + x;                // the value the atomic unit stored (RN)
    const double bb  = s - old;
    const double err = (old - (s - bb)) + (x - bb);
    if (err != 0.0) atomicAdd(&comp[g], err);
  }
};
template <typename V>
struct Acc<V, false> {
  static __device__ __forceinline__ void add(double* sum, double*, int64_t g, V v)
  {
    atomicAdd(reinterpret_cast<unsigned long long*>(sum) + g, (unsigned long long)(long long)v);
  }
};

template <typename K, typename V, bool IS_FLOAT>
__global__ void __launch_bounds__(GBT) k_aggregate(const K* __restrict__ keys, const uint32_t* __restrict__ kvalid,
                                                   const V* __restrict__ vals, const uint32_t* __restrict__ vvalid,
                                                   int64_t n, unsigned long long* table, uint32_t log2cap,
                                                   double* sum, double* comp, uint32_t* cnt_valid,
                                                   uint32_t* cnt_all, GbState* st)
{
  const int64_t stride = (int64_t)gridDim.x * GBT;
  for (int64_t i = (int64_t)blockIdx.x * GBT + threadIdx.x; i < n; i += stride) {
    if (kvalid && !bit_is_set(kvalid, i)) continue;  // null_policy::EXCLUDE (compute_groupby.cu:62-66)
    const int64_t g = find_or_insert<K>(table, log2cap, keys[i], st);
    if (g < 0) continue;
    if (cnt_all) atomicAdd(&cnt_all[g], 1u);
    if (vals && (!vvalid || bit_is_set(vvalid, i))) {
      Acc<V, IS_FLOAT>::add(sum, comp, g, vals[i]);
      atomicAdd(&cnt_valid[g], 1u);
    }
  }
}

struct OccLoader {
  const unsigned long long* table;
  uint64_t cap;
  const GbState* st;
  __device__ __forceinline__ uint32_t operator()(int64_t i) const
  {
    if ((uint64_t)i < cap) return table[i] != 0ull ? 1u : 0u;
    return st->special_used ? 1u : 0u;
  }
};

template <typename K, typename V, bool IS_FLOAT>
__global__ void __launch_bounds__(GBT) k_compact(const unsigned long long* __restrict__ table, uint64_t cap,
                                                 const uint32_t* __restrict__ pos, const uint32_t* __restrict__ total,
                                                 const double* __restrict__ sum, const double* __restrict__ comp,
                                                 const uint32_t* __restrict__ cnt_valid,
                                                 const uint32_t* __restrict__ cnt_all, const GbState* st,
                                                 int64_t max_groups, K* out_keys, void* out_sum, int32_t* out_cv,
                                                 int32_t* out_ca, long long* ngroups)
{
  const int64_t stride = (int64_t)gridDim.x * GBT;
  for (int64_t i = (int64_t)blockIdx.x * GBT + threadIdx.x; i <= (int64_t)cap; i += stride) {
    bool occ;
    K key;
    if ((uint64_t)i < cap) {
      const unsigned long long s = table[i];
      occ                        = s != 0ull;
      key                        = (K)(s - 1ull);
    } else {
      occ = st->special_used != 0ull;
      key = (K)(~0ull);
    }
    if (!occ) continue;
    const int64_t p = pos[i];
    if (p >= max_groups) continue;
    out_keys[p] = key;
    if (out_sum) {
      if (IS_FLOAT) {
        const double r = sum[i] + comp[i];
        if (sizeof(V) == 4) static_cast<float*>(out_sum)[p] = (float)r; else static_cast<double*>(out_sum)[p] = r;
      } else {
        static_cast<long long*>(out_sum)[p] = reinterpret_cast<const long long*>(sum)[i];
      }
    }
    if (out_cv) out_cv[p] = (int32_t)cnt_valid[i];
    if (out_ca) out_ca[p] = (int32_t)cnt_all[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const long long g = (long long)*total;
    *ngroups          = st->overflow ? -1ll : g;  // -1: more than max_groups distinct keys
  }
}


// ------------------------------------------------------------------------------------------------
// Partitioned path (large inputs): the chip has 256 x 160 KiB = 40 MiB of LDS, more than the
// accumulators of a million groups, but a row must reach the CU that owns its group.  So:
//   1. k_part_hist     read the keys once (4-8 B/row), histogram of the top 8 hash bits;
//   2. k_part_scatter  one radix-partition pass of (key, value) rows into 256 hash partitions
//                      (tile ranked with LDS atomics, re-ordered in LDS so that every partition
//                      leaves the tile as one contiguous run, global space reserved with one atomic
//                      per (tile, partition)); order inside a partition is irrelevant for groupby;
//   3. k_part_aggregate one 1024-thread workgroup per partition owns an open-addressing table in
//                      LDS ({key, count, sum, compensation} per group) and streams its rows through
//                      LDS atomics (ds_add_rtn_f64 with the same two_sum compensation as above);
//                      at the end the few thousand groups of the partition are merged into the
//                      global table, which also takes any row the LDS table could not hold
//                      (more distinct keys than slots): correctness never depends on the fit.
// HBM traffic: sizeof(K) + 3*(sizeof(K)+sizeof(V)) per row against the algorithmic
// sizeof(K)+sizeof(V), but every access is streaming; the global-atomic path above does ~4 random
// memory-side atomics per row instead.
// ------------------------------------------------------------------------------------------------
constexpr int NPART  = 512;          // MAXIMUM number of partitions (array sizes, launch bounds); 1 << d_gb_pbits are in use
__device__ int d_gb_pbits = 9;       // partition bits in use (8 or 9): see gx_groupby_set_partition_bits (a call's PartPlan may override it)
static int g_gb_pbits     = 9;       // host mirror
constexpr int PBT    = 512;          // scatter workgroup
constexpr int PRPT   = 16;           // rows per thread -> 8192-row tiles
constexpr int PTILE  = PBT * PRPT;
constexpr int ABT    = 1024;         // aggregate workgroup
constexpr int LDS_BUDGET = 160 * 1024 - 2048;

// The scatter pass reserves output space per (row range, partition): the input tiles are split
// into NRANGE contiguous ranges, range r is processed by the workgroups the dispatcher places on
// XCD r (xcd_swizzle; placement is a speed assumption only), and partition p is the concatenation
// of its NRANGE regions.  All writes to one region then come from one XCD, so the partial 128-B
// lines at the ends of neighbouring tiles' runs meet in that XCD's L2 and leave as full lines; with
// a single cursor per partition neighbouring runs come from different XCDs, whose L2s are not
// coherent, and every boundary line is written back twice as a masked partial line.
constexpr int NRANGE = 8;

struct PartPlan {
  unsigned long long count[NRANGE][NPART];   // rows per (range, partition) (null keys excluded)
  unsigned long long cursor[NRANGE][NPART];  // scatter cursors: exact pass = output positions (start at the region offset),
                                             // speculative pass = rows written to the (partition, range) slot so far
  unsigned long long offset[NPART + 1];      // partition starts (exact pass)
  // speculative pass: slot (range, partition) of the partitioned arrays, sized from a SAMPLE of the rows (k_slot_sample /
  // k_slot_plan below): rows of one group all go to one partition, so partition sizes carry the variance of the GROUP sizes
  // -- 1e6 sparse keys over 512 partitions are uneven by +-2.3 %, far beyond the 8 sigma of row-level noise the fixed
  // mean + margin slots of the first version allowed (every sparse-key input overflowed and paid for both passes)
  uint32_t samp[NRANGE][NPART];
  uint32_t slot0[NRANGE][NPART];
  uint32_t cap0[NRANGE][NPART];
  alignas(128) unsigned int overflow;        // speculative pass: a slot outgrew its capacity -> the exact sequence runs
  // Round 4: the partition bits of THIS call (0: the process-wide d_gb_pbits).  k_slot_plan picks 8 for DENSE ids -- the sampled
  // keys span no more than 2 x max_groups values: consecutive integers under the Fibonacci slot hash are a low-discrepancy
  // sequence, an LDS table at 58 % load has next to no collisions, and 256 partitions give the scatter 32-row runs -- and 9
  // otherwise (keys that hash like random numbers need the lower load of 512 tables: 12.7 vs 19.5 ms for sparse int32 keys,
  // while dense keys run 9.53 vs 10.08 ms at 8 vs 9 bits; profiles/r4_run5_bench_groupby_pbits*.jsonl).
  int pbits;
  int auto_pbits;                            // host: let k_slot_plan choose
  unsigned long long kmin, kmax;             // k_slot_sample: smallest / largest sampled key (order-preserving 64-bit form)
  // Round 6: DENSE ids by DIRECT ADDRESS.  Where the sampled keys span few enough values, partition p is the id
  // RANGE [dlo + p dG, dlo + (p + 1) dG) of 256, a row travels as its 8-byte value + the 16-bit remainder of its id inside the range
  // (10 B/row instead of 12: 12 + 10 + 10 = 32 GB per 1e9 rows against 36), and the aggregate kernel's LDS table is indexed by that
  // remainder -- no hashing, no probing, no key compare.  A key outside [dlo, dlo + 256 dG) (the sample missed it by more than the
  // margin) raises `overflow` like a slot that outgrew its capacity: the exact (hash) sequence produces the result.
  int dense_allowed;                         // host: this call may take the dense path (8-byte values, no value nulls, knob on)
  int dense;                                 // k_slot_plan: taken
  unsigned long long dlo;                    // first id of partition 0 (order-preserving 64-bit form)
  unsigned long long dM;                     // floor(2^40 / dG) + 1: (d dM) >> 40 = d / dG for d < 2^21
  uint32_t dG;                               // ids per partition (<= DD_GMAX)
};
constexpr uint32_t DD_GMAX = 7680;           // direct-address entries per workgroup: 20 B each (sum, compensation, count) in 150 KiB of LDS
constexpr int DD_PARTS     = 256;
__device__ __forceinline__ int part_pbits(const PartPlan* plan) { return plan->pbits ? plan->pbits : d_gb_pbits; }

// Round 3: no histogram pass in the common case.  Region (partition p, range r) owns a slot of `cap` rows at
// (p * NRANGE + r) * cap of the partitioned arrays (cap = mean + 8 sigma of a uniform hash); the scatter adds a tile's
// counts to the slot's fill counter and writes behind the earlier tiles of its range; the aggregate kernels walk the
// NRANGE slots of their partition.  Keys skewed enough to overflow a slot raise `overflow`: the speculative aggregate
// then does nothing, and the exact sequence (histogram, offsets, scatter, aggregate), enqueued behind it and otherwise
// a no-op, produces the result -- no host round trip on either branch (the sort's level 1 and the join's partition pass
// work the same way).  Saves the 4 B/row histogram read: 0.7 of 8.3 ms at 1e9 rows, 42 -> 38 GB of HBM traffic.
static inline uint32_t part_cap(int64_t n)
{
  const double mean = (double)n / (double)((1 << g_gb_pbits) * NRANGE);
  const double cap  = mean + 8.0 * __builtin_sqrt(mean + 1.0) + 64.0;
  return (uint32_t)((((int64_t)cap + 31) / 32) * 32);
}

template <typename K>
__device__ __forceinline__ uint64_t part_hash(K key)
{
  return stored_key<K>(key) * 0x9E3779B97F4A7C15ull;
}

// tiles of PTILE rows; range r = tiles [r*per, (r+1)*per), the last range takes the remainder
__host__ __device__ static inline int64_t range_tiles(int64_t n) { return div_up(n, (int64_t)PBT * PRPT) / NRANGE; }

template <typename K>
__global__ void __launch_bounds__(256) k_part_hist(const K* __restrict__ keys, const uint32_t* __restrict__ kvalid,
                                                   int64_t n, PartPlan* plan, int nrange, int gated = 0)
{
  __shared__ uint32_t s_h[NPART];
  if (gated && plan->overflow == 0) return;  // the speculative pass held
  for (int i = threadIdx.x; i < NPART; i += 256) s_h[i] = 0;
  __syncthreads();
  const int psh        = 64 - part_pbits(plan);
  const int r          = blockIdx.x % NRANGE;  // block b -> XCD b % 8 reads the rows it will scatter
  const int64_t jb     = blockIdx.x / NRANGE;
  const int64_t nb     = gridDim.x / NRANGE;
  const int64_t per    = range_tiles(n) * PTILE;
  const int64_t rbegin = (int64_t)r * per;
  const int64_t rend   = (r == NRANGE - 1) ? n : rbegin + per;
  constexpr int U = 8;
  constexpr int V = 16 / (int)sizeof(K);  // keys per 16-byte load
  if (V > 1 && (reinterpret_cast<uintptr_t>(keys) & 15u) == 0) {
    // 16-byte loads: with 4-byte keys the dword form keeps only half as many bytes in flight per lane as the 8-byte
    // key form does, and ran at 3.7 TB/s against 6.1 TB/s (range starts are multiples of PTILE: the vectors stay aligned)
    struct alignas(16) Vec { K k[V]; };
    const int64_t stride = nb * 256 * U * V;
    for (int64_t i0 = rbegin + (jb * 256 * U + threadIdx.x) * V; i0 < rend; i0 += stride) {
      Vec v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + (int64_t)u * 256 * V;
        if (i + V <= rend) {
          v[u] = *reinterpret_cast<const Vec*>(keys + i);
        } else {
#pragma unroll
          for (int e = 0; e < V; ++e) v[u].k[e] = (i + e < rend) ? keys[i + e] : K(0);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + (int64_t)u * 256 * V;
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (i + e < rend && (!kvalid || bit_is_set(kvalid, i + e))) atomicAdd(&s_h[part_hash<K>(v[u].k[e]) >> psh], 1u);
      }
    }
  } else {
    const int64_t stride = nb * 256 * U;
    for (int64_t i0 = rbegin + jb * 256 * U + threadIdx.x; i0 < rend; i0 += stride) {
      K k[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + (int64_t)u * 256;
        k[u]            = (i < rend) ? keys[i] : K(0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + (int64_t)u * 256;
        if (i < rend && (!kvalid || bit_is_set(kvalid, i))) atomicAdd(&s_h[part_hash<K>(k[u]) >> psh], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NPART; i += 256) {
    const uint32_t c = s_h[i];
    if (c) atomicAdd(&plan->count[nrange == 1 ? 0 : r][i], (unsigned long long)c);
  }
}

__global__ void __launch_bounds__(NPART) k_part_offsets(PartPlan* plan, int gated = 0)
{
  __shared__ unsigned long long s_tmp[NPART / GX_WAVE + 1];
  if (gated && plan->overflow == 0) return;
  unsigned long long c = 0;
  for (int r = 0; r < NRANGE; ++r) c += plan->count[r][threadIdx.x];
  unsigned long long total;
  unsigned long long run = block_exclusive_scan<NPART>(c, 0ull, SumOp(), s_tmp, &total);
  plan->offset[threadIdx.x] = run;
  for (int r = 0; r < NRANGE; ++r) {
    plan->cursor[r][threadIdx.x] = run;
    run += plan->count[r][threadIdx.x];
  }
  if (threadIdx.x == 0) plan->offset[NPART] = total;
}

template <typename K, typename V, bool HAS_VV, bool DENSE = false>
__global__ void __launch_bounds__(PBT, 2) k_part_scatter(const K* __restrict__ keys, const uint32_t* __restrict__ kvalid,
                                                      const V* __restrict__ vals, const uint32_t* __restrict__ vvalid,
                                                      int64_t n, PartPlan* plan, K* __restrict__ pkeys,
                                                      V* __restrict__ pvals, uint8_t* __restrict__ pflags, int nrange,
                                                      uint32_t cap = 0, int gated = 0)
{
  // cap > 0: speculative pass into padded slots; cap == 0: exact pass (gated: only after an overflow)
  if (gated && plan->overflow == 0) return;
  // DENSE (speculative pass only): partitions are id ranges, a row's key travels as a 16-bit remainder (PartPlan::dense); the two
  // instantiations are launched behind each other and the one the plan did not choose leaves
  if (cap && (DENSE != (plan->dense != 0))) return;
  typedef typename std::make_unsigned<K>::type UK;
  constexpr unsigned long long SIGN = std::is_signed<K>::value ? (1ull << (8 * sizeof(K) - 1)) : 0ull;
  const unsigned long long dlo  = DENSE ? plan->dlo : 0ull;
  const uint32_t dG             = DENSE ? plan->dG : 1u;
  const unsigned long long dlim = (unsigned long long)DD_PARTS * dG;
  const float dinv              = 1.0f / (float)dG;
  // (DENSE) the id-range partition of a key; `out`: the key lies outside the planned range.  d < 2^21 inside it: the quotient by a
  // float multiply (exact to within one) and a correction, all in 32 bits -- the 64-bit magic multiply of the first version, twice per
  // row, took the kernel from 123 to 140 registers and from two workgroups per CU to one (6.5 ms against 5.6)
  auto dense_part = [&](K k, bool& out) -> uint32_t {
    const unsigned long long d64 = ((unsigned long long)(UK)k ^ SIGN) - dlo;
    out                          = d64 >= dlim;
    const uint32_t d             = (uint32_t)d64;
    uint32_t q                   = (uint32_t)((float)d * dinv);
    const int32_t r              = (int32_t)(d - q * dG);
    if (r < 0) q -= 1u; else if ((uint32_t)r >= dG) q += 1u;
    return q;
  };
  constexpr int ESZ = sizeof(K) > sizeof(V) ? sizeof(K) : sizeof(V);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_buf       = smem;                                             // PTILE * ESZ
  uint8_t* s_flag   = reinterpret_cast<uint8_t*>(smem + (size_t)PTILE * ESZ);  // PTILE (HAS_VV)
  uint32_t* s_cnt   = reinterpret_cast<uint32_t*>(s_flag + (HAS_VV ? PTILE : 0));  // NPART
  uint32_t* s_start = s_cnt + NPART;                                    // NPART
  unsigned long long* s_delta = reinterpret_cast<unsigned long long*>(s_start + NPART);  // NPART
  unsigned long long* s_limit = s_delta + NPART;                        // NPART: end of the partition's slot (speculative pass)
  uint32_t* s_scan  = reinterpret_cast<uint32_t*>(s_limit + NPART);     // 16
  __shared__ uint32_t s_total;

  const unsigned tid = threadIdx.x;
  const int psh      = 64 - part_pbits(plan);
  const int64_t tile = nrange == 1 ? (int64_t)blockIdx.x : xcd_swizzle(blockIdx.x, gridDim.x);
  const int64_t rper = range_tiles(n);
  const int range    = nrange == 1 ? 0 : ((rper > 0 && tile / rper < NRANGE - 1) ? (int)(tile / rper) : NRANGE - 1);
  const int64_t base = tile * PTILE;
  const int nvalid   = (int)((n - base < (int64_t)PTILE) ? (n - base) : (int64_t)PTILE);
  if (tid < NPART) s_cnt[tid] = 0;
  K key[PRPT];
  V val[PRPT];
  uint8_t vflag[PRPT];
  bool live[PRPT];
#pragma unroll
  for (int j = 0; j < PRPT; ++j) {
    const int idx   = j * PBT + (int)tid;
    const int64_t i = base + idx;
    live[j]         = idx < nvalid && (!kvalid || bit_is_set(kvalid, i));
    key[j]          = (idx < nvalid) ? keys[i] : K(0);
    val[j]          = (idx < nvalid) ? vals[i] : V(0);
    vflag[j]        = HAS_VV ? (uint8_t)((idx < nvalid) && bit_is_set(vvalid, i)) : (uint8_t)1;
  }
  __syncthreads();
  uint32_t packed[PRPT];  // partition << 16 | rank inside (tile, partition)
  bool outside = false;
#pragma unroll
  for (int j = 0; j < PRPT; ++j) {
    uint32_t part;
    if (DENSE) {
      bool out;
      part = dense_part(key[j], out);
      if (live[j] && out) {
        outside = true;
        live[j] = false;
      }
      if (!live[j]) part = 0u;
    } else {
      part = (uint32_t)(part_hash<K>(key[j]) >> psh);
    }
    packed[j] = (part << 16) | (live[j] ? atomicAdd(&s_cnt[part], 1u) : 0u);
  }
  if (DENSE && outside) plan->overflow = 1u;  // a key the sample's range (+ margin) does not cover: the exact sequence will run
  __syncthreads();
  const uint32_t c  = (tid < NPART) ? s_cnt[tid] : 0u;
  uint32_t total;
  const uint32_t st = block_exclusive_scan<PBT>(c, 0u, SumOp(), s_scan, &total);
  if (tid < NPART) {
    s_start[tid] = st;
    unsigned long long g = 0;
    if (c) g = atomicAdd(&plan->cursor[range][tid], (unsigned long long)c);
    unsigned long long lim = ~0ull;
    if (cap) {
      const uint32_t scap = plan->cap0[range][tid];
      if (c && g + c > scap) plan->overflow = 1u;  // the surplus is dropped at the write-out; the exact sequence will run
      g += plan->slot0[range][tid];
      lim = (unsigned long long)plan->slot0[range][tid] + scap;
    }
    s_limit[tid] = lim;
    s_delta[tid] = g - st;
  }
  if (tid == 0) s_total = total;
  __syncthreads();
  const int ntot = (int)s_total;
  // ---- keys through LDS first: the partition of the element a thread writes out is recomputed from its key and kept
  // in a register for the value pass (round 2 staged one byte per row in LDS; with 512 partitions that would be two,
  // and the second workgroup per CU would no longer fit)
  K* s_k = reinterpret_cast<K*>(s_buf);
  uint32_t* s_code = reinterpret_cast<uint32_t*>(s_buf);  // DENSE: partition << 16 | remainder (ESZ >= 8 here: PTILE * 4 bytes fit)
#pragma unroll
  for (int j = 0; j < PRPT; ++j) {
    if (live[j]) {
      const uint32_t pos = s_start[packed[j] >> 16] + (packed[j] & 0xFFFFu);
      if (DENSE) s_code[pos] = (packed[j] & 0xFFFF0000u) | (((uint32_t)((unsigned long long)(UK)key[j] ^ SIGN) - (uint32_t)dlo - (packed[j] >> 16) * dG) & 0xFFFFu);  // partition << 16 | remainder
      else s_k[pos] = key[j];
    }
  }
  __syncthreads();
  unsigned short obin[PRPT];
#pragma unroll
  for (int j = 0; j < PRPT; ++j) {
    const int i = j * PBT + (int)tid;
    obin[j]     = 0xFFFFu;
    if (i < ntot) {
      if (DENSE) {
        const uint32_t code          = s_code[i];
        const uint32_t b             = code >> 16;
        const unsigned long long dst = s_delta[b] + (unsigned long long)i;
        if (dst >= s_limit[b]) continue;  // beyond the slot
        reinterpret_cast<unsigned short*>(pkeys)[dst] = (unsigned short)code;
        obin[j]                                       = (unsigned short)b;
      } else {
        const K k                    = s_k[i];
        const uint32_t b             = (uint32_t)(part_hash<K>(k) >> psh);
        const unsigned long long dst = s_delta[b] + (unsigned long long)i;
        if (dst >= s_limit[b]) continue;  // beyond the slot
        pkeys[dst] = k;
        obin[j]    = (unsigned short)b;
      }
    }
  }
  __syncthreads();
  // ---- values (and flags) through the same buffer
  V* s_v = reinterpret_cast<V*>(s_buf);
#pragma unroll
  for (int j = 0; j < PRPT; ++j) {
    if (live[j]) {
      const uint32_t pos = s_start[packed[j] >> 16] + (packed[j] & 0xFFFFu);
      s_v[pos]           = val[j];
      if (HAS_VV) s_flag[pos] = vflag[j];
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PRPT; ++j) {
    const int i = j * PBT + (int)tid;
    if (obin[j] != 0xFFFFu) {
      const unsigned long long dst = s_delta[obin[j]] + (unsigned long long)i;
      pvals[dst]                   = s_v[i];
      if (HAS_VV) pflags[dst] = s_flag[i];
    }
  }
}

template <typename K, bool HAS_VV>
constexpr int lds_slots()
{
  // per slot: key + count_valid + sum + comp (+ count_all)
  constexpr int slot = (int)sizeof(K) + 4 + 8 + 8 + (HAS_VV ? 4 : 0);
  return (LDS_BUDGET / slot) / 256 * 256;
}

template <typename V, bool IS_FLOAT>
struct LdsAcc;
template <typename V>
struct LdsAcc<V, true> {
  static __device__ __forceinline__ void add(double* sum, double* comp, int g, V v)
  {
    const double x   = (double)v;
    const double old = atomicAdd(&sum[g], x);  // ds_add_rtn_f64
    const double s   = old + x;
    const double bb  = s - old;
    const double err = (old - (s - bb)) + (x - bb);
    if (err != 0.0) atomicAdd(&comp[g], err);
  }
};
template <typename V>
struct LdsAcc<V, false> {
  static __device__ __forceinline__ void add(double* sum, double*, int g, V v)
  {
    atomicAdd(reinterpret_cast<unsigned long long*>(sum) + g, (unsigned long long)(long long)v);
  }
};

// merge one partially aggregated group into the global accumulators
template <bool IS_FLOAT>
__device__ __forceinline__ void global_merge(double* sum, double* comp, int64_t g, double psum, double pcomp)
{
  if (IS_FLOAT) {
    const double old = atomicAdd(&sum[g], psum);
    const double s   = old + psum;
    const double bb  = s - old;
    const double err = ((old - (s - bb)) + (psum - bb)) + pcomp;
    if (err != 0.0) atomicAdd(&comp[g], err);
  } else {
    unsigned long long u;
    __builtin_memcpy(&u, &psum, 8);
    atomicAdd(reinterpret_cast<unsigned long long*>(sum) + g, u);
  }
}

template <typename K, typename V, bool IS_FLOAT, bool HAS_VV>
__global__ void __launch_bounds__(ABT) k_part_aggregate(const K* __restrict__ pkeys, const V* __restrict__ pvals,
                                                        const uint8_t* __restrict__ pflags, const PartPlan* plan,
                                                        int nsplit, int nsub, unsigned long long* table, uint32_t log2cap,
                                                        double* sum, double* comp, uint32_t* cnt_valid,
                                                        uint32_t* cnt_all, GbState* st, uint32_t cap = 0, int gated = 0, int nsub8 = 0)
{
  // cap > 0: the speculative pass (skipped when a slot overflowed); cap == 0 && gated: the exact pass behind it
  if (cap ? (plan->overflow != 0 || plan->dense != 0) : (gated && plan->overflow == 0)) return;  // (dense: k_dense_aggregate has the speculative pass)
  // the grid is sized for the process-wide partition bits; a call whose plan chose 8 (dense ids) uses nsub8 workgroups per
  // partition and the surplus workgroups leave
  const int pb = part_pbits(plan);
  if (plan->pbits == 8 && nsub8 > 0) nsub = nsub8;
  if (blockIdx.x >= (unsigned)((1 << pb) * nsplit * nsub)) return;
  constexpr int S     = lds_slots<K, HAS_VV>();
  constexpr K EMPTYK  = K(~K(0));   // rows with this key use the dedicated slot S
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* l_sum    = reinterpret_cast<double*>(smem);            // S + 1
  double* l_comp   = l_sum + (S + 1);                             // S + 1
  K* l_key         = reinterpret_cast<K*>(l_comp + (S + 1));      // S + 1 (+1 pad)
  uint32_t* l_cv   = reinterpret_cast<uint32_t*>(l_key + (S + 2));  // S + 1
  uint32_t* l_ca   = l_cv + (S + 1);                              // S + 1 (HAS_VV)
  __shared__ uint32_t s_nkeys;
  __shared__ uint32_t s_special;

  const unsigned tid = threadIdx.x;
  for (int i = tid; i <= S; i += ABT) {
    l_sum[i]  = 0.0;
    l_comp[i] = 0.0;
    l_key[i]  = EMPTYK;
    l_cv[i]   = 0;
    if (HAS_VV) l_ca[i] = 0;
  }
  if (tid == 0) {
    s_nkeys   = 0;
    s_special = 0;
  }
  __syncthreads();

  // nsub > 1: the keys of a partition are dealt to nsub workgroups by hash bits the partition and the LDS slot do
  // not use; each reads the whole slice and keeps its own keys, so its LDS table sees 1/nsub of the groups
  const int sub   = (int)(blockIdx.x % (unsigned)nsub);
  const int part  = (int)(blockIdx.x / (unsigned)nsub) / nsplit;
  const int split = (int)(blockIdx.x / (unsigned)nsub) % nsplit;
  // the rows of this partition: the NRANGE padded slots of the speculative pass, or the exact partition
  const int nreg = cap ? NRANGE : 1;
  constexpr uint32_t MAXKEYS = (uint32_t)(S - S / 8);  // stop inserting new keys above 87.5 % load

  constexpr int U = 8;
  for (int rg = 0; rg < nreg; ++rg) {
  unsigned long long p0, p1;
  if (cap) {
    const unsigned long long fill = plan->cursor[rg][part];
    const unsigned long long scap = plan->cap0[rg][part];
    p0 = plan->slot0[rg][part];
    p1 = p0 + (fill < scap ? fill : scap);
  } else {
    p0 = plan->offset[part];
    p1 = plan->offset[part + 1];
  }
  const unsigned long long len = p1 - p0;
  const unsigned long long per = (len + nsplit - 1) / nsplit;
  const unsigned long long r0  = p0 + per * split < p1 ? p0 + per * split : p1;
  const unsigned long long r1  = r0 + per < p1 ? r0 + per : p1;
  auto process = [&](const K key, const V val, const uint8_t fl) {
      if (nsub > 1 && (int)((part_hash<K>(key) >> 16) & (uint64_t)(nsub - 1)) != sub) return;
      int slot    = -1;
      if (key == EMPTYK) {
        slot      = S;
        s_special = 1u;  // benign race: every writer stores 1
      } else {
        uint32_t h = (uint32_t)(((part_hash<K>(key) >> (32 - pb)) & 0xFFFFFFFFull) * (uint64_t)S >> 32);
        for (int probes = 0; probes < S; ++probes) {
          K cur = l_key[h];
          if (cur == EMPTYK) {
            if (s_nkeys >= MAXKEYS) break;  // table (nearly) full: this row goes to the global table
            cur = atomicCAS(&l_key[h], EMPTYK, key);
            if (cur == EMPTYK) {
              atomicAdd(&s_nkeys, 1u);
              slot = (int)h;
              break;
            }
          }
          if (cur == key) {
            slot = (int)h;
            break;
          }
          h = (h + 1 == (uint32_t)S) ? 0u : h + 1;
        }
      }
      if (slot >= 0) {
        // (workgroup scope: an ordinary atomicAdd here is merged with the one on the global counters of the other branch into ONE
        //  flat_atomic_add through a selected pointer -- every row then pays a FLAT access, and a pending FLAT access makes the compiler
        //  wait with vmcnt(0) lgkmcnt(0) for everything)
        if (HAS_VV) __hip_atomic_fetch_add(&l_ca[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (fl) {
          LdsAcc<V, IS_FLOAT>::add(l_sum, l_comp, slot, val);
          __hip_atomic_fetch_add(&l_cv[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else {
        const int64_t g = find_or_insert<K>(table, log2cap, key, st);
        if (g >= 0) {
          if (cnt_all) atomicAdd(&cnt_all[g], 1u);
          if (fl) {
            Acc<V, IS_FLOAT>::add(sum, comp, g, val);
            atomicAdd(&cnt_valid[g], 1u);
          }
        }
      }
  };
  constexpr bool VEC = sizeof(K) == 4 && sizeof(V) == 8 && !HAS_VV;
  if (VEC && cap && nsplit == 1) {
    // round 6: four consecutive rows per lane and access (one 16-byte load of keys, two of values; the speculative slots start at
    // multiples of 16 rows) -- the form that took the dense path's aggregate from 3.1 to 1.9 ms
    typedef K k4 __attribute__((ext_vector_type(4)));
    typedef V v2 __attribute__((ext_vector_type(2)));
    constexpr int Q = U / 4;
    for (unsigned long long i0 = r0 + 4ull * tid; i0 < r1; i0 += 4ull * ABT * Q) {
      k4 kq[Q];
      v2 va[Q], vb[Q];
#pragma unroll
      for (int u = 0; u < Q; ++u) {
        const unsigned long long i = i0 + 4ull * ABT * u;
        if (i + 3 < r1) {
          kq[u] = *reinterpret_cast<const k4*>(pkeys + i);
          va[u] = *reinterpret_cast<const v2*>(pvals + i);
          vb[u] = *reinterpret_cast<const v2*>(pvals + i + 2);
        } else {
          kq[u] = k4{i < r1 ? pkeys[i] : K(0), i + 1 < r1 ? pkeys[i + 1] : K(0), i + 2 < r1 ? pkeys[i + 2] : K(0), K(0)};
          va[u] = v2{i < r1 ? pvals[i] : V(0), i + 1 < r1 ? pvals[i + 1] : V(0)};
          vb[u] = v2{i + 2 < r1 ? pvals[i + 2] : V(0), V(0)};
        }
      }
#pragma unroll
      for (int u = 0; u < Q; ++u) {
        const unsigned long long i = i0 + 4ull * ABT * u;
        const K kk[4] = {kq[u].x, kq[u].y, kq[u].z, kq[u].w};
        const V vv[4] = {va[u].x, va[u].y, vb[u].x, vb[u].y};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (i + e < r1) process(kk[e], vv[e], (uint8_t)1);
      }
    }
  } else {
  for (unsigned long long i0 = r0 + tid; i0 < r1; i0 += (unsigned long long)ABT * U) {
    K k[U];
    V v[U];
    uint8_t f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long i = i0 + (unsigned long long)u * ABT;
      const bool in              = i < r1;
      k[u]                       = in ? pkeys[i] : K(0);
      v[u]                       = in ? pvals[i] : V(0);
      f[u]                       = HAS_VV ? (in ? pflags[i] : (uint8_t)0) : (uint8_t)1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long i = i0 + (unsigned long long)u * ABT;
      if (i >= r1) continue;
      process(k[u], v[u], f[u]);
    }
  }
  }
  }  // regions
  __syncthreads();
  // ---- merge this workgroup's groups into the global table
  for (int i = tid; i <= S; i += ABT) {
    const K key = l_key[i];
    const bool occ = (i < S) ? (key != EMPTYK) : (s_special != 0u);
    if (!occ) continue;
    const int64_t g = find_or_insert<K>(table, log2cap, (i < S) ? key : EMPTYK, st);
    if (g < 0) continue;
    const uint32_t cv = l_cv[i];
    if (cnt_all) atomicAdd(&cnt_all[g], HAS_VV ? l_ca[i] : cv);
    if (cv) {
      atomicAdd(&cnt_valid[g], cv);
      global_merge<IS_FLOAT>(sum, comp, g, l_sum[i], l_comp[i]);
    }
  }
}

// Round 6: the aggregate of the DENSE path (PartPlan::dense).  One workgroup per id range; the LDS table is indexed by the row's 16-bit
// remainder -- {sum, compensation, count} per id, no key, no probing; a row costs a 2-byte and an 8-byte load, one ds_add_rtn_f64 and
// one ds_add_u32.  The groups leave through the same global table as the hash path's (find_or_insert), key = dlo + p dG + remainder.
template <typename K, typename V, bool IS_FLOAT>
__global__ void __launch_bounds__(ABT) k_dense_aggregate(const unsigned short* __restrict__ prem, const V* __restrict__ pvals, const PartPlan* plan,
                                                         unsigned long long* table, uint32_t log2cap, double* sum, double* comp, uint32_t* cnt_valid,
                                                         uint32_t* cnt_all, GbState* st)
{
  if (plan->overflow != 0 || plan->dense == 0) return;
  const uint32_t G = plan->dG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* l_sum  = reinterpret_cast<double*>(smem);          // G
  double* l_comp = l_sum + DD_GMAX;                           // G
  uint32_t* l_cv = reinterpret_cast<uint32_t*>(l_comp + DD_GMAX);  // G
  const unsigned tid = threadIdx.x;
  for (uint32_t i = tid; i < G; i += ABT) {
    l_sum[i]  = 0.0;
    l_comp[i] = 0.0;
    l_cv[i]   = 0;
  }
  __syncthreads();
  const int part  = (int)blockIdx.x;
  constexpr int U = 8;
  for (int rg = 0; rg < NRANGE; ++rg) {
    const unsigned long long fill = plan->cursor[rg][part];
    const unsigned long long scap = plan->cap0[rg][part];
    const unsigned long long p0   = plan->slot0[rg][part];
    const unsigned long long p1   = p0 + (fill < scap ? fill : scap);
    // four consecutive rows per lane and access: an 8-byte load of remainders, two 16-byte loads of values (slots start at multiples of
    // 16 rows); 2-byte loads -- 128 B per wave instruction -- left the pass at 3.2 TB/s
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    typedef V v2 __attribute__((ext_vector_type(2)));
    constexpr int Q = U / 4;
    for (unsigned long long i0 = p0 + 4ull * tid; i0 < p1; i0 += 4ull * ABT * Q) {
      u16x4 r[Q];
      v2 va[Q], vb[Q];
#pragma unroll
      for (int u = 0; u < Q; ++u) {
        const unsigned long long i = i0 + 4ull * ABT * u;
        if (i + 3 < p1) {
          r[u]  = *reinterpret_cast<const u16x4*>(prem + i);
          va[u] = *reinterpret_cast<const v2*>(pvals + i);
          vb[u] = *reinterpret_cast<const v2*>(pvals + i + 2);
        } else {  // the slot's last rows
          r[u]  = u16x4{i < p1 ? prem[i] : (unsigned short)0, i + 1 < p1 ? prem[i + 1] : (unsigned short)0, i + 2 < p1 ? prem[i + 2] : (unsigned short)0, (unsigned short)0};
          va[u] = v2{i < p1 ? pvals[i] : V(0), i + 1 < p1 ? pvals[i + 1] : V(0)};
          vb[u] = v2{i + 2 < p1 ? pvals[i + 2] : V(0), V(0)};
        }
      }
#pragma unroll
      for (int u = 0; u < Q; ++u) {
        const unsigned long long i = i0 + 4ull * ABT * u;
        const unsigned short rr[4] = {r[u].x, r[u].y, r[u].z, r[u].w};
        const V vv[4]              = {va[u].x, va[u].y, vb[u].x, vb[u].y};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (i + e >= p1) continue;
          LdsAcc<V, IS_FLOAT>::add(l_sum, l_comp, (int)rr[e], vv[e]);
          atomicAdd(&l_cv[rr[e]], 1u);
        }
      }
    }
  }
  __syncthreads();
  typedef typename std::make_unsigned<K>::type UK;
  constexpr unsigned long long SIGN = std::is_signed<K>::value ? (1ull << (8 * sizeof(K) - 1)) : 0ull;
  const unsigned long long first    = plan->dlo + (unsigned long long)part * G;
  for (uint32_t i = tid; i < G; i += ABT) {
    const uint32_t cv = l_cv[i];
    if (!cv) continue;
    const K key     = (K)(UK)((first + i) ^ SIGN);
    const int64_t g = find_or_insert<K>(table, log2cap, key, st);
    if (g < 0) continue;
    if (cnt_all) atomicAdd(&cnt_all[g], cv);
    atomicAdd(&cnt_valid[g], cv);
    global_merge<IS_FLOAT>(sum, comp, g, l_sum[i], l_comp[i]);
  }
}

static inline uint32_t log2_cap(int64_t max_groups)
{
  uint32_t lg = 6;
  while ((1ull << lg) < (unsigned long long)(max_groups < 1 ? 1 : max_groups) * 2ull) ++lg;
  return lg;
}

// Workgroups per (partition, split) of the LDS aggregation: enough that the groups one LDS table has to hold stay
// under ~65 % of its slots.  Linear probing with keys that hash like random numbers (row hashes, sparse ids) grows
// long chains beyond that -- measured at 1e6 groups / 1e9 rows: 3.4 ms for dense integer keys (Fibonacci hashing
// spreads consecutive integers almost perfectly), 24.5 ms for uniformly random 64-bit keys at 69 % load
// (profiles/r2_xp_minmax_matrix.txt) -- and beyond 7/8 the rows spill to the global table one by one.
static inline int lds_nsub(int64_t max_groups, int lds_slots)
{
  const double per_part = (double)(max_groups < 1 ? 1 : max_groups) / (double)(1 << g_gb_pbits);
  int nsub = 1;
  while (nsub < 16 && per_part / nsub > 0.65 * lds_slots) nsub *= 2;
  return nsub;
}

static thread_local int g_gb_auto = 1;       // 1 = a call's plan may choose 8 partition bits for dense ids (PartPlan::pbits); gx_groupby_set_partition_bits(8 | 9) switches it off
static thread_local int g_gb_spec = 1;       // A/B knob: 1 = speculative hist-free partition pass (default), 0 = always the exact pass, 2 = speculative for every n
static thread_local int g_gb_algorithm = 0;  // 0 auto, 1 global-atomic table only, 2 partitioned whenever possible
static thread_local int g_gb_nsplit    = 1;
static thread_local int g_gb_nrange    = NRANGE;  // 1 = single cursor per partition (A/B measurement)
static thread_local int g_gb_dense     = 1;       // round 6: dense ids by direct address (PartPlan::dense); gx_groupby_set_dense(0) keeps the hash path
constexpr int64_t PART_MIN_ROWS = 1 << 19;
// the speculative pass pays off once the slots are long enough for the 8-sigma margin to be small (>= ~2000 rows per slot)
static inline bool part_speculative(int64_t n) { return g_gb_nrange == NRANGE && (g_gb_spec == 2 || (g_gb_spec == 1 && n >= (1 << 22))); }
// elements of the partitioned arrays: the padded slots of the speculative pass, or n
static inline int slot_stride(int64_t n)  // the sample takes every stride-th 64-row chunk
{
  int s = 1;
  while (s < 32 && (n >> 22) >= 2 * s) s *= 2;
  return s;
}
// rows the partitioned arrays must hold: n + the slack k_slot_plan hands out (Cauchy-Schwarz over the slots)
static inline size_t slot_elems(int64_t n, int stride)
{
  const double slots = (double)NRANGE * (double)(1 << g_gb_pbits);
  const double dev   = 8.0 * 1.25 * stride * __builtin_sqrt(slots * ((double)n / stride + 2.0 * slots));
  return (size_t)((double)n * 1.002 + dev) + (size_t)slots * (size_t)(2 * stride * GX_WAVE + 64 + 16) + 65536;
}
static inline size_t part_elems(int64_t n) { return part_speculative(n) ? slot_elems(n, slot_stride(n)) : (size_t)n; }

template <typename K>
__global__ void __launch_bounds__(256) k_slot_sample(const K* __restrict__ keys, const uint32_t* __restrict__ kvalid, int64_t n, PartPlan* plan,
                                                     int stride, int64_t range_rows)
{
  __shared__ uint32_t s_hist[NRANGE * NPART];
  const unsigned tid = threadIdx.x, lane = lane_id();
  for (int i = tid; i < NRANGE * NPART; i += 256) s_hist[i] = 0;
  __syncthreads();
  const int psh         = 64 - d_gb_pbits;  // (the sample is always taken on the process-wide bits; k_slot_plan may fold it to 8)
  const int64_t step    = (int64_t)stride * GX_WAVE;
  const int64_t nchunks = div_up(n, step);
  const int64_t nw      = (int64_t)gridDim.x * 4;
  unsigned long long kmin = ~0ull, kmax = 0ull;  // order-preserving form: sign bit flipped for signed keys
  constexpr unsigned long long SIGN = std::is_signed<K>::value ? (1ull << (8 * sizeof(K) - 1)) : 0ull;
  for (int64_t c = (int64_t)blockIdx.x * 4 + tid / GX_WAVE; c < nchunks; c += nw) {
    const int64_t row = c * step + lane;
    const bool live   = row < n && (!kvalid || bit_is_set(kvalid, row));
    const K k         = keys[row < n ? row : 0];
    const int64_t r64 = range_rows > 0 ? row / range_rows : (int64_t)(NRANGE - 1);
    const int r       = r64 < NRANGE - 1 ? (int)r64 : NRANGE - 1;
    (void)lds_rank(s_hist + r * NPART, (uint32_t)(part_hash<K>(k) >> psh), live);
    if (live) {
      typedef typename std::make_unsigned<K>::type UK;
      const unsigned long long u = (unsigned long long)(UK)k ^ SIGN;
      kmin = u < kmin ? u : kmin;
      kmax = u > kmax ? u : kmax;
    }
  }
  if (plan->auto_pbits) {
    kmin = wave_reduce(kmin, MinOp());
    kmax = wave_reduce(kmax, MaxOp());
    if (lane == 0 && kmin <= kmax) {
      atomicMin(&plan->kmin, kmin);
      atomicMax(&plan->kmax, kmax);
    }
  }
  __syncthreads();
  for (int i = tid; i < NRANGE * NPART; i += 256) {
    const uint32_t c = s_hist[i];
    if (c) atomicAdd(&plan->samp[i / NPART][i % NPART], c);
  }
}

// DENSE path: the sample again, counted per (range, id-range partition) -- the slot capacities of the dense scatter
template <typename K>
__global__ void __launch_bounds__(256) k_dense_sample(const K* __restrict__ keys, const uint32_t* __restrict__ kvalid, int64_t n, PartPlan* plan, int stride,
                                                      int64_t range_rows)
{
  if (plan->dense == 0) return;
  __shared__ uint32_t s_hist[NRANGE * DD_PARTS];
  const unsigned tid = threadIdx.x, lane = lane_id();
  for (int i = tid; i < NRANGE * DD_PARTS; i += 256) s_hist[i] = 0;
  __syncthreads();
  typedef typename std::make_unsigned<K>::type UK;
  constexpr unsigned long long SIGN = std::is_signed<K>::value ? (1ull << (8 * sizeof(K) - 1)) : 0ull;
  const unsigned long long dlo = plan->dlo, dM = plan->dM;
  const int64_t step    = (int64_t)stride * GX_WAVE;
  const int64_t nchunks = div_up(n, step);
  const int64_t nw      = (int64_t)gridDim.x * 4;
  for (int64_t c = (int64_t)blockIdx.x * 4 + tid / GX_WAVE; c < nchunks; c += nw) {
    const int64_t row = c * step + lane;
    const bool live   = row < n && (!kvalid || bit_is_set(kvalid, row));
    const K k         = keys[row < n ? row : 0];
    const int64_t r64 = range_rows > 0 ? row / range_rows : (int64_t)(NRANGE - 1);
    const int r       = r64 < NRANGE - 1 ? (int)r64 : NRANGE - 1;
    const unsigned long long d = ((unsigned long long)(UK)k ^ SIGN) - dlo;  // (sampled rows lie inside the planned range)
    uint32_t part     = (uint32_t)((d * dM) >> 40);
    if (part >= (uint32_t)DD_PARTS) part = DD_PARTS - 1;
    if (live) atomicAdd(&s_hist[r * DD_PARTS + part], 1u);
  }
  __syncthreads();
  for (int i = tid; i < NRANGE * DD_PARTS; i += 256) {
    const uint32_t c = s_hist[i];
    if (c) atomicAdd(&plan->samp[i / DD_PARTS][i % DD_PARTS], c);
  }
}

// one block of NPART threads: capacities = estimate + 8 sigma of the estimate + two sample steps + a constant, slots laid
// out partition-major (the NRANGE slots of a partition are neighbours)
template <typename Plan>
__global__ void __launch_bounds__(NPART) k_slot_plan(Plan* plan, int64_t n, int stride, int64_t range_rows, unsigned long long elems,
                                                     long long max_groups = 0, int dense_phase = 0)
{
  __shared__ uint32_t s_tmp[NPART / GX_WAVE + 1];
  const int t = threadIdx.x;
  if constexpr (std::is_same<Plan, PartPlan>::value) {
    // dense_phase 1: the second call, behind k_dense_sample -- capacities from the id-range counts (only if the first call chose the dense path)
    if (dense_phase == 1 && plan->dense == 0) return;
    if (dense_phase == 0 && plan->dense_allowed && plan->kmin <= plan->kmax) {
      // ids by direct address (PartPlan::dense): the sampled range pushed out by 1/64 of its width + 64 at both ends (the sample sees
      // one row in `stride`: the true extremes of evenly used ids lie a few ids outside it), cut into 256 ranges of dG ids
      const unsigned long long w      = plan->kmax - plan->kmin;
      const unsigned long long margin = w / 64 + 64;
      const unsigned long long lo     = plan->kmin > margin ? plan->kmin - margin : 0ull;
      const unsigned long long hi     = plan->kmax > ~0ull - margin ? ~0ull : plan->kmax + margin;
      const unsigned long long span   = hi - lo;  // (+ 1 ids)
      const bool fits = span / DD_PARTS < (unsigned long long)DD_GMAX && max_groups > 0 && 4 * max_groups <= (long long)n;
      __syncthreads();
      if (fits) {
        for (int r = 0; r < NRANGE; ++r) plan->samp[r][t] = 0;  // counted again per id range (k_dense_sample)
        if (t == 0) {
          const uint32_t G = (uint32_t)(span / DD_PARTS) + 1u;
          plan->dense      = 1;
          plan->dlo        = lo;
          plan->dG         = G;
          plan->dM         = (1ull << 40) / G + 1ull;
          plan->pbits      = 8;  // (the exact sequence, should it run, cuts 256 hash partitions)
        }
        return;
      }
    }
    // dense ids (the sampled keys span at most 2 x max_groups values, from 0): 256 partitions -- the sample's 512 bins fold pairwise
    // (partition = top bits of the hash) -- otherwise the process-wide bits.  See PartPlan::pbits.
    if (dense_phase == 0 && plan->auto_pbits && d_gb_pbits == 9) {
      // (max_groups must be a real bound -- a quarter of the rows at most: a caller that passes n says nothing about the key range)
      const bool dense = max_groups > 0 && 4 * max_groups <= (long long)n && plan->kmax < 2ull * (unsigned long long)max_groups;
      if (dense) {
        uint32_t a[NRANGE], b[NRANGE];
        for (int r = 0; r < NRANGE; ++r) {
          a[r] = t < NPART / 2 ? plan->samp[r][2 * t] : 0u;
          b[r] = t < NPART / 2 ? plan->samp[r][2 * t + 1] : 0u;
        }
        __syncthreads();
        for (int r = 0; r < NRANGE; ++r) plan->samp[r][t] = a[r] + b[r];
        __syncthreads();
      }
      if (t == 0) plan->pbits = dense ? 8 : 9;
    }
  }
  uint32_t cap[NRANGE];
  uint32_t sum = 0;
  for (int r = 0; r < NRANGE; ++r) {
    const uint32_t c = plan->samp[r][t];
    uint32_t total;
    (void)block_exclusive_scan<NPART>(c, 0u, SumOp(), s_tmp, &total);
    const int64_t b    = (int64_t)r * range_rows < n ? (int64_t)r * range_rows : n;
    const int64_t e    = (r == NRANGE - 1) ? n : (b + range_rows < n ? b + range_rows : n);
    const double rows  = (double)(e - b);
    const double scale = total ? rows / (double)total : 0.0;
    double cp          = (double)c * scale + 8.0 * scale * __builtin_sqrt((double)c + 1.0) + 2.0 * stride * GX_WAVE + 64.0;
    if (cp > rows) cp = rows;
    cap[r] = ((uint32_t)cp + 15u) & ~15u;
    sum += cap[r];
  }
  uint32_t total;
  uint32_t run = block_exclusive_scan<NPART>(sum, 0u, SumOp(), s_tmp, &total);
  if ((unsigned long long)total > elems) {
    if (t == 0) plan->overflow = 1u;
    return;
  }
  for (int r = 0; r < NRANGE; ++r) {
    plan->slot0[r][t] = run;
    plan->cap0[r][t]  = cap[r];
    run += cap[r];
  }
}
// between the speculative and the exact pass: the fill counters become position cursors again (the exact k_part_offsets sets them)
__global__ void __launch_bounds__(NPART) k_part_reset_cursors(PartPlan* plan)
{
  if (plan->overflow == 0) return;
  for (int r = 0; r < NRANGE; ++r) plan->cursor[r][threadIdx.x] = 0;
}

template <typename K, typename V, bool IS_FLOAT, bool HAS_VV>
int launch_partitioned(const K* keys, const uint32_t* kvalid, const V* vals, const uint32_t* vvalid, int64_t n,
                       PartPlan* plan, K* pkeys, V* pvals, uint8_t* pflags, unsigned long long* table, uint32_t lg,
                       double* sum, double* comp, uint32_t* cv, uint32_t* ca, GbState* st, int64_t max_groups, hipStream_t s)
{
  GX_HIP_TRY(hipMemsetAsync(plan, 0, sizeof(PartPlan), s));
  int64_t hb = div_up(n, 256 * 8 * 4 * NRANGE);
  if (hb > 256) hb = 256;
  if (hb < 1) hb = 1;
  constexpr int ESZ      = sizeof(K) > sizeof(V) ? sizeof(K) : sizeof(V);
  constexpr size_t lds_s = (size_t)PTILE * ESZ + (HAS_VV ? PTILE : 0) + NPART * 4 * 2 + NPART * 8 * 2 + 64;
  auto ks                = k_part_scatter<K, V, HAS_VV>;
  constexpr int S        = lds_slots<K, HAS_VV>();
  constexpr size_t lds_a = (size_t)(S + 1) * 16 + (size_t)(S + 2) * sizeof(K) + (size_t)(S + 1) * 4 * (HAS_VV ? 2 : 1);
  auto ka                = k_part_aggregate<K, V, IS_FLOAT, HAS_VV>;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ka), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
    attr_set = true;
  }
  const int nsplit    = g_gb_nsplit;
  const int nsub      = lds_nsub(max_groups, S);
  // the plan of a call may choose 8 bits (dense ids): workgroups per partition for that case; the grid covers both
  int nsub8 = 1;
  while (nsub8 < 16 && (double)(max_groups < 1 ? 1 : max_groups) / 256.0 / nsub8 > 0.65 * S) nsub8 *= 2;
  const unsigned sgrd = (unsigned)div_up(n, PTILE);
  unsigned agrd       = (unsigned)((1 << g_gb_pbits) * nsplit * nsub);
  if (g_gb_auto && g_gb_pbits == 9 && (unsigned)(256 * nsplit * nsub8) > agrd) agrd = (unsigned)(256 * nsplit * nsub8);
  const bool spec     = part_speculative(n);
  int gated           = 0;
  if (spec) {  // speculative pass: no histogram, padded slots (see PartPlan)
    const uint32_t cap       = 1u;  // != 0: the speculative form (slot tables in the plan)
    const int stride         = slot_stride(n);
    const int64_t range_rows = range_tiles(n) * PTILE;
    int64_t sblocks          = div_up(div_up(n, (int64_t)stride * GX_WAVE), (int64_t)4 * 8);
    if (sblocks > 2048) sblocks = 2048;
    if (g_gb_auto && g_gb_pbits == 9) {  // let k_slot_plan choose the partition bits of this call (PartPlan::pbits)
      static const int one = 1;
      static const unsigned long long all_ones = ~0ull;
      GX_HIP_TRY(hipMemcpyAsync(&plan->auto_pbits, &one, sizeof(int), hipMemcpyHostToDevice, s));
      GX_HIP_TRY(hipMemcpyAsync(&plan->kmin, &all_ones, sizeof(all_ones), hipMemcpyHostToDevice, s));  // (the plan was cleared: a minimum starts at the top)
    }
    // round 6: dense ids by direct address (PartPlan::dense) -- 8-byte values without nulls, keys of 4 or 8 bytes, knob on
    constexpr bool dense_types = !HAS_VV && sizeof(V) == 8 && sizeof(K) >= 4 && std::is_integral<K>::value;
    const bool dense_ok        = dense_types && g_gb_dense && g_gb_auto && g_gb_pbits == 9 && nsplit == 1;
    if (dense_ok) {
      static const int one = 1;
      GX_HIP_TRY(hipMemcpyAsync(&plan->dense_allowed, &one, sizeof(int), hipMemcpyHostToDevice, s));
    }
    hipLaunchKernelGGL((k_slot_sample<K>), dim3((unsigned)sblocks), dim3(256), 0, s, keys, kvalid, n, plan, stride, range_rows);
    hipLaunchKernelGGL((k_slot_plan<PartPlan>), dim3(1), dim3(NPART), 0, s, plan, n, stride, range_rows, (unsigned long long)slot_elems(n, stride),
                       (long long)max_groups, 0);
    if constexpr (dense_types) {
      if (dense_ok) {  // every kernel of the path the plan did not choose leaves at once
        constexpr size_t lds_d = (size_t)DD_GMAX * 20;
        auto ksd               = k_part_scatter<K, V, false, true>;
        auto kad               = k_dense_aggregate<K, V, IS_FLOAT>;
        static std::atomic<bool> dattr{false};
        if (!dattr) {
          GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ksd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
          GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kad), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_d));
          dattr = true;
        }
        hipLaunchKernelGGL((k_dense_sample<K>), dim3((unsigned)sblocks), dim3(256), 0, s, keys, kvalid, n, plan, stride, range_rows);
        hipLaunchKernelGGL((k_slot_plan<PartPlan>), dim3(1), dim3(NPART), 0, s, plan, n, stride, range_rows, (unsigned long long)slot_elems(n, stride),
                           (long long)max_groups, 1);
        hipLaunchKernelGGL(ksd, dim3(sgrd), dim3(PBT), lds_s, s, keys, kvalid, vals, vvalid, n, plan, pkeys, pvals, pflags, NRANGE, cap, 0);
        hipLaunchKernelGGL(kad, dim3(DD_PARTS), dim3(ABT), lds_d, s, reinterpret_cast<const unsigned short*>(pkeys), (const V*)pvals, (const PartPlan*)plan, table, lg,
                           sum, comp, cv, ca, st);
      }
    }
    hipLaunchKernelGGL(ks, dim3(sgrd), dim3(PBT), lds_s, s, keys, kvalid, vals, vvalid, n, plan, pkeys, pvals, pflags, NRANGE, cap, 0);
    hipLaunchKernelGGL(ka, dim3(agrd), dim3(ABT), lds_a, s, pkeys, pvals, pflags, plan, nsplit, nsub, table, lg, sum, comp, cv, ca, st, cap,
                       0, nsub8);
    hipLaunchKernelGGL(k_part_reset_cursors, dim3(1), dim3(NPART), 0, s, plan);
    gated = 1;  // the exact sequence below runs only if a slot overflowed
  }
  hipLaunchKernelGGL((k_part_hist<K>), dim3((unsigned)(hb * NRANGE)), dim3(256), 0, s, keys, kvalid, n, plan, g_gb_nrange, gated);
  hipLaunchKernelGGL(k_part_offsets, dim3(1), dim3(NPART), 0, s, plan, gated);
  hipLaunchKernelGGL(ks, dim3(sgrd), dim3(PBT), lds_s, s, keys, kvalid, vals, vvalid, n, plan, pkeys, pvals, pflags, g_gb_nrange, 0u, gated);
  hipLaunchKernelGGL(ka, dim3(agrd), dim3(ABT), lds_a, s, pkeys, pvals, pflags, plan, nsplit, nsub, table, lg, sum, comp, cv, ca, st, 0u,
                     gated, nsub8);
  GX_LAUNCH_CHECK();
  return 0;
}

template <typename K, typename V, bool IS_FLOAT>
int groupby_impl(const void* keys, const uint32_t* kvalid, const void* vals, const uint32_t* vvalid, int64_t n,
                 int64_t max_groups, void* out_keys, void* out_sum, int32_t* out_cv, int32_t* out_ca,
                 int64_t* ngroups, void* tmp, size_t* tmp_bytes, hipStream_t s)
{
  const uint32_t lg  = log2_cap(max_groups);
  const uint64_t cap = 1ull << lg;
  // the partitioned path needs a values column and pays off only on large inputs
  const bool partitioned = vals != nullptr && g_gb_algorithm != 1 && n > 0 &&
                           (g_gb_algorithm == 2 || n >= PART_MIN_ROWS);
  Carver c(tmp);
  GbState* st               = c.take<GbState>(1);
  unsigned long long* table = c.take<unsigned long long>(cap);
  double* sum               = c.take<double>(cap + 1);
  double* comp              = c.take<double>(cap + 1);
  uint32_t* cv              = c.take<uint32_t>(cap + 1);
  uint32_t* ca              = c.take<uint32_t>(cap + 1);
  uint32_t* pos             = c.take<uint32_t>(cap + 1);
  uint32_t* partials        = c.take<uint32_t>(scan::partials_count(cap + 1));
  // the size query cannot see `vals` (callers pass the same arguments, so it can): keep both layouts equal
  PartPlan* plan  = c.take<PartPlan>(1);
  K* pkeys        = partitioned ? c.take<K>(part_elems(n)) : nullptr;
  V* pvals        = partitioned ? c.take<V>(part_elems(n)) : nullptr;
  uint8_t* pflags = (partitioned && vvalid) ? c.take<uint8_t>(part_elems(n)) : nullptr;
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  // one memset covers state, table and accumulators (they are contiguous up to `pos`)
  GX_HIP_TRY(hipMemsetAsync(tmp, 0, (size_t)(reinterpret_cast<char*>(pos) - static_cast<char*>(tmp)), s));
  if (partitioned) {
    int rc;
    if (vvalid)
      rc = launch_partitioned<K, V, IS_FLOAT, true>(static_cast<const K*>(keys), kvalid, static_cast<const V*>(vals),
                                                    vvalid, n, plan, pkeys, pvals, pflags, table, lg, sum, comp, cv,
                                                    out_ca ? ca : nullptr, st, max_groups, s);
    else
      rc = launch_partitioned<K, V, IS_FLOAT, false>(static_cast<const K*>(keys), kvalid, static_cast<const V*>(vals),
                                                     vvalid, n, plan, pkeys, pvals, pflags, table, lg, sum, comp, cv,
                                                     out_ca ? ca : nullptr, st, max_groups, s);
    if (rc) return rc;
  } else if (n > 0) {
    int64_t blocks = div_up(n, GBT * 8);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL((k_aggregate<K, V, IS_FLOAT>), dim3((unsigned)blocks), dim3(GBT), 0, s,
                       static_cast<const K*>(keys), kvalid, static_cast<const V*>(vals), vvalid, n, table, lg, sum,
                       comp, cv, out_ca ? ca : nullptr, st);
  }
  OccLoader ld{table, cap, st};
  int rc = scan::device_scan<uint32_t, uint32_t>(ld, (int64_t)cap + 1, 0u, SumOp(), false, pos, partials, s);
  if (rc) return rc;
  int64_t blocks = div_up((int64_t)cap + 1, GBT * 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((k_compact<K, V, IS_FLOAT>), dim3((unsigned)blocks), dim3(GBT), 0, s, table, cap, pos,
                     partials + scan::num_chunks((int64_t)cap + 1), sum, comp, cv, ca, st, max_groups,
                     static_cast<K*>(out_keys), out_sum, out_cv, out_ca, reinterpret_cast<long long*>(ngroups));
  GX_LAUNCH_CHECK();
  return 0;
}

template <typename K>
int dispatch_val(int val_dtype, const void* keys, const uint32_t* kvalid, const void* vals, const uint32_t* vvalid,
                 int64_t n, int64_t max_groups, void* out_keys, void* out_sum, int32_t* out_cv, int32_t* out_ca,
                 int64_t* ngroups, void* tmp, size_t* tmp_bytes, hipStream_t s)
{
#define GX_GB(V, F) \
  return groupby_impl<K, V, F>(keys, kvalid, vals, vvalid, n, max_groups, out_keys, out_sum, out_cv, out_ca, ngroups, tmp, tmp_bytes, s)
  switch (val_dtype) {
    case GX_INT8: GX_GB(int8_t, false);
    case GX_INT16: GX_GB(int16_t, false);
    case GX_INT32: GX_GB(int32_t, false);
    case GX_INT64: GX_GB(int64_t, false);
    case GX_BOOL8:
    case GX_UINT8: GX_GB(uint8_t, false);
    case GX_UINT16: GX_GB(uint16_t, false);
    case GX_UINT32: GX_GB(uint32_t, false);
    case GX_UINT64: GX_GB(uint64_t, false);
    case GX_FLOAT32: GX_GB(float, true);
    case GX_FLOAT64: GX_GB(double, true);
    default: return GX_EDTYPE;
  }
#undef GX_GB
}

}  // namespace gb
}  // namespace gx

extern "C" {

int gx_groupby_sum_count(int key_dtype, const void* keys, const uint32_t* keys_valid, int val_dtype,
                         const void* vals, const uint32_t* vals_valid, int64_t n, int64_t max_groups,
                         void* out_keys, void* out_sum, int32_t* out_count_valid, int32_t* out_count_all,
                         int64_t* ngroups_dev, void* tmp, size_t* tmp_bytes, gx_stream_t s)
{
  if (n < 0 || max_groups < 0 || !tmp_bytes) return GX_EINVAL;
  if (tmp && (!ngroups_dev || (n > 0 && !keys) || (max_groups > 0 && !out_keys))) return GX_EINVAL;
  switch (key_dtype) {
    case GX_INT32:
    case GX_UINT32:
      return gx::gb::dispatch_val<uint32_t>(val_dtype, keys, keys_valid, vals, vals_valid, n, max_groups, out_keys,
                                            out_sum, out_count_valid, out_count_all, ngroups_dev, tmp, tmp_bytes, s);
    case GX_INT64:
    case GX_UINT64:
      return gx::gb::dispatch_val<uint64_t>(val_dtype, keys, keys_valid, vals, vals_valid, n, max_groups, out_keys,
                                            out_sum, out_count_valid, out_count_all, ngroups_dev, tmp, tmp_bytes, s);
    default: return GX_EDTYPE;
  }
}

void gx_groupby_set_partition_mode(int speculative) { gx::gb::g_gb_spec = speculative == 2 ? 2 : (speculative ? 1 : 0); }

int gx_groupby_set_partition_bits(int bits)
{
  // 0: 512 partitions process-wide, and the sum / count path picks 256 per call for dense ids (default); 8 / 9: fixed
  gx::gb::g_gb_auto = bits == 0 ? 1 : 0;
  const int v = bits == 8 ? 8 : 9;
  GX_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gx::gb::d_gb_pbits), &v, sizeof(int)));
  gx::gb::g_gb_pbits = v;
  return 0;
}

void gx_groupby_set_dense(int on) { gx::gb::g_gb_dense = on ? 1 : 0; }

// which path the last gx_groupby_sum_count that used `tmp` (with this max_groups) took: info[0] = dense ids by direct address,
// [1] = a slot overflowed / a key lay outside the planned id range (the exact sequence produced the result), [2] = partition bits of
// the call's plan (0: process-wide), [3] = ids per partition of the dense path.  Synchronises.
int gx_groupby_plan_info(const void* tmp, int64_t max_groups, int32_t* info4_host, gx_stream_t s)
{
  using namespace gx;
  using namespace gx::gb;
  if (!tmp || !info4_host) return GX_EINVAL;
  const uint64_t cap = 1ull << log2_cap(max_groups);
  Carver c(const_cast<void*>(tmp));
  (void)c.take<GbState>(1);
  (void)c.take<unsigned long long>(cap);
  (void)c.take<double>(cap + 1);
  (void)c.take<double>(cap + 1);
  (void)c.take<uint32_t>(cap + 1);
  (void)c.take<uint32_t>(cap + 1);
  (void)c.take<uint32_t>(cap + 1);
  (void)c.take<uint32_t>(scan::partials_count(cap + 1));
  const PartPlan* plan = c.take<PartPlan>(1);
  static thread_local PartPlan h;
  GX_HIP_TRY(hipMemcpyAsync(&h, plan, sizeof(PartPlan), hipMemcpyDeviceToHost, (hipStream_t)s));
  GX_HIP_TRY(hipStreamSynchronize((hipStream_t)s));
  info4_host[0] = h.dense;
  info4_host[1] = (int32_t)h.overflow;
  info4_host[2] = h.pbits;
  info4_host[3] = (int32_t)h.dG;
  return 0;
}

void gx_groupby_set_algorithm(int algo, int nsplit)
{
  gx::gb::g_gb_nrange    = (algo & 16) ? 1 : gx::gb::NRANGE;  // +16: single-cursor scatter (A/B measurement)
  algo &= 15;
  gx::gb::g_gb_algorithm = (algo >= 0 && algo <= 2) ? algo : 0;
  gx::gb::g_gb_nsplit    = (nsplit >= 1 && nsplit <= 16) ? nsplit : 1;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// result finalizers used by the host layer (cudf::groupby::aggregate): validity of SUM/MEAN from
// COUNT_VALID (cpp/src/groupby/hash/output_utils.cu:68-70: a group with no valid value is null) and
// MEAN = SUM / COUNT_VALID in double (hash_compound_agg_finalizer.cu:92-133).
// ------------------------------------------------------------------------------------------------
namespace gx {
namespace gb {

__global__ void __launch_bounds__(256) k_valid_from_counts(const int32_t* __restrict__ counts, int64_t n,
                                                           uint32_t* mask, unsigned long long* nulls)
{
  const int64_t nwords = (n + 31) / 32;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long local = 0;
  for (int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < nwords; wi += stride) {
    uint32_t bits = 0;
    for (int b = 0; b < 32; ++b) {
      const int64_t i = wi * 32 + b;
      if (i < n) {
        if (counts[i] > 0) bits |= 1u << b; else ++local;
      }
    }
    mask[wi] = bits;
  }
  local = wave_reduce(local, SumOp());
  if (lane_id() == 0 && local) atomicAdd(nulls, local);
}

template <typename S>
__global__ void __launch_bounds__(256) k_mean(const S* __restrict__ sum, const int32_t* __restrict__ cnt, int64_t n,
                                              double* out)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = cnt[i] > 0 ? (double)sum[i] / (double)cnt[i] : 0.0;
}

}  // namespace gb
}  // namespace gx

extern "C" {

int gx_valid_from_counts(const int32_t* counts, int64_t n, uint32_t* mask_out, int64_t* null_count_dev, gx_stream_t s)
{
  if (n < 0 || (n > 0 && (!counts || !mask_out)) || !null_count_dev) return GX_EINVAL;
  GX_HIP_TRY(hipMemsetAsync(null_count_dev, 0, sizeof(int64_t), s));
  if (n == 0) return 0;
  int64_t blocks = gx::div_up(gx::div_up(n, 32), 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gx::gb::k_valid_from_counts, dim3((unsigned)blocks), dim3(256), 0, s, counts, n, mask_out,
                     reinterpret_cast<unsigned long long*>(null_count_dev));
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_mean_from_sum(int sum_dtype, const void* sum, const int32_t* count, int64_t n, double* out, gx_stream_t s)
{
  if (n < 0 || (n > 0 && (!sum || !count || !out))) return GX_EINVAL;
  if (n == 0) return 0;
  int64_t blocks = gx::div_up(n, 256 * 4);
  if (blocks > 4096) blocks = 4096;
  switch (sum_dtype) {
    case GX_INT64: hipLaunchKernelGGL((gx::gb::k_mean<int64_t>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const int64_t*>(sum), count, n, out); break;
    case GX_FLOAT64: hipLaunchKernelGGL((gx::gb::k_mean<double>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const double*>(sum), count, n, out); break;
    case GX_FLOAT32: hipLaunchKernelGGL((gx::gb::k_mean<float>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const float*>(sum), count, n, out); break;
    default: return GX_EDTYPE;
  }
  GX_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// groupby MIN / MAX (cpp/src/groupby/hash/global_memory_aggregator.cuh:18-238: atomic min/max per
// group).  Values are widened to a 64-bit word whose unsigned order is the value order (sign flip
// for signed integers, the IEEE total-order flip for floats with -0.0 -> +0.0 and every NaN above
// +Inf, the row comparator's order: include/cudf/detail/row_operator/common_utils.cuh:157-169), so one
// native 64-bit unsigned atomic min / max per row serves every value type.  Small inputs use the global table
// directly, large ones the LDS-partitioned path below (k_part_minmax).
// ------------------------------------------------------------------------------------------------
namespace gx {
namespace gb {

template <typename V>
__device__ __forceinline__ unsigned long long mm_encode(V v)
{
  if constexpr (sizeof(V) == 8 && !std::is_integral<V>::value) {
    unsigned long long b;
    __builtin_memcpy(&b, &v, 8);
    return to_sortable<unsigned long long, K_FLOAT>(b, 0ull);
  } else if constexpr (!std::is_integral<V>::value) {
    const double d = (double)v;  // float -> double is exact and order preserving
    unsigned long long b;
    __builtin_memcpy(&b, &d, 8);
    return to_sortable<unsigned long long, K_FLOAT>(b, 0ull);
  } else if constexpr (std::is_signed<V>::value) {
    return (unsigned long long)(long long)v ^ 0x8000000000000000ull;
  } else {
    return (unsigned long long)v;
  }
}
template <typename V>
__device__ __forceinline__ V mm_decode(unsigned long long s)
{
  if constexpr (!std::is_integral<V>::value) {
    const unsigned long long b = (s & 0x8000000000000000ull) ? (s ^ 0x8000000000000000ull) : ~s;
    double d;
    __builtin_memcpy(&d, &b, 8);
    return (V)d;
  } else if constexpr (std::is_signed<V>::value) {
    return (V)(long long)(s ^ 0x8000000000000000ull);
  } else {
    return (V)s;
  }
}

template <typename K, typename V>
__global__ void __launch_bounds__(GBT) k_minmax(const K* __restrict__ keys, const uint32_t* __restrict__ kvalid,
                                                const V* __restrict__ vals, const uint32_t* __restrict__ vvalid, int64_t n,
                                                unsigned long long* table, uint32_t log2cap, unsigned long long* mn,
                                                unsigned long long* mx, uint32_t* cnt_valid, GbState* st)
{
  const int64_t stride = (int64_t)gridDim.x * GBT;
  for (int64_t i = (int64_t)blockIdx.x * GBT + threadIdx.x; i < n; i += stride) {
    if (kvalid && !bit_is_set(kvalid, i)) continue;
    const int64_t g = find_or_insert<K>(table, log2cap, keys[i], st);
    if (g < 0) continue;
    if (!vvalid || bit_is_set(vvalid, i)) {
      const unsigned long long e = mm_encode<V>(vals[i]);
      atomicMin(&mn[g], e);
      atomicMax(&mx[g], e);
      atomicAdd(&cnt_valid[g], 1u);
    }
  }
}

template <typename K, typename V>
__global__ void __launch_bounds__(GBT) k_minmax_compact(const unsigned long long* __restrict__ table, uint64_t cap,
                                                        const uint32_t* __restrict__ pos, const uint32_t* __restrict__ total,
                                                        const unsigned long long* __restrict__ mn,
                                                        const unsigned long long* __restrict__ mx,
                                                        const uint32_t* __restrict__ cnt_valid, const GbState* st,
                                                        int64_t max_groups, K* out_keys, V* out_min, V* out_max,
                                                        int32_t* out_cv, long long* ngroups)
{
  const int64_t stride = (int64_t)gridDim.x * GBT;
  for (int64_t i = (int64_t)blockIdx.x * GBT + threadIdx.x; i <= (int64_t)cap; i += stride) {
    bool occ;
    K key;
    if ((uint64_t)i < cap) {
      const unsigned long long s = table[i];
      occ                        = s != 0ull;
      key                        = (K)(s - 1ull);
    } else {
      occ = st->special_used != 0ull;
      key = (K)(~0ull);
    }
    if (!occ) continue;
    const int64_t p = pos[i];
    if (p >= max_groups) continue;
    out_keys[p] = key;
    const bool has = cnt_valid[i] > 0;
    if (out_min) out_min[p] = has ? mm_decode<V>(mn[i]) : V(0);
    if (out_max) out_max[p] = has ? mm_decode<V>(mx[i]) : V(0);
    if (out_cv) out_cv[p] = (int32_t)cnt_valid[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *ngroups = st->overflow ? -1ll : (long long)*total;
}

// LDS-partitioned MIN / MAX: the same radix partition of (key, value) rows as the SUM path (k_part_hist /
// k_part_scatter), then one workgroup per partition folding its rows into an LDS table of {key, min, max, count}
// with ds_min_u64 / ds_max_u64 on the order-preserving 64-bit encoding; a group reaches the global table once per
// partition (one find_or_insert + three global atomics) instead of once per row.  Rows that do not fit the LDS
// table (more than 7/8 of its slots in use) take the global path row by row, as in k_part_aggregate.
template <typename K, bool HAS_VV>
constexpr int lds_slots_mm()
{
  constexpr int slot = (int)sizeof(K) + 8 + 8 + 4;
  return (LDS_BUDGET / slot) / 256 * 256;
}

template <typename K, typename V, bool HAS_VV>
__global__ void __launch_bounds__(ABT) k_part_minmax(const K* __restrict__ pkeys, const V* __restrict__ pvals,
                                                     const uint8_t* __restrict__ pflags, const PartPlan* plan, int nsplit, int nsub,
                                                     unsigned long long* table, uint32_t log2cap, unsigned long long* mn,
                                                     unsigned long long* mx, uint32_t* cnt_valid, GbState* st, uint32_t cap = 0,
                                                     int gated = 0)
{
  if (cap ? plan->overflow != 0 : (gated && plan->overflow == 0)) return;
  constexpr int S    = lds_slots_mm<K, HAS_VV>();
  constexpr K EMPTYK = K(~K(0));  // rows with this key use the dedicated slot S
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* l_mn = reinterpret_cast<unsigned long long*>(smem);  // S + 1
  unsigned long long* l_mx = l_mn + (S + 1);                                // S + 1
  K* l_key                 = reinterpret_cast<K*>(l_mx + (S + 1));          // S + 1 (+1 pad)
  uint32_t* l_cv           = reinterpret_cast<uint32_t*>(l_key + (S + 2));  // S + 1
  __shared__ uint32_t s_nkeys;
  __shared__ uint32_t s_special;

  const unsigned tid = threadIdx.x;
  for (int i = tid; i <= S; i += ABT) {
    l_mn[i]  = ~0ull;
    l_mx[i]  = 0ull;
    l_key[i] = EMPTYK;
    l_cv[i]  = 0;
  }
  if (tid == 0) {
    s_nkeys   = 0;
    s_special = 0;
  }
  __syncthreads();

  // nsub > 1: the keys of a partition are dealt to nsub workgroups by hash bits the partition and the LDS slot do
  // not use; each reads the whole slice and keeps its own keys, so its LDS table sees 1/nsub of the groups
  const int sub   = (int)(blockIdx.x % (unsigned)nsub);
  const int part  = (int)(blockIdx.x / (unsigned)nsub) / nsplit;
  const int split = (int)(blockIdx.x / (unsigned)nsub) % nsplit;
  // the rows of this partition: the NRANGE padded slots of the speculative pass, or the exact partition
  const int nreg = cap ? NRANGE : 1;
  constexpr uint32_t MAXKEYS   = (uint32_t)(S - S / 8);

  constexpr int U = 8;
  for (int rg = 0; rg < nreg; ++rg) {
  unsigned long long p0, p1;
  if (cap) {
    const unsigned long long fill = plan->cursor[rg][part];
    const unsigned long long scap = plan->cap0[rg][part];
    p0 = plan->slot0[rg][part];
    p1 = p0 + (fill < scap ? fill : scap);
  } else {
    p0 = plan->offset[part];
    p1 = plan->offset[part + 1];
  }
  const unsigned long long len = p1 - p0;
  const unsigned long long per = (len + nsplit - 1) / nsplit;
  const unsigned long long r0  = p0 + per * split < p1 ? p0 + per * split : p1;
  const unsigned long long r1  = r0 + per < p1 ? r0 + per : p1;
  for (unsigned long long i0 = r0 + tid; i0 < r1; i0 += (unsigned long long)ABT * U) {
    K k[U];
    V v[U];
    uint8_t f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long i = i0 + (unsigned long long)u * ABT;
      const bool in              = i < r1;
      k[u]                       = in ? pkeys[i] : K(0);
      v[u]                       = in ? pvals[i] : V(0);
      f[u]                       = HAS_VV ? (in ? pflags[i] : (uint8_t)0) : (uint8_t)1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long i = i0 + (unsigned long long)u * ABT;
      if (i >= r1) continue;
      const K key = k[u];
      if (nsub > 1 && (int)((part_hash<K>(key) >> 16) & (uint64_t)(nsub - 1)) != sub) continue;
      int slot    = -1;
      if (key == EMPTYK) {
        slot      = S;
        s_special = 1u;  // benign race: every writer stores 1
      } else {
        uint32_t h = (uint32_t)(((part_hash<K>(key) >> (32 - d_gb_pbits)) & 0xFFFFFFFFull) * (uint64_t)S >> 32);
        for (int probes = 0; probes < S; ++probes) {
          K cur = l_key[h];
          if (cur == EMPTYK) {
            if (s_nkeys >= MAXKEYS) break;
            cur = atomicCAS(&l_key[h], EMPTYK, key);
            if (cur == EMPTYK) {
              atomicAdd(&s_nkeys, 1u);
              slot = (int)h;
              break;
            }
          }
          if (cur == key) {
            slot = (int)h;
            break;
          }
          h = (h + 1 == (uint32_t)S) ? 0u : h + 1;
        }
      }
      const unsigned long long e = mm_encode<V>(v[u]);
      if (slot >= 0) {
        if (f[u]) {
          atomicMin(&l_mn[slot], e);
          atomicMax(&l_mx[slot], e);
          atomicAdd(&l_cv[slot], 1u);
        }
      } else {
        const int64_t g = find_or_insert<K>(table, log2cap, key, st);
        if (g >= 0 && f[u]) {
          atomicMin(&mn[g], e);
          atomicMax(&mx[g], e);
          atomicAdd(&cnt_valid[g], 1u);
        }
      }
    }
  }
  }  // regions
  __syncthreads();
  // ---- merge this workgroup's groups into the global table (a group with only null values still gets its slot)
  for (int i = tid; i <= S; i += ABT) {
    const K key    = l_key[i];
    const bool occ = (i < S) ? (key != EMPTYK) : (s_special != 0u);
    if (!occ) continue;
    const int64_t g = find_or_insert<K>(table, log2cap, (i < S) ? key : EMPTYK, st);
    if (g < 0) continue;
    const uint32_t cv = l_cv[i];
    if (cv) {
      atomicMin(&mn[g], l_mn[i]);
      atomicMax(&mx[g], l_mx[i]);
      atomicAdd(&cnt_valid[g], cv);
    }
  }
}

template <typename K, typename V, bool HAS_VV>
int launch_partitioned_minmax(const K* keys, const uint32_t* kvalid, const V* vals, const uint32_t* vvalid, int64_t n, PartPlan* plan,
                              K* pkeys, V* pvals, uint8_t* pflags, unsigned long long* table, uint32_t lg, unsigned long long* mn,
                              unsigned long long* mx, uint32_t* cv, GbState* st, int64_t max_groups, hipStream_t s)
{
  GX_HIP_TRY(hipMemsetAsync(plan, 0, sizeof(PartPlan), s));
  int64_t hb = div_up(n, 256 * 8 * 4 * NRANGE);
  if (hb > 256) hb = 256;
  if (hb < 1) hb = 1;
  constexpr int ESZ      = sizeof(K) > sizeof(V) ? sizeof(K) : sizeof(V);
  constexpr size_t lds_s = (size_t)PTILE * ESZ + (HAS_VV ? PTILE : 0) + NPART * 4 * 2 + NPART * 8 * 2 + 64;
  auto ks                = k_part_scatter<K, V, HAS_VV>;
  constexpr int S        = lds_slots_mm<K, HAS_VV>();
  constexpr size_t lds_a = (size_t)(S + 1) * 16 + (size_t)(S + 2) * sizeof(K) + (size_t)(S + 1) * 4;
  auto ka                = k_part_minmax<K, V, HAS_VV>;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ka), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
    attr_set = true;
  }
  const int nsplit    = g_gb_nsplit;
  const int nsub      = lds_nsub(max_groups, S);
  const unsigned sgrd = (unsigned)div_up(n, PTILE);
  const unsigned agrd = (unsigned)((1 << g_gb_pbits) * nsplit * nsub);
  const bool spec     = part_speculative(n);
  int gated           = 0;
  if (spec) {
    const uint32_t cap       = 1u;  // != 0: the speculative form (slot tables in the plan)
    const int stride         = slot_stride(n);
    const int64_t range_rows = range_tiles(n) * PTILE;
    int64_t sblocks          = div_up(div_up(n, (int64_t)stride * GX_WAVE), (int64_t)4 * 8);
    if (sblocks > 2048) sblocks = 2048;
    hipLaunchKernelGGL((k_slot_sample<K>), dim3((unsigned)sblocks), dim3(256), 0, s, keys, kvalid, n, plan, stride, range_rows);
    hipLaunchKernelGGL((k_slot_plan<PartPlan>), dim3(1), dim3(NPART), 0, s, plan, n, stride, range_rows, (unsigned long long)slot_elems(n, stride));
    hipLaunchKernelGGL(ks, dim3(sgrd), dim3(PBT), lds_s, s, keys, kvalid, vals, vvalid, n, plan, pkeys, pvals, pflags, NRANGE, cap, 0);
    hipLaunchKernelGGL(ka, dim3(agrd), dim3(ABT), lds_a, s, pkeys, pvals, pflags, plan, nsplit, nsub, table, lg, mn, mx, cv, st, cap, 0);
    hipLaunchKernelGGL(k_part_reset_cursors, dim3(1), dim3(NPART), 0, s, plan);
    gated = 1;
  }
  hipLaunchKernelGGL((k_part_hist<K>), dim3((unsigned)(hb * NRANGE)), dim3(256), 0, s, keys, kvalid, n, plan, g_gb_nrange, gated);
  hipLaunchKernelGGL(k_part_offsets, dim3(1), dim3(NPART), 0, s, plan, gated);
  hipLaunchKernelGGL(ks, dim3(sgrd), dim3(PBT), lds_s, s, keys, kvalid, vals, vvalid, n, plan, pkeys, pvals, pflags, g_gb_nrange, 0u, gated);
  hipLaunchKernelGGL(ka, dim3(agrd), dim3(ABT), lds_a, s, pkeys, pvals, pflags, plan, nsplit, nsub, table, lg, mn, mx, cv, st, 0u, gated);
  GX_LAUNCH_CHECK();
  return 0;
}

template <typename K, typename V>
int minmax_impl(const void* keys, const uint32_t* kvalid, const void* vals, const uint32_t* vvalid, int64_t n,
                int64_t max_groups, void* out_keys, void* out_min, void* out_max, int32_t* out_cv, int64_t* ngroups,
                void* tmp, size_t* tmp_bytes, hipStream_t s)
{
  const uint32_t lg  = log2_cap(max_groups);
  const uint64_t cap = 1ull << lg;
  Carver c(tmp);
  GbState* st               = c.take<GbState>(1);
  unsigned long long* table = c.take<unsigned long long>(cap);
  uint32_t* cv              = c.take<uint32_t>(cap + 1);
  unsigned long long* mx    = c.take<unsigned long long>(cap + 1);  // zero-initialised with the block above
  unsigned long long* mn    = c.take<unsigned long long>(cap + 1);  // all-ones
  uint32_t* pos             = c.take<uint32_t>(cap + 1);
  uint32_t* partials        = c.take<uint32_t>(scan::partials_count(cap + 1));
  // large inputs: radix-partition the rows and fold each partition in LDS (the scratch layout depends on n only,
  // so the size query and the call agree)
  const bool partitioned = g_gb_algorithm != 1 && n > 0 && (g_gb_algorithm == 2 || n >= PART_MIN_ROWS);
  PartPlan* plan         = c.take<PartPlan>(1);
  K* pkeys               = partitioned ? c.take<K>(part_elems(n)) : nullptr;
  V* pvals               = partitioned ? c.take<V>(part_elems(n)) : nullptr;
  uint8_t* pflags        = partitioned ? c.take<uint8_t>(part_elems(n)) : nullptr;
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  GX_HIP_TRY(hipMemsetAsync(tmp, 0, (size_t)(reinterpret_cast<char*>(mn) - static_cast<char*>(tmp)), s));
  GX_HIP_TRY(hipMemsetAsync(mn, 0xFF, (cap + 1) * sizeof(unsigned long long), s));
  if (partitioned) {
    int prc;
    if (vvalid)
      prc = launch_partitioned_minmax<K, V, true>(static_cast<const K*>(keys), kvalid, static_cast<const V*>(vals), vvalid, n, plan,
                                                  pkeys, pvals, pflags, table, lg, mn, mx, cv, st, max_groups, s);
    else
      prc = launch_partitioned_minmax<K, V, false>(static_cast<const K*>(keys), kvalid, static_cast<const V*>(vals), vvalid, n, plan,
                                                   pkeys, pvals, pflags, table, lg, mn, mx, cv, st, max_groups, s);
    if (prc) return prc;
  } else if (n > 0) {
    int64_t blocks = div_up(n, GBT * 8);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL((k_minmax<K, V>), dim3((unsigned)blocks), dim3(GBT), 0, s, static_cast<const K*>(keys), kvalid,
                       static_cast<const V*>(vals), vvalid, n, table, lg, mn, mx, cv, st);
  }
  OccLoader ld{table, cap, st};
  int rc = scan::device_scan<uint32_t, uint32_t>(ld, (int64_t)cap + 1, 0u, SumOp(), false, pos, partials, s);
  if (rc) return rc;
  int64_t blocks = div_up((int64_t)cap + 1, GBT * 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((k_minmax_compact<K, V>), dim3((unsigned)blocks), dim3(GBT), 0, s, table, cap, pos,
                     partials + scan::num_chunks((int64_t)cap + 1), mn, mx, cv, st, max_groups, static_cast<K*>(out_keys),
                     static_cast<V*>(out_min), static_cast<V*>(out_max), out_cv, reinterpret_cast<long long*>(ngroups));
  GX_LAUNCH_CHECK();
  return 0;
}

template <typename K>
int minmax_dispatch(int val_dtype, const void* keys, const uint32_t* kvalid, const void* vals, const uint32_t* vvalid,
                    int64_t n, int64_t max_groups, void* out_keys, void* out_min, void* out_max, int32_t* out_cv,
                    int64_t* ngroups, void* tmp, size_t* tmp_bytes, hipStream_t s)
{
#define GX_MM(V) return minmax_impl<K, V>(keys, kvalid, vals, vvalid, n, max_groups, out_keys, out_min, out_max, out_cv, ngroups, tmp, tmp_bytes, s)
  switch (val_dtype) {
    case GX_INT8: GX_MM(int8_t);
    case GX_INT16: GX_MM(int16_t);
    case GX_INT32: GX_MM(int32_t);
    case GX_INT64: GX_MM(int64_t);
    case GX_BOOL8:
    case GX_UINT8: GX_MM(uint8_t);
    case GX_UINT16: GX_MM(uint16_t);
    case GX_UINT32: GX_MM(uint32_t);
    case GX_UINT64: GX_MM(uint64_t);
    case GX_FLOAT32: GX_MM(float);
    case GX_FLOAT64: GX_MM(double);
    default: return GX_EDTYPE;
  }
#undef GX_MM
}

}  // namespace gb
}  // namespace gx

extern "C" {

int gx_groupby_min_max(int key_dtype, const void* keys, const uint32_t* keys_valid, int val_dtype, const void* vals,
                       const uint32_t* vals_valid, int64_t n, int64_t max_groups, void* out_keys, void* out_min,
                       void* out_max, int32_t* out_count_valid, int64_t* ngroups_dev, void* tmp, size_t* tmp_bytes,
                       gx_stream_t s)
{
  if (n < 0 || max_groups < 0 || !tmp_bytes) return GX_EINVAL;
  if (tmp && (!ngroups_dev || (n > 0 && (!keys || !vals)) || (max_groups > 0 && !out_keys))) return GX_EINVAL;
  switch (key_dtype) {
    case GX_INT32:
    case GX_UINT32:
      return gx::gb::minmax_dispatch<uint32_t>(val_dtype, keys, keys_valid, vals, vals_valid, n, max_groups, out_keys, out_min,
                                               out_max, out_count_valid, ngroups_dev, tmp, tmp_bytes, s);
    case GX_INT64:
    case GX_UINT64:
      return gx::gb::minmax_dispatch<uint64_t>(val_dtype, keys, keys_valid, vals, vals_valid, n, max_groups, out_keys, out_min,
                                               out_max, out_count_valid, ngroups_dev, tmp, tmp_bytes, s);
    default: return GX_EDTYPE;
  }
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Compound aggregations on top of the passes above.
//   SUM_OF_SQUARES / M2 / VARIANCE / STD (hash path of the reference: SUM_OF_SQUARES is a single-pass
//   aggregation, M2 = sum_sqr - sum * sum / count, VAR = M2 / (count - ddof), STD = sqrt(VAR), null when
//   count - ddof <= 0: cpp/src/groupby/common/m2_var_std.cu:44-61,153-190,
//   cpp/src/groupby/hash/hash_compound_agg_finalizer.cu:135-186).  gx_square produces the squared values
//   in the SUM accumulator type (integers -> int64, wrapping like the reference's int64 accumulator;
//   floats keep their type); the caller sums them with gx_groupby_sum_count.
//   ARGMIN / ARGMAX: the row of the group's MIN / MAX value.  gx_groupby_arg_select takes the row -> group
//   map (gx_join_lookup on the output keys) and the per-group target value and keeps the SMALLEST row
//   whose value equals the target (floats: NaN == NaN, -0.0 == +0.0, the order MIN / MAX used).
// ------------------------------------------------------------------------------------------------
namespace gx {
namespace gb {

template <typename V, typename S>
__global__ void __launch_bounds__(256) k_square(const V* __restrict__ in, int64_t n, S* __restrict__ out)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    if constexpr (std::is_integral<S>::value) {
      const unsigned long long x = (unsigned long long)(long long)in[i];  // wraps mod 2^64 like an int64 accumulator
      out[i]                     = (S)(x * x);
    } else {
      const S x = (S)in[i];
      out[i]    = x * x;
    }
  }
}

// mode 0: M2, 1: VARIANCE, 2: STD
template <typename S>
__global__ void __launch_bounds__(256) k_var(const S* __restrict__ sumsq, const S* __restrict__ sum,
                                             const int32_t* __restrict__ cnt, int64_t n, int ddof, int mode,
                                             double* __restrict__ out, uint32_t* __restrict__ mask, unsigned long long* nulls)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t nround = div_up(n, (int64_t)64) * 64;  // whole waves: the ballot below needs every lane
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nround; i += stride) {
    bool ok  = false;
    double r = 0.0;
    if (i < n) {
      const int32_t c = cnt[i];
      if (c > 0) {
        const double ss = (double)sumsq[i], sm = (double)sum[i];
        const double m2 = ss - sm * sm / (double)c;
        if (mode == 0) {
          r  = m2;
          ok = true;
        } else if (c - ddof > 0) {
          r  = m2 / (double)(c - ddof);
          if (mode == 2) r = sqrt(r);
          ok = true;
        }
      } else if (mode == 0) {
        ok = true;  // M2 of an empty group is 0 and valid (m2_var_std.cu:52-53)
      }
      out[i] = r;
    }
    const uint64_t b = ballot(ok);
    if (lane_id() == 0 && i < n) {
      mask[i >> 5]       = (uint32_t)b;
      if (i + 32 < n) mask[(i >> 5) + 1] = (uint32_t)(b >> 32);
      const int64_t rows = (n - i) < 64 ? (n - i) : 64;
      const int bad      = (int)rows - __builtin_popcountll(b);
      if (bad) atomicAdd(nulls, (unsigned long long)bad);
    }
  }
}

template <typename V>
__device__ __forceinline__ bool same_value(V a, V b)
{
  if constexpr (std::is_floating_point<V>::value) return a == b || (a != a && b != b);
  return a == b;
}

template <typename V>
__global__ void __launch_bounds__(256) k_arg_select(const V* __restrict__ vals, const uint32_t* __restrict__ valid,
                                                    const int32_t* __restrict__ gid, int64_t n,
                                                    const V* __restrict__ target, int32_t* __restrict__ out)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int32_t g = gid[i];
    if (g < 0) continue;                                // null / dropped key
    if (valid && !bit_is_set(valid, i)) continue;       // null value
    if (same_value<V>(vals[i], target[g])) atomicMin(&out[g], (int32_t)i);
  }
}

template <typename V>
int arg_select_impl(const void* vals, const uint32_t* valid, const int32_t* gid, int64_t n, const void* target, int64_t g,
                    int32_t* out, hipStream_t s)
{
  GX_HIP_TRY(hipMemsetAsync(out, 0x7F, (size_t)g * sizeof(int32_t), s));  // 0x7F7F7F7F: above every row index in use
  if (n == 0) return 0;
  int64_t blocks = div_up(n, (int64_t)256 * 4);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL((k_arg_select<V>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const V*>(vals), valid, gid, n,
                     static_cast<const V*>(target), out);
  GX_LAUNCH_CHECK();
  return 0;
}

}  // namespace gb
}  // namespace gx

extern "C" {

int gx_square(int dtype, const void* in, int64_t n, void* out, gx_stream_t s)
{
  if (n < 0 || (n > 0 && (!in || !out))) return GX_EINVAL;
  if (n == 0) return 0;
  int64_t blocks = gx::div_up(n, (int64_t)256 * 8);
  if (blocks > 16384) blocks = 16384;
#define GX_SQ(V, S) hipLaunchKernelGGL((gx::gb::k_square<V, S>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const V*>(in), n, static_cast<S*>(out)); break
  switch (dtype) {
    case GX_INT8: GX_SQ(int8_t, int64_t);
    case GX_INT16: GX_SQ(int16_t, int64_t);
    case GX_INT32: GX_SQ(int32_t, int64_t);
    case GX_INT64: GX_SQ(int64_t, int64_t);
    case GX_BOOL8:
    case GX_UINT8: GX_SQ(uint8_t, int64_t);
    case GX_UINT16: GX_SQ(uint16_t, int64_t);
    case GX_UINT32: GX_SQ(uint32_t, int64_t);
    case GX_UINT64: GX_SQ(uint64_t, int64_t);
    case GX_FLOAT32: GX_SQ(float, float);
    case GX_FLOAT64: GX_SQ(double, double);
    default: return GX_EDTYPE;
  }
#undef GX_SQ
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_var_from_sums(int sum_dtype, const void* sum_sqr, const void* sum, const int32_t* count, int64_t n, int ddof, int mode,
                     double* out, uint32_t* mask_out, int64_t* null_count_dev, gx_stream_t s)
{
  if (n < 0 || mode < 0 || mode > 2 || !null_count_dev) return GX_EINVAL;
  if (n > 0 && (!sum_sqr || !sum || !count || !out || !mask_out)) return GX_EINVAL;
  GX_HIP_TRY(hipMemsetAsync(null_count_dev, 0, sizeof(int64_t), s));
  if (n == 0) return 0;
  int64_t blocks = gx::div_up(n, (int64_t)256);
  if (blocks > 4096) blocks = 4096;
  auto* nulls = reinterpret_cast<unsigned long long*>(null_count_dev);
  switch (sum_dtype) {
    case GX_INT64: hipLaunchKernelGGL((gx::gb::k_var<int64_t>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const int64_t*>(sum_sqr), static_cast<const int64_t*>(sum), count, n, ddof, mode, out, mask_out, nulls); break;
    case GX_FLOAT64: hipLaunchKernelGGL((gx::gb::k_var<double>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const double*>(sum_sqr), static_cast<const double*>(sum), count, n, ddof, mode, out, mask_out, nulls); break;
    case GX_FLOAT32: hipLaunchKernelGGL((gx::gb::k_var<float>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const float*>(sum_sqr), static_cast<const float*>(sum), count, n, ddof, mode, out, mask_out, nulls); break;
    default: return GX_EDTYPE;
  }
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_groupby_arg_select(int val_dtype, const void* vals, const uint32_t* vals_valid, const int32_t* group_of_row, int64_t n,
                          const void* target, int64_t num_groups, int32_t* out_rows, gx_stream_t s)
{
  if (n < 0 || num_groups < 0 || (n > 0 && (!vals || !group_of_row)) || (num_groups > 0 && (!target || !out_rows)))
    return GX_EINVAL;
  if (num_groups == 0) return 0;
#define GX_AS(V) return gx::gb::arg_select_impl<V>(vals, vals_valid, group_of_row, n, target, num_groups, out_rows, s)
  switch (val_dtype) {
    case GX_INT8: GX_AS(int8_t);
    case GX_INT16: GX_AS(int16_t);
    case GX_INT32: GX_AS(int32_t);
    case GX_INT64: GX_AS(int64_t);
    case GX_BOOL8:
    case GX_UINT8: GX_AS(uint8_t);
    case GX_UINT16: GX_AS(uint16_t);
    case GX_UINT32: GX_AS(uint32_t);
    case GX_UINT64: GX_AS(uint64_t);
    case GX_FLOAT32: GX_AS(float);
    case GX_FLOAT64: GX_AS(double);
    default: return GX_EDTYPE;
  }
#undef GX_AS
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// Round 3: groupby on SEVERAL key columns in one partition pass, rows compared inside the LDS table
// (the reference hashes the row once and compares rows in its probe: primitive_row_operators.cuh:95-163, 247-268).
// Round 2 reduced a multi-column key to ONE 8-byte word per row (packed values or a certified 64-bit row hash) and paid
// for the certificate with one extra MIN/MAX groupby per key column (66 ms for 2 x int64 keys at 1e9 rows).  Here the W
// key words of a row travel together:
//   k_wide_scatter    one radix-partition pass on the top bits of the row hash; the hash column goes through LDS first so
//                     that the partition of the element a thread writes out is known, then every key column and the value
//                     column follow through the same 64 KiB buffer (columns are re-read per pass -- Infinity-Cache hits --
//                     instead of being held in registers: W x 16 x 2 registers do not fit next to a second workgroup);
//   k_wide_aggregate  one 1024-thread workgroup per partition: open addressing on a 32-bit tag {busy, 30 hash bits} claimed
//                     by CAS, then the W key words are written and the tag is released; a row that meets an equal tag waits
//                     for the release and compares the W words, so two different rows that share 30 tag bits -- or all 64
//                     hash bits -- simply occupy two slots.  A partition's groups are complete (every row of a group hashes
//                     to it), so they go straight to the output: no global table, no certificate.
// Speculative padded slots as in the single-key path; a slot or LDS table that overflows, or more than max_groups groups,
// is reported through *ngroups (-2 / -1) and the caller falls back to the round-2 path / retries with a larger bound.
// No nulls (the host takes the round-2 path for nullable keys or values).
// ------------------------------------------------------------------------------------------------------------------
namespace gx {
namespace gb {

constexpr int WMAX = 4;
struct WideCols {
  const unsigned long long* k[WMAX];
};
struct WideOut {
  unsigned long long* k[WMAX];
};
__device__ __forceinline__ unsigned long long wide_mix(unsigned long long h, unsigned long long k)
{
  h = (h ^ k) * 0xFF51AFD7ED558CCDull;
  return h ^ (h >> 32);
}
__device__ __forceinline__ unsigned long long wide_fin(unsigned long long h)
{
  h *= 0xC4CEB9FE1A85EC53ull;
  return h ^ (h >> 29);
}

struct WidePlan {
  unsigned long long cursor[NRANGE][NPART];  // rows written to slot (partition, range)
  uint32_t samp[NRANGE][NPART];               // sample histogram of the partition digit, per input range
  uint32_t slot0[NRANGE][NPART];              // first row of slot (range, partition) in the partitioned arrays ...
  uint32_t cap0[NRANGE][NPART];               // ... and its capacity
  alignas(128) unsigned int overflow;         // a slot or an LDS table outgrew its capacity
  alignas(128) unsigned long long ngroups;    // output cursor
};

// Slot capacities come from a SAMPLE of the rows (every stride-th 64-row chunk), as in the sort's cursor path: rows of one
// group all go to one partition, so partition sizes carry the variance of the GROUP sizes -- 1e6 groups over 512 partitions
// are uneven by +-2.3 %, far beyond the 8 sigma of row-level noise a fixed mean + margin allows (the first version of this
// path overflowed on BASELINE-like inputs for exactly that reason); a sample sees the groups.
template <int W>
__global__ void __launch_bounds__(256) k_wide_sample(WideCols keys, int64_t n, WidePlan* plan, int stride, int64_t range_rows)
{
  __shared__ uint32_t s_hist[NRANGE * NPART];
  const unsigned tid = threadIdx.x, lane = lane_id();
  for (int i = tid; i < NRANGE * NPART; i += 256) s_hist[i] = 0;
  __syncthreads();
  const int psh         = 64 - d_gb_pbits;
  const int64_t step    = (int64_t)stride * GX_WAVE;
  const int64_t nchunks = div_up(n, step);
  const int64_t nw      = (int64_t)gridDim.x * 4;
  for (int64_t c = (int64_t)blockIdx.x * 4 + tid / GX_WAVE; c < nchunks; c += nw) {
    const int64_t row = c * step + lane;
    const bool live   = row < n;
    unsigned long long x = 0x9E3779B97F4A7C15ull;
#pragma unroll
    for (int w = 0; w < W; ++w) x = wide_mix(x, keys.k[w][live ? row : 0]);
    x = wide_fin(x);
    const int64_t r64 = range_rows > 0 ? row / range_rows : (int64_t)(NRANGE - 1);
    const int r       = r64 < NRANGE - 1 ? (int)r64 : NRANGE - 1;
    (void)lds_rank(s_hist + r * NPART, (uint32_t)(x >> psh), live);
  }
  __syncthreads();
  for (int i = tid; i < NRANGE * NPART; i += 256) {
    const uint32_t c = s_hist[i];
    if (c) atomicAdd(&plan->samp[i / NPART][i % NPART], c);
  }
}

// 4096-row tiles (8 rows per thread): the W key words of a row stay in registers from the hash to their column pass, and
// W x 8 x 2 VGPRs still leave room for the second workgroup per CU (16 rows per thread: 136 / 161 / 193 VGPRs for W = 2 / 3 / 4)
constexpr int WRPT  = 8;
constexpr int WTILE = PBT * WRPT;

template <int W, typename V>
__global__ void __launch_bounds__(PBT) k_wide_scatter(WideCols keys, const V* __restrict__ vals, int64_t n, WidePlan* plan, WideOut pkeys,
                                                      V* __restrict__ pvals)
{
  if (plan->overflow != 0) return;  // (the plan's slots do not fit: cannot happen with the host's bound)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* s_w = reinterpret_cast<unsigned long long*>(smem);                      // WTILE words
  uint32_t* s_cnt         = reinterpret_cast<uint32_t*>(smem + (size_t)WTILE * 8);           // NPART
  uint32_t* s_start       = s_cnt + NPART;                                                    // NPART
  unsigned long long* s_delta = reinterpret_cast<unsigned long long*>(s_start + NPART);      // NPART
  unsigned long long* s_limit = s_delta + NPART;                                              // NPART: end of the partition's slot
  uint32_t* s_scan        = reinterpret_cast<uint32_t*>(s_limit + NPART);                    // 16
  __shared__ uint32_t s_total;
  const unsigned tid = threadIdx.x;
  const int psh      = 64 - d_gb_pbits;
  const int64_t tile = xcd_swizzle(blockIdx.x, gridDim.x);
  const int64_t rper = div_up(n, (int64_t)WTILE) / NRANGE;  // tiles per input range (the last range takes the rest)
  const int range    = (rper > 0 && tile / rper < NRANGE - 1) ? (int)(tile / rper) : NRANGE - 1;
  const int64_t base = tile * WTILE;
  const int nvalid   = (int)((n - base < (int64_t)WTILE) ? (n - base) : (int64_t)WTILE);  // >= 1
  if (tid < NPART) s_cnt[tid] = 0;
  // ---- the row: W key words + value, loaded unconditionally (padding rows repeat the tile's last row and are never ranked)
  unsigned long long k[W][WRPT];
  V v[WRPT];
#pragma unroll
  for (int w = 0; w < W; ++w) {
    const unsigned long long* kt = keys.k[w] + base;
#pragma unroll
    for (int j = 0; j < WRPT; ++j) {
      const int idx = j * PBT + (int)tid;
      k[w][j]       = kt[idx < nvalid ? idx : nvalid - 1];
    }
  }
  {
    const V* vt = vals + base;
#pragma unroll
    for (int j = 0; j < WRPT; ++j) {
      const int idx = j * PBT + (int)tid;
      v[j]          = vt[idx < nvalid ? idx : nvalid - 1];
    }
  }
  __syncthreads();
  unsigned long long h[WRPT];
  uint32_t packed[WRPT];
#pragma unroll
  for (int j = 0; j < WRPT; ++j) {
    unsigned long long x = 0x9E3779B97F4A7C15ull;
#pragma unroll
    for (int w = 0; w < W; ++w) x = wide_mix(x, k[w][j]);
    h[j]                = wide_fin(x);
    const uint32_t part = (uint32_t)(h[j] >> psh);
    packed[j]           = (part << 16) | ((j * PBT + (int)tid < nvalid) ? atomicAdd(&s_cnt[part], 1u) : 0u);
  }
  __syncthreads();
  const uint32_t c = (tid < NPART) ? s_cnt[tid] : 0u;
  uint32_t total;
  const uint32_t st = block_exclusive_scan<PBT>(c, 0u, SumOp(), s_scan, &total);
  if (tid < NPART) {
    s_start[tid]         = st;
    unsigned long long g = 0;
    if (c) g = atomicAdd(&plan->cursor[range][tid], (unsigned long long)c);
    const uint32_t scap = plan->cap0[range][tid];
    if (c && g + c > scap) plan->overflow = 1u;
    const unsigned long long sb = plan->slot0[range][tid];
    s_limit[tid] = sb + scap;
    s_delta[tid] = sb + g - st;
  }
  if (tid == 0) s_total = total;
  __syncthreads();
  const int ntot = (int)s_total;
  // ---- the hashes through LDS: the partition of every output position, kept in registers for all column passes
#pragma unroll
  for (int j = 0; j < WRPT; ++j) {
    if (j * PBT + (int)tid < nvalid) s_w[s_start[packed[j] >> 16] + (packed[j] & 0xFFFFu)] = h[j];
  }
  __syncthreads();
  unsigned short obin[WRPT];
#pragma unroll
  for (int j = 0; j < WRPT; ++j) {
    const int i = j * PBT + (int)tid;
    obin[j]     = 0xFFFFu;
    if (i < ntot) {
      const uint32_t b             = (uint32_t)(s_w[i] >> psh);
      const unsigned long long dst = s_delta[b] + (unsigned long long)i;
      if (dst < s_limit[b]) obin[j] = (unsigned short)b;  // else: beyond the slot (flagged)
    }
  }
  // ---- key columns, then the values, through the same buffer
#pragma unroll
  for (int w = 0; w < W; ++w) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < WRPT; ++j) {
      if (j * PBT + (int)tid < nvalid) s_w[s_start[packed[j] >> 16] + (packed[j] & 0xFFFFu)] = k[w][j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < WRPT; ++j) {
      const int i = j * PBT + (int)tid;
      if (obin[j] != 0xFFFFu) pkeys.k[w][s_delta[obin[j]] + (unsigned long long)i] = s_w[i];
    }
  }
  __syncthreads();
  V* s_v = reinterpret_cast<V*>(smem);
#pragma unroll
  for (int j = 0; j < WRPT; ++j) {
    if (j * PBT + (int)tid < nvalid) s_v[s_start[packed[j] >> 16] + (packed[j] & 0xFFFFu)] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < WRPT; ++j) {
    const int i = j * PBT + (int)tid;
    if (obin[j] != 0xFFFFu) pvals[s_delta[obin[j]] + (unsigned long long)i] = s_v[i];
  }
}

template <int W>
constexpr int wide_slots()
{
  return (LDS_BUDGET / (4 + 8 * W + 4 + 16)) / 256 * 256;
}

template <int W, typename V, bool IS_FLOAT>
__global__ void __launch_bounds__(ABT) k_wide_aggregate(WideOut pkeys, const V* __restrict__ pvals, WidePlan* plan, int nsub,
                                                        int64_t max_groups, WideOut out_keys, void* out_sum, int32_t* out_cv)
{
  if (plan->overflow != 0) return;
  constexpr int S            = wide_slots<W>();
  constexpr uint32_t MAXKEYS = (uint32_t)(S - S / 8);
  constexpr uint32_t BUSY = 0x80000000u, LIVE = 0x40000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* l_sum           = reinterpret_cast<double*>(smem);                       // S
  double* l_comp          = l_sum + S;                                             // S
  unsigned long long* l_k = reinterpret_cast<unsigned long long*>(l_comp + S);     // W x S
  uint32_t* l_tag         = reinterpret_cast<uint32_t*>(l_k + (size_t)W * S);      // S
  uint32_t* l_cv          = l_tag + S;                                             // S
  __shared__ uint32_t s_nkeys;
  __shared__ uint32_t s_scan[ABT / GX_WAVE + 1];
  __shared__ unsigned long long s_base;
  const unsigned tid = threadIdx.x;
  for (int i = tid; i < S; i += ABT) {
    l_sum[i]  = 0.0;
    l_comp[i] = 0.0;
    l_tag[i]  = 0u;
    l_cv[i]   = 0u;
  }
  if (tid == 0) s_nkeys = 0;
  __syncthreads();
  const int sub  = (int)(blockIdx.x % (unsigned)nsub);
  const int part = (int)(blockIdx.x / (unsigned)nsub);
  const int pb   = d_gb_pbits;
  constexpr int U = 4;
  for (int rg = 0; rg < NRANGE; ++rg) {
    const unsigned long long fill = plan->cursor[rg][part];
    const unsigned long long cap  = plan->cap0[rg][part];
    const unsigned long long p0   = plan->slot0[rg][part];
    const unsigned long long p1   = p0 + (fill < cap ? fill : cap);
    for (unsigned long long i0 = p0 + tid; i0 < p1; i0 += (unsigned long long)ABT * U) {
      unsigned long long k[U][W];
      V v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned long long i = i0 + (unsigned long long)u * ABT;
        const bool in              = i < p1;
#pragma unroll
        for (int w = 0; w < W; ++w) k[u][w] = in ? pkeys.k[w][i] : 0ull;
        v[u] = in ? pvals[i] : V(0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (i0 + (unsigned long long)u * ABT >= p1) continue;
        unsigned long long hh = 0x9E3779B97F4A7C15ull;
#pragma unroll
        for (int w = 0; w < W; ++w) hh = wide_mix(hh, k[u][w]);
        hh = wide_fin(hh);
        if (nsub > 1 && (int)((hh >> 16) & (unsigned long long)(nsub - 1)) != sub) continue;
        const uint32_t want = ((uint32_t)hh & 0x3FFFFFFFu) | LIVE;  // released tag of this row's hash
        uint32_t h          = (uint32_t)((((hh >> (32 - pb)) & 0xFFFFFFFFull) * (unsigned long long)S) >> 32);
        int slot            = -1;
        // NO exit edge between a successful claim and its release: a `break` there makes the claim block a loop exit, which
        // the compiler lays out BEHIND the loop -- the owner lane would then sit at the loop's end with its tag still busy
        // while lanes of its own wave spin on that tag inside the loop (seen as a hang in the first GPU run of this kernel)
        bool searching = true;
        for (int probes = 0; searching && probes < S; ++probes) {
          uint32_t tag = __hip_atomic_load(&l_tag[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (tag == 0u) {
            if (s_nkeys >= MAXKEYS) {  // table (nearly) full: the caller falls back
              searching = false;
            } else {
              tag = atomicCAS(&l_tag[h], 0u, want | BUSY);
              if (tag == 0u) {  // claimed: write the row's key words, then release the tag
#pragma unroll
                for (int w = 0; w < W; ++w) l_k[(size_t)w * S + h] = k[u][w];
                __hip_atomic_store(&l_tag[h], want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                atomicAdd(&s_nkeys, 1u);
                slot      = (int)h;
                searching = false;
                tag       = want;
              }
            }
          }
          if (searching && (tag & 0x7FFFFFFFu) == want) {
            // same 30 hash bits: wait for the owner's key words (an owner in this wave released in the branch above, which
            // every lane of the wave has left by now), then compare the row
            while (tag & BUSY) {
              __builtin_amdgcn_s_sleep(1);
              tag = __hip_atomic_load(&l_tag[h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            bool same = true;
#pragma unroll
            for (int w = 0; w < W; ++w) same = same && (l_k[(size_t)w * S + h] == k[u][w]);
            if (same) {
              slot      = (int)h;
              searching = false;
            }
          }
          if (searching) h = (h + 1 == (uint32_t)S) ? 0u : h + 1;
        }
        if (slot < 0) {
          plan->overflow = 1u;
          continue;
        }
        LdsAcc<V, IS_FLOAT>::add(l_sum, l_comp, slot, v[u]);
        atomicAdd(&l_cv[slot], 1u);
      }
    }
  }
  __syncthreads();
  // ---- this workgroup's groups are complete: straight to the output
  uint32_t mine = 0;
  for (int i = tid; i < S; i += ABT) mine += l_tag[i] != 0u ? 1u : 0u;
  uint32_t total;
  const uint32_t before = block_exclusive_scan<ABT>(mine, 0u, SumOp(), s_scan, &total);
  if (tid == 0) s_base = total ? atomicAdd(&plan->ngroups, (unsigned long long)total) : 0ull;
  __syncthreads();
  unsigned long long p = s_base + before;
  for (int i = tid; i < S; i += ABT) {
    if (l_tag[i] == 0u) continue;
    if ((long long)p < max_groups) {
#pragma unroll
      for (int w = 0; w < W; ++w) out_keys.k[w][p] = l_k[(size_t)w * S + i];
      if (IS_FLOAT) {
        const double r = l_sum[i] + l_comp[i];
        if (sizeof(V) == 4) static_cast<float*>(out_sum)[p] = (float)r; else static_cast<double*>(out_sum)[p] = r;
      } else {
        static_cast<long long*>(out_sum)[p] = reinterpret_cast<const long long*>(l_sum)[i];
      }
      out_cv[p] = (int32_t)l_cv[i];
    }
    ++p;
  }
}

__global__ void k_wide_finish(const WidePlan* plan, int64_t max_groups, long long* ngroups)
{
  const long long g = (long long)plan->ngroups;
  *ngroups          = plan->overflow ? -2ll : (g > max_groups ? -1ll : g);
}

template <int W, typename V, bool IS_FLOAT>
int wide_impl(const void* const* key_cols, const void* vals, int64_t n, int64_t max_groups, void* const* out_key_cols, void* out_sum,
              int32_t* out_cv, int64_t* ngroups, void* tmp, size_t* tmp_bytes, hipStream_t s)
{
  const int stride   = slot_stride(n);
  const size_t elems = slot_elems(n, stride);
  Carver c(tmp);
  WidePlan* plan = c.take<WidePlan>(1);
  WideOut pk{};
  for (int w = 0; w < W; ++w) pk.k[w] = c.take<unsigned long long>(elems);
  V* pv = c.take<V>(elems);
  if (tmp == nullptr) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  WideCols kc{};
  WideOut ok{};
  for (int w = 0; w < W; ++w) {
    kc.k[w] = static_cast<const unsigned long long*>(key_cols[w]);
    ok.k[w] = static_cast<unsigned long long*>(out_key_cols[w]);
    if (n > 0 && !kc.k[w]) return GX_EINVAL;
    if (max_groups > 0 && !ok.k[w]) return GX_EINVAL;
  }
  GX_HIP_TRY(hipMemsetAsync(plan, 0, sizeof(WidePlan), s));
  if (n > 0) {
    constexpr size_t lds_s = (size_t)WTILE * 8 + NPART * 4 * 2 + NPART * 8 * 2 + 64;
    constexpr int S        = wide_slots<W>();
    constexpr size_t lds_a = (size_t)S * (16 + 8 * W + 8);
    auto ks                = k_wide_scatter<W, V>;
    auto ka                = k_wide_aggregate<W, V, IS_FLOAT>;
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ka), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
      attr_set = true;
    }
    // workgroups per partition: the groups one LDS table has to hold stay under half of its slots
    const double per_part = (double)(max_groups < 1 ? 1 : max_groups) / (double)(1 << g_gb_pbits);
    int nsub              = 1;
    while (nsub < 16 && per_part / nsub > 0.5 * S) nsub *= 2;
    const int64_t wtiles     = div_up(n, (int64_t)WTILE);
    const int64_t range_rows = (wtiles / NRANGE) * WTILE;  // rows per input range (whole tiles; the last range takes the rest)
    int64_t sblocks          = div_up(div_up(n, (int64_t)stride * GX_WAVE), (int64_t)4 * 8);
    if (sblocks > 2048) sblocks = 2048;
    if (sblocks < 1) sblocks = 1;
    hipLaunchKernelGGL((k_wide_sample<W>), dim3((unsigned)sblocks), dim3(256), 0, s, kc, n, plan, stride, range_rows);
    hipLaunchKernelGGL((k_slot_plan<WidePlan>), dim3(1), dim3(NPART), 0, s, plan, n, stride, range_rows, (unsigned long long)elems);
    hipLaunchKernelGGL(ks, dim3((unsigned)wtiles), dim3(PBT), lds_s, s, kc, static_cast<const V*>(vals), n, plan, pk, pv);
    hipLaunchKernelGGL(ka, dim3((unsigned)((1 << g_gb_pbits) * nsub)), dim3(ABT), lds_a, s, pk, pv, plan, nsub, max_groups, ok, out_sum, out_cv);
  }
  hipLaunchKernelGGL(k_wide_finish, dim3(1), dim3(1), 0, s, plan, max_groups, reinterpret_cast<long long*>(ngroups));
  GX_LAUNCH_CHECK();
  return 0;
}

template <int W>
int wide_dispatch_val(int val_dtype, const void* const* key_cols, const void* vals, int64_t n, int64_t max_groups, void* const* out_key_cols,
                      void* out_sum, int32_t* out_cv, int64_t* ngroups, void* tmp, size_t* tmp_bytes, hipStream_t s)
{
  switch (val_dtype) {
    case GX_INT32: return wide_impl<W, int32_t, false>(key_cols, vals, n, max_groups, out_key_cols, out_sum, out_cv, ngroups, tmp, tmp_bytes, s);
    case GX_INT64: return wide_impl<W, int64_t, false>(key_cols, vals, n, max_groups, out_key_cols, out_sum, out_cv, ngroups, tmp, tmp_bytes, s);
    case GX_FLOAT32: return wide_impl<W, float, true>(key_cols, vals, n, max_groups, out_key_cols, out_sum, out_cv, ngroups, tmp, tmp_bytes, s);
    case GX_FLOAT64: return wide_impl<W, double, true>(key_cols, vals, n, max_groups, out_key_cols, out_sum, out_cv, ngroups, tmp, tmp_bytes, s);
    default: return GX_EDTYPE;
  }
}

}  // namespace gb
}  // namespace gx

extern "C" {

int gx_groupby_sum_count_wide(int nkeys, const void* const* key_cols, int val_dtype, const void* vals, int64_t n, int64_t max_groups,
                              void* const* out_key_cols, void* out_sum, int32_t* out_count, int64_t* ngroups_dev, void* tmp,
                              size_t* tmp_bytes, gx_stream_t s)
{
  if (n < 0 || max_groups < 0 || !tmp_bytes || nkeys < 2 || nkeys > gx::gb::WMAX) return GX_EINVAL;
  if (tmp && (!key_cols || !out_key_cols || !ngroups_dev || (n > 0 && !vals) || (max_groups > 0 && (!out_sum || !out_count)))) return GX_EINVAL;
  static const void* const nulls[gx::gb::WMAX]  = {nullptr, nullptr, nullptr, nullptr};
  static void* const nulls_out[gx::gb::WMAX]    = {nullptr, nullptr, nullptr, nullptr};
  const void* const* kc = key_cols ? key_cols : nulls;
  void* const* oc       = out_key_cols ? out_key_cols : nulls_out;
  switch (nkeys) {
    case 2: return gx::gb::wide_dispatch_val<2>(val_dtype, kc, vals, n, max_groups, oc, out_sum, out_count, ngroups_dev, tmp, tmp_bytes, s);
    case 3: return gx::gb::wide_dispatch_val<3>(val_dtype, kc, vals, n, max_groups, oc, out_sum, out_count, ngroups_dev, tmp, tmp_bytes, s);
    default: return gx::gb::wide_dispatch_val<4>(val_dtype, kc, vals, n, max_groups, oc, out_sum, out_count, ngroups_dev, tmp, tmp_bytes, s);
  }
}

}  // extern "C"
