// gx_join.hip -- hash join build / count / probe for gfx950 (single 4- or 8-byte key column).
//
// Replaces the cuco::static_multiset the reference builds over {murmur3(row), row}
// (cpp/src/join/hash_join/hash_join.cu:62-99,112-148; hash_join_impl.cuh:50-57) and its
// count / retrieve probes (size_impl.cuh:26-62, retrieve_impl.cuh:28-113).
//
// MI355X-first differences (DESIGN.md "hash join"):
//   * slots hold the KEY itself ({key,row}: 16 B for 8-byte keys, 8 B for 4-byte keys), so a probe
//     is one random line fetch -- no second random read of the build column to verify equality
//     (the reference compares hashes in the slot, then dereferences the row: dispatch.cuh:58-75);
//   * power-of-two capacity, Fibonacci hash, linear probing: a chain stays inside one 128-B line;
//   * probe output is reserved with ONE atomic per 4096-row workgroup chunk (a single HBM-side
//     cursor word saturates near 88 atomics/us on this chip), and pairs leave in wave-contiguous
//     runs; the reference flushes per warp (partitioned_retrieve_kernels.cuh:57-211).
// Result order is unspecified (cpp/include/cudf/join/join.hpp:131-134).
#include "gx_common.hpp"
#include "gx_scan.hpp"
#include <cstdlib>

namespace gx {
namespace join {

constexpr int JBT  = 256;
constexpr int JRPT = 16;  // probe rows per thread
constexpr int JCHUNK = JBT * JRPT;
constexpr int32_t EMPTY_ROW = -1;
constexpr int32_t NO_MATCH  = INT32_MIN;  // cudf::JoinNoMatch (include/cudf/join/join.hpp:72)

struct alignas(256) TableHeader {
  uint64_t capacity;  // slots, power of two
  uint32_t log2cap;
  uint32_t key_size;
  int64_t build_rows;
};

template <typename K>
struct Slot;
template <>
struct alignas(16) Slot<uint64_t> {
  uint64_t key;
  int32_t row;
  int32_t pad;
};
template <>
struct alignas(8) Slot<uint32_t> {
  uint32_t key;
  int32_t row;
};

static inline uint32_t log2_capacity(int64_t build_rows, double load_factor)
{
  if (!(load_factor > 0.0) || load_factor > 1.0) load_factor = 0.5;  // CUCO_DESIRED_LOAD_FACTOR
  double want  = (double)(build_rows < 1 ? 1 : build_rows) / load_factor + 1.0;  // never completely full
  uint32_t lg  = 4;
  while ((double)(1ull << lg) < want) ++lg;
  return lg;
}

template <typename K>
__device__ __forceinline__ uint64_t slot_of(K key, uint32_t log2cap)
{
  return ((uint64_t)key * 0x9E3779B97F4A7C15ull) >> (64 - log2cap);
}

template <typename K>
__device__ __forceinline__ void load_slot(const Slot<K>* s, K& key, int32_t& row);
template <>
__device__ __forceinline__ void load_slot<uint64_t>(const Slot<uint64_t>* s, uint64_t& key, int32_t& row)
{
  const uint4 v = *reinterpret_cast<const uint4*>(s);  // one 16-B load
  key           = ((uint64_t)v.y << 32) | v.x;
  row           = (int32_t)v.z;
}
template <>
__device__ __forceinline__ void load_slot<uint32_t>(const Slot<uint32_t>* s, uint32_t& key, int32_t& row)
{
  const uint2 v = *reinterpret_cast<const uint2*>(s);
  key           = v.x;
  row           = (int32_t)v.y;
}

// 4-bit tag per slot, two per byte, stored behind the slots: 0 = empty slot, 1..15 = hash bits of the
// resident key just below the slot-index bits.  The partitioned probe keeps the tags of its ~128 k-slot
// sub-table in LDS and walks the probe chain THERE; only a tag match costs an L2 request.
constexpr int PJ_SUB_LOG2 = 17;  // slots per sub-table of the partitioned probe (64 KiB of tags)
template <typename K>
__device__ __forceinline__ uint32_t tag_of(K key, uint32_t log2cap)
{
  const uint32_t t = (uint32_t)(((uint64_t)key * 0x9E3779B97F4A7C15ull) >> (60 - log2cap)) & 15u;
  return t ? t : 8u;
}
template <typename K>
__global__ void __launch_bounds__(256) k_tags(const Slot<K>* __restrict__ slots, uint64_t cap, uint32_t log2cap,
                                              uint8_t* __restrict__ tags, const unsigned int* __restrict__ gate = nullptr)
{
  if (gate && *gate == 0) return;  // the fallback behind the window build (k_bw_build) that was not needed
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < (int64_t)(cap / 2); b += stride) {
    K k0, k1;
    int32_t r0, r1;
    load_slot<K>(&slots[2 * b], k0, r0);
    load_slot<K>(&slots[2 * b + 1], k1, r1);
    const uint32_t t0 = r0 == EMPTY_ROW ? 0u : tag_of<K>(k0, log2cap);
    const uint32_t t1 = r1 == EMPTY_ROW ? 0u : tag_of<K>(k1, log2cap);
    tags[b] = (uint8_t)(t0 | (t1 << 4));
  }
}
template <typename K>
static int launch_tags(void* table, uint32_t lg, hipStream_t s, const unsigned int* gate = nullptr)
{
  char* base     = static_cast<char*>(table);
  auto* slots    = reinterpret_cast<const Slot<K>*>(base + sizeof(TableHeader));
  uint8_t* tags  = reinterpret_cast<uint8_t*>(base + sizeof(TableHeader) + (sizeof(Slot<K>) << lg));
  int64_t blocks = div_up((int64_t)((1ull << lg) / 2), 256 * 8);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((k_tags<K>), dim3((unsigned)blocks), dim3(256), 0, s, slots, 1ull << lg, lg, tags, gate);
  GX_LAUNCH_CHECK();
  return 0;
}

template <typename K>
__global__ void __launch_bounds__(JBT) k_build(const K* __restrict__ keys, const uint32_t* __restrict__ valid,
                                               int64_t n, Slot<K>* slots, uint32_t log2cap, const int32_t* __restrict__ payload = nullptr)
{
  // payload != NULL: the slot carries payload[i] (>= 0) instead of the row number i -- the sharded join stores an encoded
  // global row there, so that its pairs need no gather afterwards
  const uint64_t mask  = (1ull << log2cap) - 1;
  const int64_t stride = (int64_t)gridDim.x * JBT;
  for (int64_t i = (int64_t)blockIdx.x * JBT + threadIdx.x; i < n; i += stride) {
    if (valid && !bit_is_set(valid, i)) continue;  // null build rows are never inserted
    const K key = keys[i];
    uint64_t h  = slot_of<K>(key, log2cap);
    for (;;) {
      const int32_t old = atomicCAS(&slots[h].row, EMPTY_ROW, payload ? payload[i] : (int32_t)i);
      if (old == EMPTY_ROW) {
        slots[h].key = key;  // nobody reads keys before the build kernel has finished
        break;
      }
      h = (h + 1) & mask;
    }
  }
}

// walk the chain of `key`: count matches, remember the first one
template <typename K>
__device__ __forceinline__ uint32_t chain_count(const Slot<K>* slots, uint64_t mask, uint32_t log2cap, K key,
                                                int32_t& first)
{
  uint32_t c = 0;
  uint64_t h = slot_of<K>(key, log2cap);
  for (;;) {
    K k;
    int32_t r;
    load_slot<K>(&slots[h], k, r);
    if (r == EMPTY_ROW) break;
    if (k == key) {
      if (c == 0) first = r;
      ++c;
    }
    h = (h + 1) & mask;
  }
  return c;
}

template <typename K, bool WRITE>
__global__ void __launch_bounds__(JBT) k_probe(const K* __restrict__ keys, const uint32_t* __restrict__ valid,
                                               int64_t n, const Slot<K>* __restrict__ slots, uint32_t log2cap,
                                               int left_outer, int32_t* __restrict__ out_probe,
                                               int32_t* __restrict__ out_build, int64_t capacity,
                                               unsigned long long* cursor)
{
  constexpr int NWJ = JBT / GX_WAVE;
  __shared__ unsigned long long s_wave_tot[NWJ];
  __shared__ unsigned long long s_base;
  const uint64_t mask   = (1ull << log2cap) - 1;
  const unsigned lane   = lane_id();
  const unsigned w      = threadIdx.x / GX_WAVE;
  const int64_t nchunks = div_up(n, (int64_t)JCHUNK);
  for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const int64_t wbase = chunk * JCHUNK + (int64_t)w * (JRPT * GX_WAVE) + lane;
    uint32_t cnt[JRPT];
    int32_t first[JRPT];
    uint32_t wave_total = 0;
#pragma unroll
    for (int j = 0; j < JRPT; ++j) {
      const int64_t i = wbase + j * GX_WAVE;
      cnt[j]          = 0;
      first[j]        = NO_MATCH;
      if (i < n) {
        const bool ok = !valid || bit_is_set(valid, i);
        if (ok) cnt[j] = chain_count<K>(slots, mask, log2cap, keys[i], first[j]);
        // (i, JoinNoMatch); first[j] already NO_MATCH.  left_outer bit 1: a null probe row emits nothing -- its
        // partners are the null build rows, which the caller appends (null == null)
        if ((left_outer & 1) && cnt[j] == 0 && (ok || !(left_outer & 2))) cnt[j] = 1;
      }
    }
    if (!WRITE) {
#pragma unroll
      for (int j = 0; j < JRPT; ++j) wave_total += cnt[j];
      wave_total = wave_reduce(wave_total, SumOp());
      if (lane == 0 && wave_total) atomicAdd(cursor, (unsigned long long)wave_total);
      continue;
    }
    // ---- per-lane exclusive offsets inside the wave, row-major over j
    uint32_t off[JRPT];
#pragma unroll
    for (int j = 0; j < JRPT; ++j) {
      uint32_t inc;
      if (ballot(cnt[j] > 1) == 0) {  // at most one match per row in this wave-row: one ballot
        const uint64_t b = ballot(cnt[j] == 1);
        off[j]           = wave_total + (uint32_t)__builtin_popcountll(b & lanemask_lt());
        inc              = (uint32_t)__builtin_popcountll(b);
      } else {
        const uint32_t s = wave_inclusive_scan(cnt[j], SumOp());
        off[j]           = wave_total + s - cnt[j];
        inc              = shfl(s, GX_WAVE - 1);
      }
      wave_total += inc;
    }
    // ---- one reservation per workgroup chunk
    if (lane == 0) s_wave_tot[w] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long tot = 0;
      for (int k = 0; k < NWJ; ++k) {
        const unsigned long long t = s_wave_tot[k];
        s_wave_tot[k]              = tot;
        tot += t;
      }
      s_base = tot ? atomicAdd(cursor, tot) : 0ull;
    }
    __syncthreads();
    const unsigned long long wave_base = s_base + s_wave_tot[w];
#pragma unroll
    for (int j = 0; j < JRPT; ++j) {
      const int64_t i = wbase + j * GX_WAVE;
      if (cnt[j] == 0) continue;
      unsigned long long pos = wave_base + off[j];
      if (cnt[j] == 1) {
        if ((int64_t)pos < capacity) {
          out_probe[pos] = (int32_t)i;
          out_build[pos] = first[j];
        }
      } else {  // duplicate build keys: walk the chain again (lines are cache-resident)
        const K key = keys[i];
        uint64_t h  = slot_of<K>(key, log2cap);
        for (;;) {
          K k;
          int32_t r;
          load_slot<K>(&slots[h], k, r);
          if (r == EMPTY_ROW) break;
          if (k == key) {
            if ((int64_t)pos < capacity) {
              out_probe[pos] = (int32_t)i;
              out_build[pos] = r;
            }
            ++pos;
          }
          h = (h + 1) & mask;
        }
      }
    }
    __syncthreads();  // s_wave_tot / s_base reused by the next chunk
  }
}

// ------------------------------------------------------------------------------------------------
// lookup / contains: the probe forms that need no output reservation.
//   k_lookup   : out[i] = first build row with probe key i, or JoinNoMatch -- a left join against
//                DISTINCT build keys (cudf::distinct_hash_join::left_join returns exactly this vector,
//                distinct_hash_join.hpp:111-116), and the row -> group map of the compound groupby
//                aggregations.
//   k_contains : one bit per probe row (any match), plus the number of set bits per 4096-row chunk;
//   k_emit_selected turns the bits into the ascending list of selected rows -- the reference's
//                contains map + thrust::copy_if of filtered_join::semi_anti_join
//                (src/join/filtered_join/filtered_join.cu:124-156), so the order matches it too.
// ------------------------------------------------------------------------------------------------
template <typename K>
__device__ __forceinline__ int32_t chain_first(const Slot<K>* slots, uint64_t mask, uint32_t log2cap, K key)
{
  uint64_t h = slot_of<K>(key, log2cap);
  for (;;) {
    K k;
    int32_t r;
    load_slot<K>(&slots[h], k, r);
    if (r == EMPTY_ROW) return NO_MATCH;
    if (k == key) return r;
    h = (h + 1) & mask;
  }
}

template <typename K>
__global__ void __launch_bounds__(256) k_lookup(const K* __restrict__ keys, const uint32_t* __restrict__ valid, int64_t n,
                                                const Slot<K>* __restrict__ slots, uint32_t log2cap,
                                                int32_t* __restrict__ out)
{
  const uint64_t mask  = (1ull << log2cap) - 1;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    int32_t r = NO_MATCH;
    if (!valid || bit_is_set(valid, i)) r = chain_first<K>(slots, mask, log2cap, keys[i]);
    out[i] = r;
  }
}

// matches per probe row (cudf::hash_join::{inner,left,full}_join_match_context: hash_join.hpp:259-340; the
// reference's per-row count pass of size_impl.cuh:26-62): counts[i] = max(min_count, #build rows with key i)
template <typename K>
__global__ void __launch_bounds__(256) k_count_rows(const K* __restrict__ keys, const uint32_t* __restrict__ valid, int64_t n,
                                                    const Slot<K>* __restrict__ slots, uint32_t log2cap, int32_t min_count,
                                                    int32_t* __restrict__ counts)
{
  const uint64_t mask  = (1ull << log2cap) - 1;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    int32_t c = 0, first;
    if (!valid || bit_is_set(valid, i)) c = (int32_t)chain_count<K>(slots, mask, log2cap, keys[i], first);
    counts[i] = c < min_count ? min_count : c;
  }
}

__global__ void __launch_bounds__(256) k_add_i32(int32_t* __restrict__ data, int64_t n, int32_t value)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    if (data[i] != NO_MATCH) data[i] += value;
}

constexpr int SEL_CHUNK = 4096;  // rows per chunk of the ordered selection (256 threads x 16 wave-rows of 64)

// bits[i] = (row i has a match) != invert ; null probe rows count as matching iff null_matches
template <typename K>
__global__ void __launch_bounds__(256) k_contains(const K* __restrict__ keys, const uint32_t* __restrict__ valid, int64_t n,
                                                  const Slot<K>* __restrict__ slots, uint32_t log2cap, int invert,
                                                  int null_matches, uint64_t* __restrict__ bits,
                                                  long long* __restrict__ chunk_count)
{
  __shared__ unsigned int s_cnt;
  const uint64_t mask = (1ull << log2cap) - 1;
  const unsigned lane = lane_id();
  const unsigned w    = threadIdx.x / GX_WAVE;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SEL_CHUNK;
  unsigned int mine  = 0;
  for (int k = 0; k < SEL_CHUNK / 256; ++k) {
    const int64_t i = base + (int64_t)(k * 4 + w) * GX_WAVE + lane;  // wave w takes wave-rows w, w+4, ...
    bool sel        = false;
    if (i < n) {
      const bool ok = !valid || bit_is_set(valid, i);
      const bool m  = ok ? chain_first<K>(slots, mask, log2cap, keys[i]) != NO_MATCH : (null_matches != 0);
      sel           = m != (invert != 0);
    }
    const uint64_t b = ballot(sel);
    if (lane == 0 && base + (int64_t)(k * 4 + w) * GX_WAVE < n) {
      bits[(base >> 6) + (k * 4 + w)] = b;
      mine += (unsigned int)__builtin_popcountll(b);
    }
  }
  if (lane == 0 && mine) atomicAdd(&s_cnt, mine);
  __syncthreads();
  if (threadIdx.x == 0) chunk_count[blockIdx.x] = s_cnt;
}

__global__ void __launch_bounds__(256) k_emit_selected(const uint64_t* __restrict__ bits, int64_t n,
                                                       const long long* __restrict__ chunk_start,
                                                       int32_t* __restrict__ out, long long* count_out, int64_t nchunks)
{
  __shared__ unsigned int s_row[SEL_CHUNK / GX_WAVE + 1];
  const int64_t base   = (int64_t)blockIdx.x * SEL_CHUNK;
  const int64_t nwords = div_up(n, (int64_t)GX_WAVE);
  // exclusive popcount scan of the chunk's 64 words (one wave does it)
  if (threadIdx.x < GX_WAVE) {
    const int64_t wi     = (base >> 6) + threadIdx.x;
    const unsigned int c = wi < nwords ? (unsigned int)__builtin_popcountll(bits[wi]) : 0u;
    const unsigned int s = wave_inclusive_scan(c, SumOp());
    s_row[threadIdx.x]   = s - c;
  }
  __syncthreads();
  const unsigned lane      = lane_id();
  const unsigned w         = threadIdx.x / GX_WAVE;
  const long long start    = chunk_start[blockIdx.x];
  for (int k = w; k < SEL_CHUNK / GX_WAVE; k += 256 / GX_WAVE) {
    const int64_t wi = (base >> 6) + k;
    if (wi >= nwords) break;
    const uint64_t b = bits[wi];
    if ((b >> lane) & 1ull) out[start + s_row[k] + __builtin_popcountll(b & lanemask_lt())] = (int32_t)(base + (int64_t)k * GX_WAVE + lane);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && count_out) *count_out = chunk_start[nchunks];
}

template <typename K>
int filter_impl(const void* keys, const uint32_t* valid, int64_t n, const void* table, size_t table_bytes, uint32_t lg,
                int anti, int null_matches, int32_t* out_idx, int64_t* count_dev, void* tmp, size_t* tmp_bytes,
                hipStream_t s)
{
  const int64_t nchunks = div_up(n, (int64_t)SEL_CHUNK);
  Carver c(tmp);
  uint64_t* bits  = c.take<uint64_t>((size_t)div_up(n, (int64_t)GX_WAVE) + 1);
  long long* part = c.take<long long>((size_t)nchunks + 1);
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  const size_t need = sizeof(TableHeader) + (sizeof(Slot<K>) << lg);
  if (table_bytes < need) return GX_ETMP;
  if (n == 0) {
    if (count_dev) GX_HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(int64_t), s));
    return 0;
  }
  const Slot<K>* slots = reinterpret_cast<const Slot<K>*>(static_cast<const char*>(table) + sizeof(TableHeader));
  hipLaunchKernelGGL((k_contains<K>), dim3((unsigned)nchunks), dim3(256), 0, s, static_cast<const K*>(keys), valid, n, slots,
                     lg, anti, null_matches, bits, part);
  hipLaunchKernelGGL((scan::k_partials_scan<long long, SumOp>), dim3(1), dim3(1024), 0, s, part, nchunks, 0ll, SumOp(),
                     (const int*)nullptr);
  hipLaunchKernelGGL(k_emit_selected, dim3((unsigned)nchunks), dim3(256), 0, s, bits, n, part, out_idx,
                     reinterpret_cast<long long*>(count_dev), nchunks);
  GX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Partitioned probe (large probes against tables far beyond the L2s).
// The table is addressed by the TOP bits of key * phi, so its slots [p, p+1) * capacity / P already
// form P sub-tables selected by the top log2(P) hash bits.  P is chosen so that a sub-table is about
// 2 MiB -- half of one XCD's L2.  The probe rows are first radix-partitioned on those bits (one
// streaming pass: LDS-atomic ranking, LDS reorder, space reserved per (XCD range, partition) so the
// short runs of neighbouring tiles merge in one L2), then probed partition by partition with the
// workgroups of XCD x taking the partitions of list x in order: all of them hammer the same 2 MiB
// sub-table at the same time, so after the first touch every probe is an L2 hit instead of a random
// HBM sector read.  Traffic: 8 + 12 (partition) + 12 + 8/match (probe) B/row, all streaming.
// ------------------------------------------------------------------------------------------------
constexpr int PJ_MAXP  = 4096;
constexpr int PJ_BT    = 512;
constexpr int PJ_NR    = 8;               // XCD ranges
constexpr int PJ_CHUNK = 8192;            // probe rows per output reservation

struct alignas(128) PjCounter {
  unsigned int v;
  unsigned int pad[31];
};
struct PjPlan {
  unsigned long long count[PJ_NR][PJ_MAXP];   // rows per (range, partition)
  unsigned long long cursor[PJ_NR][PJ_MAXP];  // scatter cursors
  unsigned long long offset[PJ_MAXP + 1];     // partition starts
  unsigned int chunk0[PJ_MAXP + 1];           // first probe chunk of each partition (partitions in list order)
  unsigned int list_chunk0[PJ_NR + 1];
  PjCounter ticket[PJ_NR];                    // chunk tickets per XCD list, each on its own line
  alignas(128) unsigned int spec_overflow;    // speculative form of the scatter (gx_partition_rows_spec_at): a group outgrew its slot
};

__device__ __forceinline__ unsigned pj_xcc() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u; }
// rows per XCD range of the input: a whole number of scatter tiles, so that no tile straddles two ranges
static inline int64_t pj_range_rows(int64_t n, int64_t tile_rows) { return div_up(n, tile_rows) / PJ_NR * tile_rows; }

// Which partition a key goes to.  TableTop: the top bits of the table hash -- partition p owns slots [p, p+1) << 17 of
// the join table (the partitioned probe / build).  AltHash: top bits of a SECOND multiplicative hash, independent of
// the table's slot bits (the rank a row is sent to in the distributed join: the rows of one rank must still spread
// over the whole local table).  Range: number of splitters <= key in sort order (the distributed sort's exchange).
template <typename K>
struct TableTop {
  int pbits;
  __device__ __forceinline__ unsigned int operator()(K key) const { return (unsigned int)slot_of<K>(key, (uint32_t)pbits); }
};
template <typename K>
struct AltHash {
  unsigned int nparts;  // any count >= 1: the top 32 hash bits scaled into [0, nparts) -- for a power of two these ARE the top bits
  __device__ __forceinline__ unsigned int operator()(K key) const
  {
    const uint32_t h = (uint32_t)(((uint64_t)key * 0xD6E8FEB86659FD93ull) >> 32);
    return (unsigned int)(((uint64_t)h * nparts) >> 32);
  }
};
constexpr int PJ_MAX_SPLIT = 15;
template <typename K, int KIND>
struct RangeSplit {
  K split[PJ_MAX_SPLIT];  // sortable form, ascending
  int nsplit;
  __device__ __forceinline__ unsigned int operator()(K key) const
  {
    const K sk     = to_sortable<K, KIND>(key, K(0));
    unsigned int d = 0;
#pragma unroll
    for (int i = 0; i < PJ_MAX_SPLIT; ++i) d += (i < nsplit && split[i] <= sk) ? 1u : 0u;
    return d;
  }
};

template <typename K, typename F>
__global__ void __launch_bounds__(256) k_pj_hist(const K* __restrict__ keys, int64_t n, PjPlan* plan, int pbits, int64_t rrows, F part_of,
                                                 const unsigned int* gate = nullptr)
{
  if (gate && *gate == 0) return;  // the exact sequence behind a speculative partition pass that held
  __shared__ unsigned int s_h[PJ_MAXP];
  const int P = 1 << pbits;
  for (int i = threadIdx.x; i < P; i += 256) s_h[i] = 0;
  __syncthreads();
  const int r          = blockIdx.x % PJ_NR;
  const int64_t jb     = blockIdx.x / PJ_NR;
  const int64_t nb     = gridDim.x / PJ_NR;
  const int64_t rbegin = (int64_t)r * rrows;
  const int64_t rend   = (r == PJ_NR - 1) ? n : rbegin + rrows;
  constexpr int U      = 8;
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
      if (i < rend) atomicAdd(&s_h[part_of(k[u])], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < P; i += 256) {
    const unsigned int c = s_h[i];
    if (c) atomicAdd(&plan->count[r][i], (unsigned long long)c);
  }
}

// one block of 1024 threads: offsets, cursors, probe-chunk numbering (partitions of list x = [x, x+1) * P / 8)
__global__ void __launch_bounds__(1024) k_pj_offsets(PjPlan* plan, int pbits, unsigned int chunk_rows, const unsigned int* gate = nullptr)
{
  if (gate && *gate == 0) return;
  __shared__ unsigned long long s_tmp[1024 / GX_WAVE + 1];
  __shared__ unsigned long long s_carry;
  __shared__ unsigned int s_ccarry;
  const int P = 1 << pbits;
  if (threadIdx.x == 0) {
    s_carry  = 0;
    s_ccarry = 0;
  }
  __syncthreads();
  for (int base = 0; base < P; base += 1024) {
    const int p = base + threadIdx.x;
    unsigned long long c = 0;
    if (p < P)
      for (int r = 0; r < PJ_NR; ++r) c += plan->count[r][p];
    unsigned long long total;
    unsigned long long run = block_exclusive_scan<1024>(c, 0ull, SumOp(), s_tmp, &total) + s_carry;
    const unsigned int nch = (unsigned int)((c + chunk_rows - 1) / chunk_rows);
    unsigned long long ctotal;
    const unsigned int ch0 = (unsigned int)block_exclusive_scan<1024>((unsigned long long)nch, 0ull, SumOp(), s_tmp, &ctotal) + s_ccarry;
    if (p < P) {
      plan->offset[p] = run;
      plan->chunk0[p] = ch0;
      if (p % (P / PJ_NR) == 0) plan->list_chunk0[p / (P / PJ_NR)] = ch0;
      for (int r = 0; r < PJ_NR; ++r) {
        plan->cursor[r][p] = run;
        run += plan->count[r][p];
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      s_carry += total;
      s_ccarry += (unsigned int)ctotal;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    plan->offset[P]           = s_carry;
    plan->chunk0[P]           = s_ccarry;
    plan->list_chunk0[PJ_NR]  = s_ccarry;
  }
}

// Scatter of the partition pass.  A workgroup of BTt threads ranks a tile of BTt * RPT rows with LDS atomics,
// reorders the keys in LDS so that every partition leaves as one contiguous run, then sends the row indices
// through the same buffer.  The partition of an element is recomputed from its key at write-out (one 64-bit
// multiply) instead of being staged: the whole LDS budget goes to rows, and a tile of 16384 rows gives runs
// of 8 rows (64 B of keys) at P = 2048 where the 4096-row tile of round 1 gave 2.
template <typename K, int RPT, int BTt, typename F>
__global__ void __launch_bounds__(BTt) k_pj_scatter(const K* __restrict__ keys, int64_t n, PjPlan* plan, int pbits,
                                                    int64_t rrows, K* __restrict__ pkeys, int32_t* __restrict__ pidx, F part_of,
                                                    int32_t row0 = 0, uint32_t spec_cap = 0, const int32_t* __restrict__ payload = nullptr)
{
  // payload != NULL: a row travels with payload[row] instead of its row number
  // spec_cap > 0 (gx_partition_rows_spec_at): no histogram ran -- group b owns the slot [b, b + 1) * spec_cap of the output, ONE
  // fill counter per group (cursor[0][b], from zero); rows beyond a slot are dropped and flagged (the caller re-partitions)
  constexpr int TILE = BTt * RPT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  K* s_k                = reinterpret_cast<K*>(smem);                                  // TILE (reused for idx)
  unsigned int* s_cnt   = reinterpret_cast<unsigned int*>(smem + (size_t)TILE * sizeof(K));  // P
  unsigned int* s_start = s_cnt + (1 << pbits);                                        // P
  unsigned int* s_delta = s_start + (1 << pbits);                                      // P: global position - tile position (mod 2^32)
  __shared__ unsigned int s_scan[BTt / GX_WAVE + 1];
  __shared__ unsigned int s_carry;

  const int P         = 1 << pbits;
  const unsigned tid  = threadIdx.x;
  const int64_t tile  = xcd_swizzle(blockIdx.x, gridDim.x);
  const int64_t base  = tile * TILE;
  const int range     = spec_cap ? 0 : ((rrows > 0 && base / rrows < PJ_NR - 1) ? (int)(base / rrows) : PJ_NR - 1);
  const int nvalid    = (int)((n - base < (int64_t)TILE) ? (n - base) : (int64_t)TILE);
  for (int i = tid; i < P; i += BTt) s_cnt[i] = 0;
  if (tid == 0) s_carry = 0;
  K key[RPT];
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int idx = j * BTt + (int)tid;
    key[j]        = (idx < nvalid) ? __builtin_nontemporal_load(&keys[base + idx]) : K(0);
  }
  __syncthreads();
  unsigned int packed[RPT];  // partition << 16 | rank inside (tile, partition)
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int idx           = j * BTt + (int)tid;
    const unsigned int part = part_of(key[j]);
    // few bins (the exchange partition: <= 16 destination ranks): wave-aggregated ranks, see lds_rank_few
    const unsigned int rank = pbits <= 4 ? lds_rank_few<4>(s_cnt, part, idx < nvalid) : ((idx < nvalid) ? atomicAdd(&s_cnt[part], 1u) : 0u);
    packed[j]               = (part << 16) | rank;
  }
  __syncthreads();
  // exclusive scan of the P counts (P may exceed the block: strips of BTt)
  for (int b0 = 0; b0 < P; b0 += BTt) {
    const int b          = b0 + (int)tid;
    const unsigned int c = b < P ? s_cnt[b] : 0u;
    unsigned int total;
    const unsigned int st = block_exclusive_scan<BTt>(c, 0u, SumOp(), s_scan, &total) + s_carry;
    if (b < P) {
      s_start[b] = st;
      unsigned long long g = 0;
      if (c) g = atomicAdd(&plan->cursor[range][b], (unsigned long long)c);
      if (spec_cap) {
        if (c && g + c > spec_cap) plan->spec_overflow = 1u;
        g += (unsigned long long)b * spec_cap;
      }
      s_delta[b] = (unsigned int)g - st;
    }
    __syncthreads();
    if (tid == 0) s_carry += total;
    __syncthreads();
  }
  // keys through LDS
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int idx = j * BTt + (int)tid;
    if (idx < nvalid) s_k[s_start[packed[j] >> 16] + (packed[j] & 0xFFFFu)] = key[j];
  }
  __syncthreads();
  unsigned short obin[RPT];  // partition of the element this thread writes out
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int i = j * BTt + (int)tid;
    obin[j]     = 0;
    if (i < nvalid) {
      const K k = s_k[i];
      obin[j]   = (unsigned short)part_of(k);
      const unsigned int p = s_delta[obin[j]] + (unsigned int)i;
      if (spec_cap && p >= ((unsigned int)obin[j] + 1u) * spec_cap) obin[j] = 0xFFFFu;  // beyond the slot: dropped
      else pkeys[p] = k;
    }
  }
  if (pidx == nullptr) return;  // keys only (range partition of a sort)
  __syncthreads();
  // row indices through the same buffer
  int32_t* s_i = reinterpret_cast<int32_t*>(smem);
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int idx = j * BTt + (int)tid;
    if (idx < nvalid) s_i[s_start[packed[j] >> 16] + (packed[j] & 0xFFFFu)] = payload ? payload[base + idx] : (int32_t)(base + idx) + row0;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int i = j * BTt + (int)tid;
    if (i < nvalid && obin[j] != 0xFFFFu) pidx[(unsigned int)(s_delta[obin[j]] + (unsigned int)i)] = s_i[i];
  }
}

template <typename K>
__global__ void __launch_bounds__(PJ_BT) k_pj_probe(const K* __restrict__ pkeys, const int32_t* __restrict__ pidx,
                                                    PjPlan* plan, int pbits, const Slot<K>* __restrict__ slots,
                                                    uint32_t log2cap, int left_outer, int32_t* __restrict__ out_probe,
                                                    int32_t* __restrict__ out_build, int64_t capacity,
                                                    unsigned long long* cursor)
{
  constexpr int NWJ  = PJ_BT / GX_WAVE;
  constexpr int RPT  = PJ_CHUNK / PJ_BT;  // 16
  __shared__ unsigned long long s_wave_tot[NWJ];
  __shared__ unsigned long long s_base;
  __shared__ unsigned int s_misc[4];
  const int P        = 1 << pbits;
  const int LISTP    = P / PJ_NR;
  const uint64_t mask = (1ull << log2cap) - 1;
  const unsigned tid  = threadIdx.x;
  const unsigned lane = lane_id();
  const unsigned w    = tid / GX_WAVE;

  // ---- take a chunk: own XCD's list first
  if (tid == 0) {
    const unsigned x = pj_xcc();
    unsigned int g   = 0xFFFFFFFFu;
    for (int i = 0; i < PJ_NR; ++i) {
      const unsigned y       = (x + i) % PJ_NR;
      const unsigned int nch = plan->list_chunk0[y + 1] - plan->list_chunk0[y];
      if (nch == 0) continue;
      const unsigned int t = atomicAdd(&plan->ticket[y].v, 1u);
      if (t < nch) {
        g         = plan->list_chunk0[y] + t;
        s_misc[1] = y;
        break;
      }
    }
    s_misc[0] = g;
  }
  __syncthreads();
  const unsigned int g = s_misc[0];
  if (g == 0xFFFFFFFFu) return;
  {  // partition of chunk g inside its list: one table entry per thread (LISTP <= 512)
    const unsigned y = s_misc[1];
    for (int e = (int)tid; e < LISTP; e += PJ_BT) {
      const int p           = (int)y * LISTP + e;
      const unsigned int lo = plan->chunk0[p], hi = plan->chunk0[p + 1];
      if (lo <= g && g < hi) {
        s_misc[2] = (unsigned int)p;
        s_misc[3] = g - lo;
      }
    }
  }
  __syncthreads();
  const unsigned int part = s_misc[2];
  const unsigned long long p0 = plan->offset[part], p1 = plan->offset[part + 1];
  const unsigned long long c0 = p0 + (unsigned long long)s_misc[3] * PJ_CHUNK;
  const unsigned long long c1 = c0 + PJ_CHUNK < p1 ? c0 + PJ_CHUNK : p1;

  const unsigned long long wbase = c0 + (unsigned long long)w * (RPT * GX_WAVE) + lane;
  uint32_t cnt[RPT];
  int32_t first[RPT], row[RPT];
  uint32_t wave_total = 0;
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const unsigned long long i = wbase + (unsigned long long)j * GX_WAVE;
    cnt[j]   = 0;
    first[j] = NO_MATCH;
    row[j]   = 0;
    if (i < c1) {
      row[j] = __builtin_nontemporal_load(&pidx[i]);
      cnt[j] = chain_count<K>(slots, mask, log2cap, __builtin_nontemporal_load(&pkeys[i]), first[j]);
      if (left_outer && cnt[j] == 0) cnt[j] = 1;
    }
  }
  uint32_t off[RPT];
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    uint32_t inc;
    if (ballot(cnt[j] > 1) == 0) {
      const uint64_t b = ballot(cnt[j] == 1);
      off[j]           = wave_total + (uint32_t)__builtin_popcountll(b & lanemask_lt());
      inc              = (uint32_t)__builtin_popcountll(b);
    } else {
      const uint32_t sc = wave_inclusive_scan(cnt[j], SumOp());
      off[j]            = wave_total + sc - cnt[j];
      inc               = shfl(sc, GX_WAVE - 1);
    }
    wave_total += inc;
  }
  if (lane == 0) s_wave_tot[w] = wave_total;
  __syncthreads();
  if (tid == 0) {
    unsigned long long tot = 0;
    for (int k = 0; k < NWJ; ++k) {
      const unsigned long long t = s_wave_tot[k];
      s_wave_tot[k]              = tot;
      tot += t;
    }
    s_base = tot ? atomicAdd(cursor, tot) : 0ull;
  }
  __syncthreads();
  const unsigned long long wave_base = s_base + s_wave_tot[w];
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    if (cnt[j] == 0) continue;
    unsigned long long pos = wave_base + off[j];
    if (cnt[j] == 1) {
      if ((int64_t)pos < capacity) {
        __builtin_nontemporal_store(row[j], &out_probe[pos]);
        __builtin_nontemporal_store(first[j], &out_build[pos]);
      }
    } else {
      const unsigned long long i = wbase + (unsigned long long)j * GX_WAVE;
      const K key = pkeys[i];
      uint64_t h  = slot_of<K>(key, log2cap);
      for (;;) {
        K k;
        int32_t r;
        load_slot<K>(&slots[h], k, r);
        if (r == EMPTY_ROW) break;
        if (k == key) {
          if ((int64_t)pos < capacity) {
            out_probe[pos] = row[j];
            out_build[pos] = r;
          }
          ++pos;
        }
        h = (h + 1) & mask;
      }
    }
  }
}

// ---- the partitioned probe on LDS tags ------------------------------------------------------------
// The 8 tags of local slots [li, li+8) of the chain (one ds_read2_b32 + v_alignbit): `cand` gets a
// flag at the top bit of every nibble that carries `tagpat`'s tag and lies before the first empty
// slot; returns true when an empty slot ends the chain inside the window.  Nibble-zero detection is
// the carry-free form ((x & 7..7) + 7..7 | x), exact for every nibble.
__device__ __forceinline__ bool scan_tags8(const uint32_t* s_tagw, uint32_t li, uint32_t tagpat, uint32_t& cand)
{
  const uint32_t w0 = s_tagw[li >> 3], w1 = s_tagw[(li >> 3) + 1];
  const uint32_t x  = __builtin_amdgcn_alignbit(w1, w0, (li & 7u) * 4u);
  const uint32_t y  = x ^ tagpat;
  const uint32_t z  = ~(((x & 0x77777777u) + 0x77777777u) | x) & 0x88888888u;  // empty slots
  const uint32_t m  = ~(((y & 0x77777777u) + 0x77777777u) | y) & 0x88888888u;  // tag matches
  cand              = m & ((z & (0u - z)) - 1u);                                 // ... below the first empty one
  return z != 0;
}

// A workgroup takes `chunk_rows` (a multiple of PJ_CHUNK) rows of ONE partition, copies the 64 KiB of
// tags of that partition's sub-table into LDS once, and probes PJ_CHUNK rows at a time: (A) every
// lane runs its rows' chains on the LDS tags up to the first tag match, (B) the candidate slots of
// all its rows are fetched together -- independent L2 requests in flight instead of one dependent
// request per chain step -- (C) keys are compared and the chains resume (duplicates, tag collisions).
// A probe row that misses usually costs no L2 request at all.
template <typename K>
__global__ void __launch_bounds__(PJ_BT, 4)
k_pj_probe_tags(const K* __restrict__ pkeys, const int32_t* __restrict__ pidx, PjPlan* plan, int pbits,
                const Slot<K>* __restrict__ slots, uint32_t log2cap, int left_outer, int32_t* __restrict__ out_probe,
                int32_t* __restrict__ out_build, int64_t capacity, unsigned long long* cursor, unsigned int chunk_rows)
{
  constexpr int NWJ      = PJ_BT / GX_WAVE;
  constexpr int RPT      = PJ_CHUNK / PJ_BT;  // 16
  constexpr int HB       = 8;                 // rows whose slot fetches are batched
  constexpr uint32_t SUB = 1u << PJ_SUB_LOG2;
  __shared__ unsigned long long s_wave_tot[NWJ];
  __shared__ unsigned long long s_base;
  __shared__ unsigned int s_misc[4];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint8_t* s_tags        = reinterpret_cast<uint8_t*>(smem);
  const uint32_t* s_tagw = reinterpret_cast<const uint32_t*>(smem);
  const int P         = 1 << pbits;
  const int LISTP     = P / PJ_NR;
  const uint64_t mask = (1ull << log2cap) - 1;
  const unsigned tid  = threadIdx.x;
  const unsigned lane = lane_id();
  const unsigned w    = tid / GX_WAVE;

  if (tid == 0) {  // take a chunk: own XCD's list first
    const unsigned x = pj_xcc();
    unsigned int g   = 0xFFFFFFFFu;
    for (int i = 0; i < PJ_NR; ++i) {
      const unsigned y       = (x + i) % PJ_NR;
      const unsigned int nch = plan->list_chunk0[y + 1] - plan->list_chunk0[y];
      if (nch == 0) continue;
      const unsigned int t = atomicAdd(&plan->ticket[y].v, 1u);
      if (t < nch) {
        g         = plan->list_chunk0[y] + t;
        s_misc[1] = y;
        break;
      }
    }
    s_misc[0] = g;
  }
  __syncthreads();
  const unsigned int g = s_misc[0];
  if (g == 0xFFFFFFFFu) return;
  {
    const unsigned y = s_misc[1];
    for (int e = (int)tid; e < LISTP; e += PJ_BT) {
      const int p           = (int)y * LISTP + e;
      const unsigned int lo = plan->chunk0[p], hi = plan->chunk0[p + 1];
      if (lo <= g && g < hi) {
        s_misc[2] = (unsigned int)p;
        s_misc[3] = g - lo;
      }
    }
  }
  __syncthreads();
  const unsigned int part     = s_misc[2];
  const unsigned long long p0 = plan->offset[part], p1 = plan->offset[part + 1];
  const unsigned long long c0 = p0 + (unsigned long long)s_misc[3] * chunk_rows;
  const unsigned long long c1 = c0 + chunk_rows < p1 ? c0 + chunk_rows : p1;
  const uint64_t sub_base     = (uint64_t)part << PJ_SUB_LOG2;
  {
    const uint8_t* gtags = reinterpret_cast<const uint8_t*>(slots + (mask + 1)) + (sub_base >> 1);
    const uint4* src     = reinterpret_cast<const uint4*>(gtags);
    uint4* dst           = reinterpret_cast<uint4*>(s_tags);
    for (uint32_t i = tid; i < SUB / 2 / 16; i += PJ_BT) dst[i] = src[i];
  }
  __syncthreads();

  for (unsigned long long sc0 = c0; sc0 < c1; sc0 += PJ_CHUNK) {
    const unsigned long long sc1   = sc0 + PJ_CHUNK < c1 ? sc0 + PJ_CHUNK : c1;
    const unsigned long long wbase = sc0 + (unsigned long long)w * (RPT * GX_WAVE) + lane;
    uint32_t cnt[RPT];
    int32_t first[RPT];
#pragma unroll
    for (int h = 0; h < RPT; h += HB) {
      K key[HB];
      uint32_t li[HB], cand[HB];
      uint32_t active = 0, ended = 0;
      // finish a chain on the slots themselves (it leaves the sub-table: a handful of rows per partition)
      auto finish_global = [&](int j, uint64_t gs) {
        for (;;) {
          K k;
          int32_t r;
          load_slot<K>(&slots[gs & mask], k, r);
          if (r == EMPTY_ROW) break;
          if (k == key[j]) {
            if (cnt[h + j] == 0) first[h + j] = r;
            ++cnt[h + j];
          }
          ++gs;
        }
      };
#pragma unroll
      for (int j = 0; j < HB; ++j) {
        const unsigned long long i = wbase + (unsigned long long)(h + j) * GX_WAVE;
        cnt[h + j]   = 0;
        first[h + j] = NO_MATCH;
        key[j]       = K(0);
        li[j]        = 0;
        cand[j]      = 0;
        if (i < sc1) {
          key[j]              = __builtin_nontemporal_load(&pkeys[i]);
          const uint64_t prod = (uint64_t)key[j] * 0x9E3779B97F4A7C15ull;
          li[j]               = (uint32_t)((prod >> (64 - log2cap)) - sub_base);
          uint32_t t          = (uint32_t)(prod >> (60 - log2cap)) & 15u;
          t                   = t ? t : 8u;
          if (li[j] > SUB - 8) {
            finish_global(j, sub_base + li[j]);
          } else {
            const bool e = scan_tags8(s_tagw, li[j], t * 0x11111111u, cand[j]);
            if (e) ended |= 1u << j;
            if (cand[j] || !e) active |= 1u << j;
          }
        }
      }
      while (active) {
        K k[HB];
        int32_t r[HB];
#pragma unroll
        for (int j = 0; j < HB; ++j) {  // candidate slots of all rows, in flight together
          k[j] = K(0);
          r[j] = EMPTY_ROW;
          if (ballot((active >> j) & 1u) == 0) continue;  // wave-uniform: nobody left on row j
          if ((active & (1u << j)) && cand[j]) {
            load_slot<K>(&slots[sub_base + li[j] + ((uint32_t)__builtin_ctz(cand[j]) >> 2)], k[j], r[j]);
          }
        }
#pragma unroll
        for (int j = 0; j < HB; ++j) {
          if (ballot((active >> j) & 1u) == 0) continue;
          if (!(active & (1u << j))) continue;
          if (cand[j]) {
            if (k[j] == key[j]) {
              if (cnt[h + j] == 0) first[h + j] = r[j];
              ++cnt[h + j];
            }
            cand[j] &= cand[j] - 1;
          }
          if (cand[j] == 0) {
            if (ended & (1u << j)) {
              active &= ~(1u << j);
            } else {  // the chain runs on: next 8 slots
              li[j] += 8;
              if (li[j] > SUB - 8) {
                finish_global(j, sub_base + li[j]);
                active &= ~(1u << j);
              } else {
                const bool e = scan_tags8(s_tagw, li[j], tag_of<K>(key[j], log2cap) * 0x11111111u, cand[j]);
                if (e) {
                  ended |= 1u << j;
                  if (cand[j] == 0) active &= ~(1u << j);
                }
              }
            }
          }
        }
      }
    }
    uint32_t off[RPT];
    uint32_t wave_total = 0;
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const unsigned long long i = wbase + (unsigned long long)j * GX_WAVE;
      if (left_outer && cnt[j] == 0 && i < sc1) cnt[j] = 1;
      uint32_t inc;
      if (ballot(cnt[j] > 1) == 0) {
        const uint64_t b = ballot(cnt[j] == 1);
        off[j]           = wave_total + (uint32_t)__builtin_popcountll(b & lanemask_lt());
        inc              = (uint32_t)__builtin_popcountll(b);
      } else {
        const uint32_t sc = wave_inclusive_scan(cnt[j], SumOp());
        off[j]            = wave_total + sc - cnt[j];
        inc               = shfl(sc, GX_WAVE - 1);
      }
      wave_total += inc;
    }
    if (lane == 0) s_wave_tot[w] = wave_total;
    __syncthreads();
    if (tid == 0) {
      unsigned long long tot = 0;
      for (int k = 0; k < NWJ; ++k) {
        const unsigned long long t = s_wave_tot[k];
        s_wave_tot[k]              = tot;
        tot += t;
      }
      s_base = tot ? atomicAdd(cursor, tot) : 0ull;  // one reservation per PJ_CHUNK rows (per wave: measured 30 % slower)
    }
    __syncthreads();
    const unsigned long long wave_base = s_base + s_wave_tot[w];
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      if (cnt[j] == 0) continue;
      const unsigned long long i = wbase + (unsigned long long)j * GX_WAVE;
      unsigned long long pos     = wave_base + off[j];
      const int32_t row          = __builtin_nontemporal_load(&pidx[i]);
      if (cnt[j] == 1) {
        if ((int64_t)pos < capacity) {
          __builtin_nontemporal_store(row, &out_probe[pos]);
          __builtin_nontemporal_store(first[j], &out_build[pos]);
        }
      } else {
        const K key = pkeys[i];
        uint64_t hh = slot_of<K>(key, log2cap);
        for (;;) {
          K k;
          int32_t r;
          load_slot<K>(&slots[hh], k, r);
          if (r == EMPTY_ROW) break;
          if (k == key) {
            if ((int64_t)pos < capacity) {
              out_probe[pos] = row;
              out_build[pos] = r;
            }
            ++pos;
          }
          hh = (hh + 1) & mask;
        }
      }
    }
    __syncthreads();  // s_wave_tot / s_base are reused by the next PJ_CHUNK rows
  }
}

// ---- software-pipelined tag probe ---------------------------------------------------------------------
// k_pj_probe_tags waits for every memory level in turn (keys from HBM, then slots from L2, then the output
// cursor, then the row ids from HBM) with 16 waves per CU to hide it: 85 % of its wave cycles were parked on
// s_waitcnt (profiles/r1_run28_pmc_join_sq.txt).  Here ONE persistent 1024-thread workgroup per CU keeps 64 KiB
// of tags in LDS, the rest of the LDS stages the output, and a trip of the loop runs three stages of three
// different 3840-row PIECES on 15 probe waves:
//   S1(t)   issue the (key, row id) loads of piece t                           -> consumed one trip later
//   S2(t-1) keys arrived: run the chain heads on the LDS tags, issue the slot load of the first candidate
//           (unconditionally, lanes without a candidate read one shared dummy slot: a conditional load would
//           make the compiler wait for it on the spot)
//   S3(t-2) slots arrived: compare, finish the (rare) longer chains on the tags in global memory, append the
//           matches to LDS staging buffer (t-2) % 3 at positions handed out by an LDS counter;
//   after the trip's ONE barrier the probe waves send the staged pairs of piece t-3 to HBM as two fully
//   coalesced streams, while the 16th wave -- which probes nothing -- reserves the output of piece t-2 with one
//   device atomic, takes the ticket of piece t+2 and resolves it to (partition, rows), all for later trips.
// Nothing a probe wave does in a trip waits for a memory operation issued in the same trip, except the rare
// chain continuations.
// Pieces are handed out per XCD list in partition order, so the 32 workgroups of an XCD work on the SAME
// partition (or the next one) at any time: its 2 MiB of slots stay in that XCD's L2.  (With 131072-row chunks,
// the first version of this kernel, the workgroups of an XCD were spread over ~8 partitions = 16 MiB of slots:
// 347 M L2 misses, 40 GB fetched for 12 GB of input -- profiles/r2_run6_pmc_join_traffic.txt.)  A workgroup
// reloads its tags when its next piece belongs to another partition; the pipeline never drains in between.
constexpr int PP_BT    = 1024;
constexpr int PP_PW    = PP_BT / GX_WAVE - 1;       // 15 probe waves
constexpr int PP_R     = 4;                         // rows per thread and piece
constexpr int PP_ROWS  = PP_PW * GX_WAVE * PP_R;    // 3840 rows per piece; also the capacity of a staging buffer
struct Raw3 {  // the 12 bytes of a 16-B slot that matter: a 4th dword in flight would be a register the compiler may reuse early
  uint32_t x, y, z;
};
template <typename K> struct SlotRaw;
template <> struct SlotRaw<uint64_t> { typedef Raw3 type; };
template <> struct SlotRaw<uint32_t> { typedef uint2 type; };
__device__ __forceinline__ void unpack_slot(const Raw3& v, uint64_t& key, int32_t& row)
{
  key = ((uint64_t)v.y << 32) | v.x;
  row = (int32_t)v.z;
}
__device__ __forceinline__ void unpack_slot(const uint2& v, uint32_t& key, int32_t& row)
{
  key = v.x;
  row = (int32_t)v.y;
}
struct PpPiece {
  unsigned long long c0, c1;  // rows [c0, c1) of the partitioned probe arrays
  unsigned int part;          // their partition
  unsigned int valid;         // 0: no piece left -- the pipeline drains
};
// scan_tags8 on the tags in GLOBAL memory (chain continuations): `tagw` = the sub-table's tag words
__device__ __forceinline__ bool scan_tags8_global(const uint32_t* tagw, uint32_t li, uint32_t tagpat, uint32_t& cand)
{
  const uint32_t w0 = tagw[li >> 3];
  const uint32_t w1 = (li & 7u) ? tagw[(li >> 3) + 1] : 0u;  // unused when the window is word aligned (also the table's last word)
  const uint32_t x  = __builtin_amdgcn_alignbit(w1, w0, (li & 7u) * 4u);
  const uint32_t y  = x ^ tagpat;
  const uint32_t z  = ~(((x & 0x77777777u) + 0x77777777u) | x) & 0x88888888u;
  const uint32_t m  = ~(((y & 0x77777777u) + 0x77777777u) | y) & 0x88888888u;
  cand              = m & ((z & (0u - z)) - 1u);
  return z != 0;
}

template <typename K>
__global__ void __launch_bounds__(PP_BT)
k_pj_probe_pipe(const K* __restrict__ pkeys, const int32_t* __restrict__ pidx, PjPlan* plan, int pbits,
                const Slot<K>* __restrict__ slots, uint32_t log2cap, int left_outer, int32_t* __restrict__ out_probe,
                int32_t* __restrict__ out_build, int64_t capacity, unsigned long long* cursor)
{
  typedef typename SlotRaw<K>::type Raw;
  constexpr uint32_t SUB = 1u << PJ_SUB_LOG2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t* s_tagw = reinterpret_cast<const uint32_t*>(smem);                       // 64 KiB of tags
  int32_t* s_sidx        = reinterpret_cast<int32_t*>(smem + (SUB >> 1));                  // [3][PP_ROWS] staged probe rows
  int32_t* s_sfirst      = s_sidx + 3 * PP_ROWS;                                           // [3][PP_ROWS] staged build rows
  __shared__ unsigned int s_cnt[8];           // pairs staged by piece (it & 7)
  __shared__ unsigned long long s_base[3];    // output position of staging buffer (it % 3)
  __shared__ PpPiece s_piece[4];              // piece of trip (t & 3), resolved two trips ahead by the service wave
  const int P         = 1 << pbits;
  const int LISTP     = P / PJ_NR;
  const uint64_t mask = (1ull << log2cap) - 1;
  const unsigned tid  = threadIdx.x;
  const unsigned lane = lane_id();
  const unsigned w    = tid / GX_WAVE;
  const uint8_t* gtags = reinterpret_cast<const uint8_t*>(slots + (mask + 1));

  if (tid < 8) s_cnt[tid] = 0;

  if (w == PP_PW) {
    // ------------------------------------------------------------------ service wave: tickets and reservations
    // tickets: own XCD's list first, then the others; a list that has run dry stays dry
    const unsigned x0 = pj_xcc();
    unsigned ylist    = 0;  // lists tried so far
    auto take_piece = [&](PpPiece& pc) {
      pc.valid = 0;
      pc.c0 = pc.c1 = 0;
      pc.part = 0;
      unsigned int g = 0xFFFFFFFFu, y = 0;
      if (lane == 0) {
        while (ylist < PJ_NR) {
          y                      = (x0 + ylist) % PJ_NR;
          const unsigned int nch = plan->list_chunk0[y + 1] - plan->list_chunk0[y];
          if (nch) {
            const unsigned int t = atomicAdd(&plan->ticket[y].v, 1u);
            if (t < nch) {
              g = plan->list_chunk0[y] + t;
              break;
            }
          }
          ++ylist;
        }
      }
      g     = (unsigned int)__builtin_amdgcn_readfirstlane((int)g);
      y     = (unsigned int)__builtin_amdgcn_readfirstlane((int)y);
      ylist = (unsigned int)__builtin_amdgcn_readfirstlane((int)ylist);
      if (g == 0xFFFFFFFFu) return;
      // partition of piece g inside list y: the one whose piece interval contains it (LISTP <= 512 entries, 64 lanes)
      unsigned int part = 0, loc = 0;
      bool hit = false;
      for (int e = (int)lane; e < LISTP; e += GX_WAVE) {
        const int p           = (int)y * LISTP + e;
        const unsigned int lo = plan->chunk0[p], hi = plan->chunk0[p + 1];
        if (lo <= g && g < hi) {
          part = (unsigned int)p;
          loc  = g - lo;
          hit  = true;
        }
      }
      const uint64_t hb = ballot(hit);
      const int src     = __builtin_ctzll(hb);
      part              = shfl(part, src);
      loc               = shfl(loc, src);
      const unsigned long long p0 = plan->offset[part], p1 = plan->offset[part + 1];
      pc.c0    = p0 + (unsigned long long)loc * PP_ROWS;
      pc.c1    = pc.c0 + PP_ROWS < p1 ? pc.c0 + PP_ROWS : p1;
      pc.part  = part;
      pc.valid = 1;
    };
    PpPiece pc;
    take_piece(pc);
    if (lane == 0) s_piece[0] = pc;
    take_piece(pc);
    if (lane == 0) s_piece[1] = pc;
    __syncthreads();  // prologue barrier
    unsigned long long pending = 0;
    int done_at = -1;
    unsigned int partC = 0;
    bool tags_loaded   = false;
    for (int t = 0;; ++t) {
      const int itf = t - 3, it3 = t - 2;
      if (t >= 1) {  // R(t): the probe waves reload their tags when the piece entering S2 belongs to another partition
        const PpPiece pa = s_piece[(t - 1) & 3];
        if (pa.valid && (!tags_loaded || pa.part != partC)) {
          partC       = pa.part;
          tags_loaded = true;
          __syncthreads();
        }
      }
      if (done_at < 0 && !s_piece[t & 3].valid) done_at = t;  // same test as the probe waves make
      if (lane == 0 && itf >= 0) s_base[(unsigned)itf % 3u] = pending;
      // piece t + 2, for the S1 of two trips from now (its slot in the ring was read last in trip t - 2)
      take_piece(pc);
      if (lane == 0) s_piece[(t + 2) & 3] = pc;
      __syncthreads();  // X(t)
      if (done_at >= 0 && t >= done_at + 3) break;
      if (lane == 0) {
        if (it3 >= 0) {
          unsigned int c = s_cnt[(unsigned)it3 & 7u];
          c              = c < (unsigned)PP_ROWS ? c : (unsigned)PP_ROWS;
          pending        = c ? atomicAdd(cursor, (unsigned long long)c) : 0ull;
        }
        s_cnt[(unsigned)(t + 2) & 7u] = 0;  // counter of piece t+2 (staged in trip t+4); its last user, piece t-6, left in trip t-3
      }
    }
    return;
  }

  // ---------------------------------------------------------------------- probe waves
  __syncthreads();  // prologue barrier: pieces 0 and 1 are resolved
  K kA[PP_R];                // S1 -> S2
  int32_t iA[PP_R];
  K kB[PP_R];                // S2 -> S3
  int32_t iB[PP_R];
  uint32_t li[PP_R], cand[PP_R];
  Raw sv[PP_R];              // first candidate slot, in flight from S2 to S3
  uint32_t fl = 0;           // per row j: bit j = live row, bit 4+j = chain ended inside the scanned window, bit 8+j = chain left the LDS window
  uint32_t partA = 0, partB = 0, partC = 0;     // partition of the piece in S1->S2, S2->S3 (this trip's S3), and of the LDS tags
  bool tags_loaded = false;
  unsigned long long cA0 = 0, cA1 = 0;          // row range of the piece loaded by the last S1
  bool validA = false, validB = false;          // a piece sits between S1 and S2 / between S2 and S3
#pragma unroll
  for (int j = 0; j < PP_R; ++j) {
    kA[j] = kB[j] = K(0);
    iA[j] = iB[j] = 0;
    li[j] = cand[j] = 0;
    sv[j] = Raw{};
  }
  int done_at = -1;

  for (int t = 0;; ++t) {
    // ---------------- S3(t-2): compare, finish chains, stage the matches
    const int it3 = t - 2;
    if (validB) {
      const unsigned buf       = (unsigned)it3 % 3u;
      const uint64_t sub_base  = (uint64_t)partB << PJ_SUB_LOG2;
      const uint32_t* gtagw    = reinterpret_cast<const uint32_t*>(gtags + (sub_base >> 1));
      uint32_t m[PP_R];
      int32_t first[PP_R];
      K sk[PP_R];
      int32_t sr[PP_R];
      uint32_t active = 0, ended = (fl >> 4) & 15u;
#pragma unroll
      for (int j = 0; j < PP_R; ++j) {
        m[j]     = 0;
        first[j] = NO_MATCH;
        unpack_slot(sv[j], sk[j], sr[j]);
        if (!((fl >> j) & 1u)) continue;
        if ((fl >> (8 + j)) & 1u) {  // chain starts in the last slots of the sub-table: walk the slots themselves
          uint64_t gs = sub_base + li[j];
          for (;;) {
            K k;
            int32_t r;
            load_slot<K>(&slots[gs & mask], k, r);
            if (r == EMPTY_ROW) break;
            if (k == kB[j]) {
              if (m[j] == 0) first[j] = r;
              ++m[j];
            }
            ++gs;
          }
        } else if (cand[j] || !((ended >> j) & 1u)) {
          active |= 1u << j;
        }
      }
      bool preloaded = true;  // the first candidate of every row was fetched by S2
      while (active) {
        if (!preloaded) {
#pragma unroll
          for (int j = 0; j < PP_R; ++j) {
            if (ballot((active >> j) & 1u) == 0) continue;
            if ((active & (1u << j)) && cand[j])
              load_slot<K>(&slots[sub_base + li[j] + ((uint32_t)__builtin_ctz(cand[j]) >> 2)], sk[j], sr[j]);
          }
        }
        preloaded = false;
#pragma unroll
        for (int j = 0; j < PP_R; ++j) {
          if (ballot((active >> j) & 1u) == 0) continue;
          if (!(active & (1u << j))) continue;
          if (cand[j]) {
            if (sk[j] == kB[j]) {  // a tagged slot is never empty
              if (m[j] == 0) first[j] = sr[j];
              ++m[j];
            }
            cand[j] &= cand[j] - 1;
          }
          if (cand[j] == 0) {
            if (ended & (1u << j)) {
              active &= ~(1u << j);
            } else {  // the chain runs on: next 8 slots (tags from global memory: the LDS may hold another partition's)
              li[j] += 8;
              if (li[j] > SUB - 8) {
                uint64_t gs = sub_base + li[j];
                for (;;) {
                  K k;
                  int32_t r;
                  load_slot<K>(&slots[gs & mask], k, r);
                  if (r == EMPTY_ROW) break;
                  if (k == kB[j]) {
                    if (m[j] == 0) first[j] = r;
                    ++m[j];
                  }
                  ++gs;
                }
                active &= ~(1u << j);
              } else {
                const bool e = scan_tags8_global(gtagw, li[j], tag_of<K>(kB[j], log2cap) * 0x11111111u, cand[j]);
                if (e) {
                  ended |= 1u << j;
                  if (cand[j] == 0) active &= ~(1u << j);
                }
              }
            }
          }
        }
      }
      // ---- stage: pairs of this piece go to s_sidx / s_sfirst [buf] at positions handed out by an LDS counter
#pragma unroll
      for (int j = 0; j < PP_R; ++j) {
        const bool live = (fl >> j) & 1u;
        if (left_outer && live && m[j] == 0) m[j] = 1;  // (row, JoinNoMatch); first[j] is NO_MATCH
        uint32_t off, tot;
        if (ballot(m[j] > 1) == 0) {
          const uint64_t bb = ballot(m[j] == 1);
          if (bb == 0) continue;
          off = (uint32_t)__builtin_popcountll(bb & lanemask_lt());
          tot = (uint32_t)__builtin_popcountll(bb);
        } else {
          const uint32_t sc = wave_inclusive_scan(m[j], SumOp());
          off               = sc - m[j];
          tot               = shfl(sc, GX_WAVE - 1);
        }
        uint32_t wbase = 0;
        if (lane == 0) wbase = atomicAdd(&s_cnt[(unsigned)it3 & 7u], tot);
        wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
        if (m[j] == 0) continue;
        const uint32_t pos = wbase + off;
        if (m[j] == 1) {
          if (pos < (uint32_t)PP_ROWS) {
            s_sidx[buf * PP_ROWS + pos]   = iB[j];
            s_sfirst[buf * PP_ROWS + pos] = first[j];
          } else {  // staging full (duplicate build keys): reserve and write directly
            const unsigned long long gp = atomicAdd(cursor, 1ull);
            if ((int64_t)gp < capacity) {
              out_probe[gp] = iB[j];
              out_build[gp] = first[j];
            }
          }
        } else {  // duplicate build keys: walk the chain again (lines are cache-resident)
          const uint32_t room   = pos < (uint32_t)PP_ROWS ? (uint32_t)PP_ROWS - pos : 0u;
          const uint32_t staged = room < m[j] ? room : m[j];
          unsigned long long gp = 0;
          if (staged < m[j]) gp = atomicAdd(cursor, (unsigned long long)(m[j] - staged));
          uint32_t seen = 0;
          uint64_t hh   = slot_of<K>(kB[j], log2cap);
          for (;;) {
            K k;
            int32_t r;
            load_slot<K>(&slots[hh], k, r);
            if (r == EMPTY_ROW) break;
            if (k == kB[j]) {
              if (seen < staged) {
                s_sidx[buf * PP_ROWS + pos + seen]   = iB[j];
                s_sfirst[buf * PP_ROWS + pos + seen] = r;
              } else {
                if ((int64_t)gp < capacity) {
                  out_probe[gp] = iB[j];
                  out_build[gp] = r;
                }
                ++gp;
              }
              ++seen;
            }
            hh = (hh + 1) & mask;
          }
        }
      }
    }
    // ---------------- tags of the partition S2 is about to probe (workgroup-uniform branch; the S2 of the previous
    // trip finished before X(t-1), S3 does not read the LDS tags)
    if (validA && (!tags_loaded || partA != partC)) {
      const uint4* src = reinterpret_cast<const uint4*>(gtags + (((uint64_t)partA << PJ_SUB_LOG2) >> 1));
      uint4* dst       = reinterpret_cast<uint4*>(smem);
      for (uint32_t i = tid; i < SUB / 2 / 16; i += PP_PW * GX_WAVE) dst[i] = src[i];
      partC       = partA;
      tags_loaded = true;
      __syncthreads();  // R(t): every probe wave's share of the tags is in place (the service wave joins this barrier)
    }
    // ---------------- S2(t-1): chain heads on the LDS tags, first candidate slot in flight
    fl     = 0;
    validB = validA;
    partB  = partA;
    if (validA) {
      const uint64_t sub_base = (uint64_t)partA << PJ_SUB_LOG2;
      const Slot<K>* dummy    = slots + sub_base;  // what lanes without a candidate read: one line for the whole wave
      const unsigned long long pb = cA0 + (unsigned long long)w * (PP_R * GX_WAVE) + lane;
#pragma unroll
      for (int j = 0; j < PP_R; ++j) {
        kB[j]   = kA[j];
        iB[j]   = iA[j];
        cand[j] = 0;
        li[j]   = 0;
        if (pb + (unsigned long long)j * GX_WAVE < cA1) {
          fl |= 1u << j;
          const uint64_t prod = (uint64_t)kB[j] * 0x9E3779B97F4A7C15ull;
          li[j]               = (uint32_t)((prod >> (64 - log2cap)) - sub_base);
          uint32_t tg         = (uint32_t)(prod >> (60 - log2cap)) & 15u;
          tg                  = tg ? tg : 8u;
          if (li[j] > SUB - 8) {
            fl |= 1u << (8 + j);
          } else if (scan_tags8(s_tagw, li[j], tg * 0x11111111u, cand[j])) {
            fl |= 1u << (4 + j);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < PP_R; ++j) {
        const Slot<K>* sp = cand[j] ? slots + (sub_base + li[j] + ((uint32_t)__builtin_ctz(cand[j]) >> 2)) : dummy;
        sv[j]             = *reinterpret_cast<const Raw*>(sp);
      }
    }
    // ---------------- S1(t): loads of the next piece
    {
      const PpPiece pc = s_piece[t & 3];
      validA           = pc.valid != 0;
      if (done_at < 0 && !validA) done_at = t;
      if (validA) {
        partA = pc.part;
        cA0   = pc.c0;
        cA1   = pc.c1;
        const unsigned long long pb = cA0 + (unsigned long long)w * (PP_R * GX_WAVE) + lane;
#pragma unroll
        for (int j = 0; j < PP_R; ++j) {
          const unsigned long long i  = pb + (unsigned long long)j * GX_WAVE;
          const unsigned long long ic = i < cA1 ? i : cA0;  // clamped: the load is unconditional, dead rows are masked by `fl`
          kA[j] = __builtin_nontemporal_load(&pkeys[ic]);
          iA[j] = __builtin_nontemporal_load(&pidx[ic]);
        }
      }
    }
    __syncthreads();  // X(t): staging of piece t-2 complete, s_base of piece t-3 visible
    if (done_at >= 0 && t >= done_at + 3) break;
    // ---------------- flush of piece t-3: two coalesced streams
    const int itf = t - 3;
    if (itf >= 0) {
      const unsigned buf          = (unsigned)itf % 3u;
      unsigned int c              = s_cnt[(unsigned)itf & 7u];
      c                           = c < (unsigned)PP_ROWS ? c : (unsigned)PP_ROWS;
      const unsigned long long gb = s_base[buf];
      for (unsigned int i = tid; i < c; i += PP_PW * GX_WAVE) {
        const unsigned long long gp = gb + i;
        if ((int64_t)gp < capacity) {
          __builtin_nontemporal_store(s_sidx[buf * PP_ROWS + i], &out_probe[gp]);
          __builtin_nontemporal_store(s_sfirst[buf * PP_ROWS + i], &out_build[gp]);
        }
      }
    }
  }
}

// Partitioned build: the build rows go through the same partition pass, then each chunk inserts into
// the ~2 MiB sub-table its partition maps to while the other workgroups of the XCD insert into the
// same one -- the CAS and the 16-B slot write hit L2 instead of scattering over the whole table
// (k_build wrote 9.6 GB to HBM for 1.6 GB of slots at 1e8 rows).
template <typename K>
__global__ void __launch_bounds__(PJ_BT) k_pj_build(const K* __restrict__ pkeys, const int32_t* __restrict__ pidx,
                                                    PjPlan* plan, int pbits, Slot<K>* __restrict__ slots, uint32_t log2cap,
                                                    const unsigned int* __restrict__ gate = nullptr)
{
  if (gate && *gate == 0) return;
  constexpr int RPT = PJ_CHUNK / PJ_BT;
  __shared__ unsigned int s_misc[4];
  const int P         = 1 << pbits;
  const int LISTP     = P / PJ_NR;
  const uint64_t mask = (1ull << log2cap) - 1;
  const unsigned tid  = threadIdx.x;
  if (tid == 0) {
    const unsigned x = pj_xcc();
    unsigned int g   = 0xFFFFFFFFu;
    for (int i = 0; i < PJ_NR; ++i) {
      const unsigned y       = (x + i) % PJ_NR;
      const unsigned int nch = plan->list_chunk0[y + 1] - plan->list_chunk0[y];
      if (nch == 0) continue;
      const unsigned int t = atomicAdd(&plan->ticket[y].v, 1u);
      if (t < nch) {
        g         = plan->list_chunk0[y] + t;
        s_misc[1] = y;
        break;
      }
    }
    s_misc[0] = g;
  }
  __syncthreads();
  const unsigned int g = s_misc[0];
  if (g == 0xFFFFFFFFu) return;
  {
    const unsigned y = s_misc[1];
    for (int e = (int)tid; e < LISTP; e += PJ_BT) {
      const int p           = (int)y * LISTP + e;
      const unsigned int lo = plan->chunk0[p], hi = plan->chunk0[p + 1];
      if (lo <= g && g < hi) {
        s_misc[2] = (unsigned int)p;
        s_misc[3] = g - lo;
      }
    }
  }
  __syncthreads();
  const unsigned int part = s_misc[2];
  const unsigned long long p0 = plan->offset[part], p1 = plan->offset[part + 1];
  const unsigned long long c0 = p0 + (unsigned long long)s_misc[3] * PJ_CHUNK;
  const unsigned long long c1 = c0 + PJ_CHUNK < p1 ? c0 + PJ_CHUNK : p1;
#pragma unroll 4
  for (int j = 0; j < RPT; ++j) {
    const unsigned long long i = c0 + (unsigned long long)j * PJ_BT + tid;
    if (i >= c1) continue;
    const K key       = pkeys[i];
    const int32_t row = pidx[i];
    uint64_t h        = slot_of<K>(key, log2cap);
    for (;;) {
      const int32_t old = atomicCAS(&slots[h].row, EMPTY_ROW, row);
      if (old == EMPTY_ROW) {
        slots[h].key = key;
        break;
      }
      h = (h + 1) & mask;
    }
  }
}

// ================================================================================================
// Round 4: the sub-table build.  k_pj_build above claims slots with device-scope CAS on the table itself -- memory-side
// atomics, one read-modify-write of a line per row (12.9 GB of HBM traffic for 1.6 GB of slots at 1e8 rows) -- and k_tags then
// reads the whole table back (4.5 GB) to derive the tags.  But a slot's 4-bit tag is non-zero exactly when the slot is
// occupied, so the TAGS are the occupancy map: one workgroup owns a sub-table (2^17 slots = one partition), keeps its 64 KiB
// of tags in LDS, claims a slot with an LDS compare-and-swap on the tag word (first zero nibble at or behind the key's home,
// SWAR over the 8 nibbles of a word), stores the 16-byte slot straight to its final place (nobody reads it back here) and
// writes the finished tag block out in one coalesced run.  No global atomic, no table read.  A chain that runs off the end of
// the sub-table (a few hundred rows per 1e8; everything for heavily repeated keys) is parked in a list and inserted afterwards
// by k_bs_fixup with device-scope CAS on the GLOBAL tag words, starting at the first slot of the next sub-table -- every slot
// from the key's home to the end of its own sub-table is occupied, so the linear-probe invariant holds.
// ================================================================================================
constexpr int BS_BT   = 512;
constexpr int BS_LIST = 1 << 18;  // parked rows kept as positions; beyond that they are marked in place (row -> ~row)
struct alignas(128) BuildFix {
  unsigned int count;  // rows whose chain left their sub-table
  unsigned int pad[31];
  unsigned int list[BS_LIST];  // their positions in the partitioned arrays
};

template <typename K>
__device__ __forceinline__ void store_slot(Slot<K>* s, K key, int32_t row);
template <>
__device__ __forceinline__ void store_slot<uint64_t>(Slot<uint64_t>* s, uint64_t key, int32_t row)
{
  *reinterpret_cast<uint4*>(s) = make_uint4((uint32_t)key, (uint32_t)(key >> 32), (uint32_t)row, 0u);
}
template <>
__device__ __forceinline__ void store_slot<uint32_t>(Slot<uint32_t>* s, uint32_t key, int32_t row)
{
  *reinterpret_cast<uint2*>(s) = make_uint2(key, (uint32_t)row);
}
// LSB of every nibble of v that is zero, restricted to nibbles >= first
__device__ __forceinline__ uint32_t zero_nibbles_from(uint32_t v, uint32_t first)
{
  const uint32_t any = v | (v >> 1) | (v >> 2) | (v >> 3);
  return ~any & 0x11111111u & (0xFFFFFFFFu << (4u * first));
}

template <typename K>
__global__ void __launch_bounds__(BS_BT) k_bs_build(const K* __restrict__ pkeys, int32_t* __restrict__ pidx, const PjPlan* __restrict__ plan,
                                                    Slot<K>* __restrict__ slots, uint8_t* __restrict__ tags, uint32_t log2cap, BuildFix* fix)
{
  constexpr uint32_t SUB   = 1u << PJ_SUB_LOG2;
  constexpr uint32_t WORDS = SUB / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* s_tags = reinterpret_cast<uint32_t*>(smem);
  const unsigned tid  = threadIdx.x;
  const unsigned part = blockIdx.x;
  for (uint32_t i = tid; i < WORDS / 4; i += BS_BT) reinterpret_cast<uint4*>(s_tags)[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  const unsigned long long p0 = plan->offset[part], p1 = plan->offset[part + 1];
  Slot<K>* sub = slots + ((size_t)part << PJ_SUB_LOG2);
  constexpr int U = 4;
  for (unsigned long long i0 = p0 + tid; i0 < p1; i0 += (unsigned long long)BS_BT * U) {
    K key[U];
    int32_t row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long i = i0 + (unsigned long long)u * BS_BT;
      key[u] = i < p1 ? __builtin_nontemporal_load(&pkeys[i]) : K(0);
      row[u] = i < p1 ? __builtin_nontemporal_load(&pidx[i]) : 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long i = i0 + (unsigned long long)u * BS_BT;
      if (i >= p1) continue;
      const uint32_t t = tag_of<K>(key[u], log2cap);
      uint32_t pos     = (uint32_t)slot_of<K>(key[u], log2cap) & (SUB - 1);
      uint32_t w = pos >> 3, first = pos & 7u;
      bool placed = false;
      while (w < WORDS) {
        uint32_t v = s_tags[w];
        for (;;) {
          const uint32_t z = zero_nibbles_from(v, first);
          if (z == 0) break;
          const uint32_t nib = (uint32_t)__builtin_ctz(z) >> 2;
          const uint32_t old = atomicCAS(&s_tags[w], v, v | (t << (4u * nib)));
          if (old == v) {
            store_slot<K>(&sub[(w << 3) + nib], key[u], row[u]);
            placed = true;
            break;
          }
          v = old;
        }
        if (placed) break;
        ++w;
        first = 0;
      }
      if (!placed) {  // the chain ran off the end of this sub-table
        const unsigned int e = atomicAdd(&fix->count, 1u);
        if (e < (unsigned int)BS_LIST) fix->list[e] = (unsigned int)i;
        else pidx[i] = ~row[u];  // (rows and payloads are >= 0)
      }
    }
  }
  __syncthreads();
  uint4* out = reinterpret_cast<uint4*>(tags + ((size_t)part << (PJ_SUB_LOG2 - 1)));
  for (uint32_t i = tid; i < WORDS / 4; i += BS_BT) out[i] = reinterpret_cast<const uint4*>(s_tags)[i];
}

// one parked row: claim the first free slot at or behind `start` through the GLOBAL tag words (device-scope CAS), write the slot
template <typename K>
__device__ __forceinline__ void bs_insert_global(K key, int32_t row, uint64_t start, Slot<K>* slots, uint32_t* tagw, uint32_t log2cap)
{
  const uint64_t mask = (1ull << log2cap) - 1;
  const uint32_t t    = tag_of<K>(key, log2cap);
  uint64_t pos        = start & mask;
  for (;;) {
    const uint64_t w = pos >> 3;
    uint32_t v       = __hip_atomic_load(&tagw[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t first   = (uint32_t)pos & 7u;
    for (;;) {
      const uint32_t z = zero_nibbles_from(v, first);
      if (z == 0) break;
      const uint32_t nib = (uint32_t)__builtin_ctz(z) >> 2;
      const uint32_t old = atomicCAS(&tagw[w], v, v | (t << (4u * nib)));
      if (old == v) {
        store_slot<K>(&slots[(w << 3) + nib], key, row);
        return;
      }
      v = old;
    }
    pos = ((w + 1) << 3) & mask;
  }
}
template <typename K>
__global__ void __launch_bounds__(256) k_bs_fixup(const K* __restrict__ pkeys, const int32_t* __restrict__ pidx, int64_t n, Slot<K>* slots,
                                                  uint8_t* tags, uint32_t log2cap, const BuildFix* __restrict__ fix, int scan_all)
{
  const unsigned int cnt = fix->count;
  if (cnt == 0 || (scan_all && cnt <= (unsigned int)BS_LIST)) return;
  uint32_t* tagw       = reinterpret_cast<uint32_t*>(tags);
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t gid    = (int64_t)blockIdx.x * 256 + threadIdx.x;
  auto next_sub = [&](K key) { return ((slot_of<K>(key, log2cap) >> PJ_SUB_LOG2) + 1) << PJ_SUB_LOG2; };
  if (!scan_all) {
    const int64_t m = cnt < (unsigned int)BS_LIST ? cnt : BS_LIST;
    for (int64_t e = gid; e < m; e += stride) {
      const unsigned int i = fix->list[e];
      const K key          = pkeys[i];
      bs_insert_global<K>(key, pidx[i], next_sub(key), slots, tagw, log2cap);
    }
  } else {  // the list overflowed: the surplus rows were marked in place
    for (int64_t i = gid; i < n; i += stride) {
      const int32_t r = pidx[i];
      if (r >= 0) continue;
      const K key = pkeys[i];
      bs_insert_global<K>(key, ~r, next_sub(key), slots, tagw, log2cap);
    }
  }
}

// ================================================================================================
// Round 4b: the WINDOW build.  k_bs_build above stores every 16-byte slot straight into a 2 MiB sub-table that nothing keeps
// cached: 1e8 partial-line writes into a table that a 4.3 GB memset has to fill with EMPTY first (3.65 + 0.8 of the 5.3 ms,
// profiles/r4_run4_join_kernel_stats.txt).  Here the table is written ONCE, in full lines, and never pre-filled:
//   k_bw_split  one workgroup per partition (sub-table): counts its rows per WINDOW (2^12 slots), then regroups them by window
//               through LDS (tiles of 4096 rows: 128-row runs) into a second pair of arrays; window offsets go to `woffs`;
//   k_bw_build  one workgroup per sub-table walks its 32 windows in order: the window's slots live in LDS (keys + rows, 48 KiB),
//               rows claim them with an LDS compare-and-swap, the finished window leaves as 64 KiB of coalesced 16-byte stores
//               (empty slots included) + 2 KiB of tags.  A chain that runs off the end of a window is CARRIED into the next one
//               (a small LDS list, inserted first: every slot from the key's home to the window's end is occupied, so the
//               linear-probe invariant holds); what does not fit the list, and what leaves the sub-table, is parked in a global
//               list with the slot to resume at and inserted by k_bw_fixup through the global tag words, as before.
// A parked list that overflows (hundreds of thousands of chains running off their windows: one key repeated without end)
// raises `failed`: the round-2 kernels, enqueued behind and gated on it, then build the table from the level-1 partition.
// ================================================================================================
constexpr int BW_LOG2  = 12;                           // slots per window
constexpr int BW_SLOTS = 1 << BW_LOG2;
constexpr int BW_PER   = 1 << (PJ_SUB_LOG2 - BW_LOG2);  // windows per sub-table (32)
constexpr int BW_BT    = 512;
constexpr int BW_RPT   = 8;
constexpr int BW_TILE  = BW_BT * BW_RPT;                // rows per split tile
constexpr int BW_CARRY = 128;                           // rows carried from one window into the next through LDS
constexpr int BW_LIST  = 1 << 18;                       // parked rows
struct alignas(128) BuildFix2 {
  unsigned int count;   // parked rows
  unsigned int failed;  // the list overflowed: the gated round-2 kernels build the table
  unsigned int pad[30];
  uint4 list[BW_LIST];  // {key lo, key hi, row, slot to resume at}
};

template <typename K>
__device__ __forceinline__ uint32_t bw_window(K key, uint32_t log2cap)
{
  return ((uint32_t)slot_of<K>(key, log2cap) >> BW_LOG2) & (uint32_t)(BW_PER - 1);
}

template <typename K>
__global__ void __launch_bounds__(BW_BT) k_bw_split(const K* __restrict__ pkeys, const int32_t* __restrict__ pidx, const PjPlan* __restrict__ plan,
                                                    K* __restrict__ wkeys, int32_t* __restrict__ widx, uint32_t* __restrict__ woffs, uint32_t log2cap)
{
  __shared__ K s_k[BW_TILE];
  __shared__ int32_t s_i[BW_TILE];
  __shared__ uint32_t s_tot[BW_PER], s_woff[BW_PER + 1], s_run[BW_PER], s_tc[BW_PER], s_start[BW_PER], s_delta[BW_PER];
  const unsigned tid  = threadIdx.x;
  const unsigned part = blockIdx.x;
  const unsigned long long o0 = plan->offset[part], o1 = plan->offset[part + 1];
  if (tid < BW_PER) {
    s_tot[tid] = 0;
    s_run[tid] = 0;
  }
  __syncthreads();
  // ---- sweep A: rows per window
  for (unsigned long long i0 = o0; i0 < o1; i0 += BW_TILE) {
    K key[BW_RPT];
#pragma unroll
    for (int j = 0; j < BW_RPT; ++j) {
      const unsigned long long i = i0 + (unsigned long long)j * BW_BT + tid;
      key[j] = i < o1 ? pkeys[i] : K(0);
    }
#pragma unroll
    for (int j = 0; j < BW_RPT; ++j) {
      const unsigned long long i = i0 + (unsigned long long)j * BW_BT + tid;
      (void)lds_rank_few<5>(s_tot, bw_window<K>(key[j], log2cap), i < o1);
    }
  }
  __syncthreads();
  if (tid < GX_WAVE) {  // one wave: exclusive scan of the 32 counts
    const uint32_t c   = tid < BW_PER ? s_tot[tid] : 0u;
    const uint32_t inc = wave_inclusive_sum_dpp(c);
    if (tid < BW_PER) {
      s_woff[tid] = inc - c;
      woffs[(size_t)part * (BW_PER + 1) + tid] = inc - c;
    }
    if (tid == BW_PER - 1) {
      s_woff[BW_PER] = inc;
      woffs[(size_t)part * (BW_PER + 1) + BW_PER] = inc;
    }
  }
  __syncthreads();
  // ---- sweep B: regroup by window, tile by tile
  for (unsigned long long i0 = o0; i0 < o1; i0 += BW_TILE) {
    const int nvalid = (int)(o1 - i0 < (unsigned long long)BW_TILE ? o1 - i0 : (unsigned long long)BW_TILE);
    if (tid < BW_PER) s_tc[tid] = 0;
    K key[BW_RPT];
    int32_t row[BW_RPT];
#pragma unroll
    for (int j = 0; j < BW_RPT; ++j) {
      const int idx = j * BW_BT + (int)tid;
      key[j] = idx < nvalid ? pkeys[i0 + idx] : K(0);
      row[j] = idx < nvalid ? pidx[i0 + idx] : 0;
    }
    __syncthreads();
    uint32_t packed[BW_RPT];
#pragma unroll
    for (int j = 0; j < BW_RPT; ++j) {
      const int idx    = j * BW_BT + (int)tid;
      const uint32_t w = bw_window<K>(key[j], log2cap);
      packed[j]        = (w << 16) | lds_rank_few<5>(s_tc, w, idx < nvalid);
    }
    __syncthreads();
    if (tid < GX_WAVE) {
      const uint32_t c   = tid < BW_PER ? s_tc[tid] : 0u;
      const uint32_t inc = wave_inclusive_sum_dpp(c);
      if (tid < BW_PER) {
        s_start[tid] = inc - c;
        s_delta[tid] = s_woff[tid] + s_run[tid] - (inc - c);
        s_run[tid] += c;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BW_RPT; ++j) {
      const int idx = j * BW_BT + (int)tid;
      if (idx < nvalid) {
        const uint32_t l = s_start[packed[j] >> 16] + (packed[j] & 0xFFFFu);
        s_k[l]           = key[j];
        s_i[l]           = row[j];
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BW_RPT; ++j) {
      const int i = j * BW_BT + (int)tid;
      if (i < nvalid) {
        const K k          = s_k[i];
        const uint32_t w   = bw_window<K>(k, log2cap);
        const uint64_t dst = o0 + (uint64_t)(uint32_t)(s_delta[w] + (uint32_t)i);
        wkeys[dst]         = k;
        widx[dst]          = s_i[i];
      }
    }
    __syncthreads();  // the next tile reuses the LDS
  }
}

template <typename K>
__global__ void __launch_bounds__(BW_BT) k_bw_build(const K* __restrict__ wkeys, const int32_t* __restrict__ widx, const PjPlan* __restrict__ plan,
                                                    const uint32_t* __restrict__ woffs, Slot<K>* __restrict__ slots, uint8_t* __restrict__ tags,
                                                    uint32_t log2cap, BuildFix2* fix)
{
  __shared__ K s_key[BW_SLOTS];
  __shared__ int32_t s_row[BW_SLOTS];
  __shared__ K s_ck[2][BW_CARRY];
  __shared__ int32_t s_cr[2][BW_CARRY];
  __shared__ uint32_t s_nc[2];
  const unsigned tid  = threadIdx.x;
  const unsigned part = blockIdx.x;
  const unsigned long long o0 = plan->offset[part];
  const uint32_t* wo  = woffs + (size_t)part * (BW_PER + 1);
  Slot<K>* sub        = slots + ((size_t)part << PJ_SUB_LOG2);
  uint32_t* tagw      = reinterpret_cast<uint32_t*>(tags) + ((size_t)part << (PJ_SUB_LOG2 - 3));
  if (tid < 2) s_nc[tid] = 0;
  // a row whose chain leaves the window (h == BW_SLOTS): into the LDS list of the next window, or parked with the slot to resume at
  auto carry = [&](K key, int32_t row, int nxt, uint64_t resume) {
    const uint32_t e = atomicAdd(&s_nc[nxt], 1u);
    if (e < (uint32_t)BW_CARRY) {
      s_ck[nxt][e] = key;
      s_cr[nxt][e] = row;
    } else {
      const unsigned int g = atomicAdd(&fix->count, 1u);
      if (g < (unsigned int)BW_LIST) fix->list[g] = make_uint4((uint32_t)(uint64_t)key, (uint32_t)((uint64_t)key >> 32), (uint32_t)row, (uint32_t)resume);
      else fix->failed = 1u;
    }
  };
  auto insert = [&](K key, int32_t row, uint32_t h, int nxt, uint64_t resume) {
    for (;;) {
      if (h >= (uint32_t)BW_SLOTS) {
        carry(key, row, nxt, resume);
        return;
      }
      if (atomicCAS(&s_row[h], EMPTY_ROW, row) == EMPTY_ROW) {
        s_key[h] = key;
        return;
      }
      ++h;
    }
  };
  for (int w = 0; w < BW_PER; ++w) {
    const int cur = w & 1, nxt = cur ^ 1;
    const uint64_t resume = ((uint64_t)part << PJ_SUB_LOG2) + ((uint64_t)(w + 1) << BW_LOG2);  // first slot behind this window
    for (int i = tid; i < BW_SLOTS; i += BW_BT) s_row[i] = EMPTY_ROW;
    __syncthreads();  // (also: s_nc[cur] of the previous round is complete, s_nc[nxt] may be reset)
    if (tid == 0) s_nc[nxt] = 0;
    const uint32_t ncar = s_nc[cur] < (uint32_t)BW_CARRY ? s_nc[cur] : (uint32_t)BW_CARRY;
    __syncthreads();
    for (uint32_t e = tid; e < ncar; e += BW_BT) insert(s_ck[cur][e], s_cr[cur][e], 0u, nxt, resume);  // carried rows start at slot 0
    const unsigned long long r0 = o0 + wo[w], r1 = o0 + wo[w + 1];
    for (unsigned long long i = r0 + tid; i < r1; i += BW_BT) {
      const K key = __builtin_nontemporal_load(&wkeys[i]);
      insert(key, __builtin_nontemporal_load(&widx[i]), (uint32_t)slot_of<K>(key, log2cap) & (uint32_t)(BW_SLOTS - 1), nxt, resume);
    }
    __syncthreads();
    // the finished window: 16-byte slots (empty ones as {0, EMPTY_ROW}) and one tag word per 8 slots
    Slot<K>* out = sub + ((size_t)w << BW_LOG2);
    for (int i = tid; i < BW_SLOTS; i += BW_BT) {
      const int32_t r = s_row[i];
      store_slot<K>(&out[i], r == EMPTY_ROW ? K(0) : s_key[i], r);
    }
    {
      uint32_t word = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = (int)tid * 8 + k;
        if (s_row[i] != EMPTY_ROW) word |= tag_of<K>(s_key[i], log2cap) << (4 * k);
      }
      tagw[((size_t)w << (BW_LOG2 - 3)) + tid] = word;  // BW_SLOTS / 8 == BW_BT words per window
    }
    __syncthreads();  // the next window reuses the LDS
  }
  // what the last window carried leaves the sub-table: parked, to resume at the first slot of the next one
  {
    const int last        = BW_PER & 1;
    const uint32_t ncar   = s_nc[last] < (uint32_t)BW_CARRY ? s_nc[last] : (uint32_t)BW_CARRY;
    const uint64_t resume = (uint64_t)(part + 1) << PJ_SUB_LOG2;
    for (uint32_t e = tid; e < ncar; e += BW_BT) {
      const unsigned int g = atomicAdd(&fix->count, 1u);
      const K key          = s_ck[last][e];
      if (g < (unsigned int)BW_LIST) fix->list[g] = make_uint4((uint32_t)(uint64_t)key, (uint32_t)((uint64_t)key >> 32), (uint32_t)s_cr[last][e], (uint32_t)resume);
      else fix->failed = 1u;
    }
  }
}
static_assert(BW_SLOTS / 8 == BW_BT, "k_bw_build: one tag word per thread and window");

template <typename K>
__global__ void __launch_bounds__(256) k_bw_fixup(Slot<K>* slots, uint8_t* tags, uint32_t log2cap, const BuildFix2* __restrict__ fix)
{
  const unsigned int cnt = fix->count;
  if (cnt == 0 || fix->failed) return;
  uint32_t* tagw       = reinterpret_cast<uint32_t*>(tags);
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)cnt; e += stride) {
    const uint4 r = fix->list[e];
    const K key   = (K)(((uint64_t)r.y << 32) | r.x);
    bs_insert_global<K>(key, (int32_t)r.z, (uint64_t)r.w, slots, tagw, log2cap);
  }
}
// the gated fallback needs an EMPTY table first (the window build never pre-fills it)
__global__ void __launch_bounds__(256) k_bw_fill_empty(uint4* __restrict__ p, size_t n16, const unsigned int* __restrict__ gate)
{
  if (*gate == 0) return;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
}

// ================================================================================================
// Round 3: the partition pass without its histogram, persistent, and the probe over region tables.
//
// (1) No k_pj_hist.  Region e = partition * PJ_NR + range owns a fixed slot of `cap` rows in the partitioned arrays
//     (cap = mean + 8 sigma of a uniform hash, so a row count that overflows its slot means skewed keys, not bad
//     luck); a tile adds its count to the region's fill counter and writes behind what the earlier tiles of that
//     range wrote.  A region that outgrows its slot raises `overflow`: its surplus rows are dropped, k_pj2_offsets
//     turns the flag into `fallback`, the speculative probe finds no piece to take, and the EXACT sequence -- histogram,
//     offsets, scatter into exactly sized partitions, probe -- which is enqueued behind it and exits at once
//     otherwise, produces the result.  No host round trip on either branch (the trick of the sort's level-1 pass).
//     Saves the 8 B/row histogram read: 1.35 ms of 14.3 at 1e9 rows.
// (2) Persistent scatter.  One 1024-thread workgroup per CU owns ~130 KiB of LDS, so nothing else is resident to
//     hide its serial phases; per tile it used to pay the HBM latency of its key loads, two returning device-scope
//     atomics, the drain of its stores and the launch of its successor (~11 of 27 us).  Now a workgroup walks tiles
//     v = blockIdx.x + k * gridDim.x (same XCD for every k, so range == XCD as before): the keys of the NEXT tile are
//     requested as soon as the current ones sit in LDS (their registers are free from there on) and arrive under the
//     write-out; the fill-counter atomics of ALL bins of a thread are issued together, before the block scan, and are
//     consumed only after the keys have been moved to LDS.
// (3) Probe pieces come from a region table (chunk0 per region) instead of per-partition row ranges, so the same
//     kernel serves the padded regions and the exact partitions; and the loads of a piece are issued at the TOP of a
//     trip into their own registers (copied into the S1->S2 set at the bottom), which gives HBM a whole trip to
//     deliver them: the old order left them the barrier, the flush and S3 (~1.5 us) and every trip stalled on them.
// ================================================================================================
struct alignas(128) Pj2Plan {
  unsigned int fill[PJ_MAXP * PJ_NR];        // rows written to region e = partition * PJ_NR + range
  // The counters the partition pass bumps, RANGE-major: fillx[range * P + partition]; k_pj2_offsets copies them into fill[].  A range is
  // walked by ONE XCD, so a line of fillx is touched by one L2 only.  With the atomics on fill[] itself -- the eight ranges' counters of
  // a partition side by side, every line of counters bumped by all eight XCDs -- the lines travelled from L2 to L2: WRITE_SIZE of the
  // pass 16.9 GB for 12 GB of records (profiles/r6_pmc_traffic_1e9.json at 387665f), where the sort's and the groupby's partition
  // passes, whose cursors have always been range-major, write 1.03 - 1.05 x their bytes.
  unsigned int fillx[PJ_MAXP * PJ_NR];
  unsigned int chunk0[PJ_MAXP * PJ_NR + 1];  // first probe piece of region e, regions in (partition, range) order
  unsigned int list_chunk0[PJ_NR + 1];       // first piece of XCD list y (partitions [y, y + 1) * P / PJ_NR)
  alignas(128) unsigned int overflow;        // a region outgrew its slot
  alignas(128) unsigned int fallback;        // set by k_pj2_offsets: the exact sequence must run
  alignas(128) PjCounter ticket[PJ_NR];
  PjCounter ticket_exact[PJ_NR];
};

// rows per region slot: mean + 8 standard deviations of a binomial(range rows, 1/P) count, a multiple of 32 rows
static inline uint32_t pj2_cap(int64_t n, int pbits)
{
  const double mean = (double)n / (double)PJ_NR / (double)(1 << pbits);
  double cap        = mean + 8.0 * __builtin_sqrt(mean + 1.0) + 64.0;
  return (uint32_t)((((int64_t)cap + 31) / 32) * 32);
}

template <typename K, int RPT, int BTt, bool EXACT, typename F>
__global__ void __launch_bounds__(BTt) k_pj2_scatter(const K* __restrict__ keys, int64_t n, Pj2Plan* plan2, PjPlan* plan, int pbits,
                                                     int64_t rrows, uint32_t cap, int64_t ntiles, K* __restrict__ pkeys,
                                                     int32_t* __restrict__ pidx, F part_of, int32_t row0,
                                                     const int32_t* __restrict__ payload, int xmajor)
{
  constexpr int TILE = BTt * RPT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  K* s_k                = reinterpret_cast<K*>(smem);                                        // TILE (reused for the row indices)
  unsigned int* s_cd    = reinterpret_cast<unsigned int*>(smem + (size_t)TILE * sizeof(K));  // P: counts, then global - tile position
  unsigned int* s_start = s_cd + (1 << pbits);                                               // P
  __shared__ unsigned int s_scan[BTt / GX_WAVE + 1];
  if (EXACT && plan2->fallback == 0) return;  // the speculative pass held: nothing to redo
  const int P        = 1 << pbits;
  const unsigned tid = threadIdx.x;
  const int bpt      = P > BTt ? P / BTt : 1;                 // bins per thread (consecutive)
  const int b0       = P > BTt ? (int)tid * bpt : (int)tid;   // (threads >= P own no bin when P < BTt)
  const bool owner   = P > BTt || (int)tid < P;

  K key[RPT];
  int64_t v = blockIdx.x;
  if (v >= ntiles) return;
  int64_t base;
  int nvalid, range;
  auto locate = [&](int64_t vv, int64_t& b, int& nv, int& r) {
    const int64_t tile = xcd_swizzle(vv, ntiles);
    b                  = tile * TILE;
    nv                 = (int)((n - b < (int64_t)TILE) ? (n - b) : (int64_t)TILE);
    r                  = (rrows > 0 && b / rrows < PJ_NR - 1) ? (int)(b / rrows) : PJ_NR - 1;
  };
  auto load = [&](int64_t b, int nv) {
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const int idx = j * BTt + (int)tid;
      key[j]        = __builtin_nontemporal_load(&keys[b + (idx < nv ? idx : 0)]);  // unconditional: rows >= nv are masked below
    }
  };
  locate(v, base, nvalid, range);
  load(base, nvalid);
  for (;;) {
    for (int i = tid; i < P; i += BTt) s_cd[i] = 0;
    __syncthreads();  // also: the previous tile's last LDS reads are done
    unsigned int packed[RPT];  // partition << 16 | rank inside (tile, partition)
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const int idx           = j * BTt + (int)tid;
      const unsigned int part = part_of(key[j]);
      const unsigned int rank = (idx < nvalid) ? atomicAdd(&s_cd[part], 1u) : 0u;
      packed[j]               = (part << 16) | rank;
    }
    __syncthreads();
    // counts of this thread's bins; the fill-counter atomics leave together and stay in flight across the scan
    unsigned int c[4], g[4];  // bpt <= 4 (P <= 4096, BTt >= 1024 when P > BTt ... checked on the host)
    unsigned int sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      c[k] = (owner && k < bpt) ? s_cd[b0 + k] : 0u;
      g[k] = 0;
      sum += c[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (c[k]) {
        if (EXACT) g[k] = (unsigned int)atomicAdd(&plan->cursor[range][b0 + k], (unsigned long long)c[k]);
        else g[k] = xmajor ? atomicAdd(&plan2->fillx[range * P + b0 + k], c[k]) : atomicAdd(&plan2->fill[(b0 + k) * PJ_NR + range], c[k]);
      }
    }
    unsigned int st = block_exclusive_scan<BTt>(sum, 0u, SumOp(), s_scan, (unsigned int*)nullptr);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (owner && k < bpt) s_start[b0 + k] = st;
      st += c[k];
    }
    __syncthreads();
    // keys through LDS: every partition becomes one contiguous run of the tile.  The position inside the tile is kept
    // (16 bits, two per register) for the row indices that follow through the same buffer.
    unsigned int lpos2[RPT / 2];
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const int idx        = j * BTt + (int)tid;
      const unsigned int l = s_start[packed[j] >> 16] + (packed[j] & 0xFFFFu);  // < TILE <= 65536
      if (idx < nvalid) s_k[l] = key[j];
      if (j & 1) lpos2[j / 2] |= l << 16; else lpos2[j / 2] = l & 0xFFFFu;
    }
    // the key registers are free: request the next tile now, it lands under the write-out
    const int64_t vn = v + gridDim.x;
    int64_t nbase    = 0;
    int nnvalid = 0, nrange = 0;
    const bool more = vn < ntiles;
    if (more) {
      locate(vn, nbase, nnvalid, nrange);
      load(nbase, nnvalid);
    }
    // global position of a run = slot base + rows already there (the atomics have had the scan and the LDS scatter to return)
    {
      unsigned int st2 = s_start[owner ? b0 : 0];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (owner && k < bpt) {
          unsigned int gb = g[k];
          if (!EXACT) {
            gb += (unsigned int)((b0 + k) * PJ_NR + range) * cap;
            if (c[k] && g[k] + c[k] > cap) plan2->overflow = 1u;  // the surplus is dropped below; the exact sequence will run
          }
          s_cd[b0 + k] = gb - st2;
        }
        st2 += c[k];
      }
    }
    __syncthreads();
    unsigned int obin2[RPT / 2];  // partition of the element this thread writes out (16 bits each; 0xFFFF: dropped)
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const int i     = j * BTt + (int)tid;
      unsigned int pt = 0xFFFFu;
      if (i < nvalid) {
        const K k = s_k[i];
        pt        = part_of(k);
        const unsigned int p = s_cd[pt] + (unsigned int)i;
        if (!EXACT && p >= (unsigned int)(pt * PJ_NR + range + 1) * cap) pt = 0xFFFFu;
        else pkeys[p] = k;
      }
      if (j & 1) obin2[j / 2] |= pt << 16; else obin2[j / 2] = pt;
    }
    __syncthreads();
    // row indices through the same buffer
    int32_t* s_i = reinterpret_cast<int32_t*>(smem);
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const int idx        = j * BTt + (int)tid;
      const unsigned int l = (j & 1) ? lpos2[j / 2] >> 16 : lpos2[j / 2] & 0xFFFFu;
      if (idx < nvalid) s_i[l] = payload ? payload[base + idx] : (int32_t)(base + idx) + row0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const int i           = j * BTt + (int)tid;
      const unsigned int pt = (j & 1) ? obin2[j / 2] >> 16 : obin2[j / 2] & 0xFFFFu;
      if (pt != 0xFFFFu) pidx[(unsigned int)(s_cd[pt] + (unsigned int)i)] = s_i[i];
    }
    if (!more) break;
    __syncthreads();  // the row-index write-out reads s_cd / s_i: nobody may start zeroing the counters of the next tile before
    v      = vn;
    base   = nbase;
    nvalid = nnvalid;
    range  = nrange;
  }
}

// Round 6: the partition pass writing 12-byte RECORDS {key, row} instead of a key array and a row array (gx_join_set_experiment
// bit 0; 8-byte keys).
//
// k_pj2_scatter at P = 2048 leaves runs of 16384 / 2048 = 8 rows per (tile, partition): a 64-B key run and a 32-B row run at an
// arbitrary 8- / 4-byte offset.  A wave's store instruction then touches 8 runs = ~11.5 128-B lines for its 64 keys and ~9.8 for
// its 64 rows, where 64 contiguous keys + rows would touch 6; the 256-way form of the same kernel (64-row runs) moves the same
// rows in 4.43 ms against 6.1 (profiles/r5_join_probe_ab.txt), i.e. the pass is bound by line transactions at the L2, not by
// bytes.  One 96-B record run per (tile, partition) touches ~13.5 lines per 64 rows instead of ~21, and the probe reads a lane's
// four rows as three 16-byte loads.  To have key AND row of a position at hand when it is written, the tile goes through LDS in
// windows of 8192 positions (64 KiB of keys + 32 KiB of rows) instead of keys first, rows second; RPT = 24 keys per thread
// (24576-row tiles, 12-row runs, three windows) costs 16 more registers and a third fewer fill-counter atomics.
// FULL tiles only (ntiles = n / TILE): the ragged tail is a second launch on the last rows with `tail_base`.
__device__ __forceinline__ void pj_opaque(uint64_t& k)
{  // the compiler must not carry values derived from k across this point (it would keep the partition numbers live next to the keys)
  uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
  asm volatile("" : "+v"(lo), "+v"(hi));
  k = ((uint64_t)hi << 32) | lo;
}
struct PjRec {  // one partitioned probe row
  uint32_t klo, khi;
  int32_t row;
};
template <int RPT, bool EXACT, bool TAIL, bool AOS, bool PAY, typename F>
__global__ void __launch_bounds__(1024) k_pj2_scatter_rec(const uint64_t* __restrict__ keys, int64_t n, Pj2Plan* plan2, PjPlan* plan, int pbits,
                                                          int64_t rrows, uint32_t cap, int64_t ntiles, int64_t tile0_row,
                                                          PjRec* __restrict__ precs, F part_of, int32_t row0,
                                                          const int32_t* __restrict__ payload, int32_t* __restrict__ soa_idx, int xmajor)
{
  typedef uint64_t K;
  constexpr int BTt  = 1024;
  constexpr int TILE = BTt * RPT;
  constexpr int WIN  = 8192;            // tile positions per LDS window
  constexpr int NWIN = (TILE + WIN - 1) / WIN;
  constexpr int RPW  = WIN / BTt;       // positions a thread writes out per window
  static_assert(TILE <= 65536 && TILE % WIN == 0, "tile positions travel as 16-bit halves; whole windows");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  K* s_k                = reinterpret_cast<K*>(smem);                                             // WIN keys
  int32_t* s_i          = reinterpret_cast<int32_t*>(smem + (size_t)WIN * sizeof(K));             // WIN rows
  unsigned int* s_cd    = reinterpret_cast<unsigned int*>(smem + (size_t)WIN * (sizeof(K) + 4));  // P: counts, then global - tile position
  unsigned int* s_start = s_cd + (1 << pbits);                                                    // P
  __shared__ unsigned int s_scan[BTt / GX_WAVE + 1];
  if (EXACT && plan2->fallback == 0) return;
  const int P      = 1 << pbits;
  unsigned tid     = threadIdx.x;
  const int bpt    = P > BTt ? P / BTt : 1;
  const int b0     = P > BTt ? (int)threadIdx.x * bpt : (int)threadIdx.x;
  const bool owner = P > BTt || (int)threadIdx.x < P;

  K key[RPT];
  int32_t pay[PAY ? RPT : 1];  // PAY: the rows' payloads (the sharded probe's encoded source rows) travel with the keys -- fetched inside the
                               // window loop, under a divergent test, their latency lay bare: 8.4 ms per 1e9 rows against 5.5 without payload
  int64_t v = blockIdx.x;
  if (v >= ntiles) return;
  // TAIL: the tail launch -- ONE partial tile starting at row tile0_row (nvalid < TILE), always in the last range; otherwise every
  // tile is full and the validity tests below fold away (hoisted out of the tile loop they cost 2 SGPRs per key, and spills)
  auto locate = [&](int64_t vv, int64_t& b, int& nv, int& r) {
    if (TAIL) {
      b  = tile0_row;
      nv = (int)(n - tile0_row);
      r  = PJ_NR - 1;
      return;
    }
    const int64_t tile = xcd_swizzle(vv, ntiles);
    b                  = tile * TILE;
    nv                 = TILE;
    r                  = (rrows > 0 && b / rrows < PJ_NR - 1) ? (int)(b / rrows) : PJ_NR - 1;
  };
  auto load = [&](int64_t b, int nv) {
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const int idx = j * BTt + (int)tid;
      key[j]        = __builtin_nontemporal_load(&keys[b + (idx < nv ? idx : 0)]);
      if (PAY) pay[j] = __builtin_nontemporal_load(&payload[b + (idx < nv ? idx : 0)]);
    }
  };
  int64_t base;
  int range, nvalid_;
  locate(v, base, nvalid_, range);
  load(base, nvalid_);
  for (;;) {
    const int nvalid = TAIL ? nvalid_ : TILE;
    asm volatile("" : "+v"(tid));  // per-iteration addresses are recomputed, not hoisted out of the tile loop and spilled (DESIGN, compiler note)
    for (int i = tid; i < P; i += BTt) s_cd[i] = 0;
    __syncthreads();
    unsigned int lpos2[RPT / 2];  // rank inside (tile, partition), then the tile position: 16 bits each, two per register (0xFFFF: no row)
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const int idx           = j * BTt + (int)tid;
      const unsigned int rank = idx < nvalid ? atomicAdd(&s_cd[part_of(key[j])], 1u) : 0xFFFFu;  // < TILE < 65535
      if (j & 1) lpos2[j / 2] |= rank << 16; else lpos2[j / 2] = rank;
    }
#pragma unroll
    for (int j = 0; j < RPT; ++j) pj_opaque(key[j]);
    __syncthreads();
    unsigned int c[4], g[4];
    unsigned int sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      c[k] = (owner && k < bpt) ? s_cd[b0 + k] : 0u;
      g[k] = 0;
      sum += c[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (c[k]) {
        if (EXACT) g[k] = (unsigned int)atomicAdd(&plan->cursor[range][b0 + k], (unsigned long long)c[k]);
        else g[k] = xmajor ? atomicAdd(&plan2->fillx[range * P + b0 + k], c[k]) : atomicAdd(&plan2->fill[(b0 + k) * PJ_NR + range], c[k]);
      }
    }
    unsigned int st = block_exclusive_scan<BTt>(sum, 0u, SumOp(), s_scan, (unsigned int*)nullptr);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (owner && k < bpt) s_start[b0 + k] = st;
      st += c[k];
    }
    __syncthreads();
    // rank -> tile position (the partition is recomputed from the key: a multiply and a shift against RPT more live registers)
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const unsigned int h   = (j & 1) ? lpos2[j / 2] >> 16 : lpos2[j / 2] & 0xFFFFu;
      const unsigned int add = h == 0xFFFFu ? 0u : s_start[part_of(key[j])];  // run start + rank < TILE: no carry out of a half
      if (j & 1) lpos2[j / 2] += add << 16; else lpos2[j / 2] += add;
    }
    {  // (s_cd's counts were consumed above and nobody reads it again before the barrier behind the first window's keys)
      unsigned int st2 = s_start[owner ? b0 : 0];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (owner && k < bpt) {
          unsigned int gb = g[k];
          if (!EXACT) {
            gb += (unsigned int)((b0 + k) * PJ_NR + range) * cap;
            if (c[k] && g[k] + c[k] > cap) plan2->overflow = 1u;
          }
          s_cd[b0 + k] = gb - st2;
        }
        st2 += c[k];
      }
    }
    const int64_t vn = v + gridDim.x;
    int64_t nbase    = 0;
    int nrange = 0, nnvalid = 0;
    const bool more  = !TAIL && vn < ntiles;
#pragma unroll
    for (int q = 0; q < NWIN; ++q) {
      asm volatile("" : "+v"(tid));
      // ---- keys and rows of window q into LDS
      const int32_t rowt = (int32_t)base + (int32_t)tid + row0;
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        const unsigned int l = (j & 1) ? lpos2[j / 2] >> 16 : lpos2[j / 2] & 0xFFFFu;
        if ((int)(l / WIN) == q) {  // (0xFFFF / WIN = 7 is no window)
          s_k[l % WIN] = key[j];
          s_i[l % WIN] = PAY ? pay[j] : rowt + j * BTt;
        }
      }
      if (q == NWIN - 1 && more) {  // the key registers are free: the next tile's keys land under the write-out
        locate(vn, nbase, nnvalid, nrange);
        load(nbase, nnvalid);
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < RPW; ++j) {
        const int i  = j * BTt + (int)tid;
        const int gi = q * WIN + i;
        if (gi < nvalid) {
          const K k             = s_k[i];
          const int32_t r       = s_i[i];
          const unsigned int pt = part_of(k);
          const unsigned int p  = s_cd[pt] + (unsigned int)gi;
          if (EXACT || p < (unsigned int)(pt * PJ_NR + range + 1) * cap) {  // (a run that outgrew its slot is dropped: `overflow` is up)
            if (AOS) precs[p] = PjRec{(uint32_t)k, (uint32_t)(k >> 32), r};
            else {  // the same pass writing the two arrays of k_pj2_scatter (A/B: is it the record runs or the window structure that pays?)
              reinterpret_cast<K*>(precs)[p] = k;
              soa_idx[p]                     = r;
            }
          }
        }
      }
      __syncthreads();  // the windows and s_cd are read until here
    }
    if (!more) break;
    v       = vn;
    base    = nbase;
    range   = nrange;
    nvalid_ = nnvalid;
  }
}

// After the speculative scatter: piece numbering over the regions, or the verdict "fallback".  One block of 1024 threads,
// each handling a run of consecutive regions.
__global__ void __launch_bounds__(1024) k_pj2_offsets(Pj2Plan* plan2, int pbits, uint32_t cap, unsigned int piece_rows, int xmajor)
{
  __shared__ unsigned int s_tmp[1024 / GX_WAVE + 1];
  const int E   = (1 << pbits) * PJ_NR;
  const int per = E > 1024 ? E / 1024 : 1;
  const int e0  = (int)threadIdx.x * per;
  if (xmajor)  // the pass counted range-major (Pj2Plan::fillx): region order from here on; a thread reads only the entries it wrote
    for (int k = 0; k < per; ++k) {
      const int e = e0 + k;
      if (e < E) plan2->fill[e] = plan2->fillx[(e % PJ_NR) * (E / PJ_NR) + e / PJ_NR];
    }
  if (plan2->overflow) {  // chunk0 / list_chunk0 stay zero (the plan was cleared): the speculative probe takes no piece
    if (threadIdx.x == 0) plan2->fallback = 1u;
    return;
  }
  unsigned int sum = 0;
  for (int k = 0; k < per; ++k) {
    const int e = e0 + k;
    if (e < E) {
      unsigned int c = plan2->fill[e];
      c              = c < cap ? c : cap;
      sum += (c + piece_rows - 1) / piece_rows;
    }
  }
  unsigned int total;
  unsigned int run = block_exclusive_scan<1024>(sum, 0u, SumOp(), s_tmp, &total);
  const int LISTE  = E / PJ_NR;  // regions per XCD list
  for (int k = 0; k < per; ++k) {
    const int e = e0 + k;
    if (e < E) {
      plan2->chunk0[e] = run;
      if (e % LISTE == 0) plan2->list_chunk0[e / LISTE] = run;
      unsigned int c = plan2->fill[e];
      c              = c < cap ? c : cap;
      run += (c + piece_rows - 1) / piece_rows;
    }
  }
  if (threadIdx.x == 0) {
    plan2->chunk0[E]           = total;
    plan2->list_chunk0[PJ_NR]  = total;
  }
}

// where the pieces of a probe launch come from
struct PieceTable {
  const unsigned int* chunk0;        // [regions + 1] first piece of each region
  const unsigned int* list_chunk0;   // [PJ_NR + 1]
  PjCounter* ticket;                 // [PJ_NR]
  const unsigned long long* start;   // exact partitions: row range of region e = [start[e], start[e + 1]); NULL = padded slots
  const unsigned int* fill;          // padded slots: rows in region e (clamped to cap)
  unsigned int cap;                  // padded slots: region e starts at e * cap
  int nr;                            // regions per partition: PJ_NR (padded slots) or 1 (exact partitions)
  // round 6 (k_pj2_probe_pipe, padded slots only): fixedk > 0 = EVERY region is cut into fixedk = ceil(cap / piece rows) pieces, so a
  // ticket t of list y IS (region, piece in region) = (t / fixedk, t % fixedk) -- no search through chunk0 -- and the service wave's
  // ticket atomic and fill-counter read are issued one trip ahead of their use.  `gate` (plan2->fallback): non-zero = no piece at all.
  unsigned int fixedk;
  const unsigned int* gate;
  // round 6 (k_pj2_probe_pipe<LONG>): rows whose chain needs a dependent read are not settled inside the pipelined loop -- they go to the
  // workgroup's slice of an overflow list (ovf + blockIdx.x * ovf_cap, ovf_count[blockIdx.x] rows) and k_pj2_probe_rare settles them
  PjRec* ovf;
  unsigned int ovf_cap;
  unsigned int* ovf_count;
};

// DEFER: a row whose chain is not settled by its first (preloaded) candidate slot -- another slot carries its tag, or the
// chain runs past the 8-slot tag window -- used to make its whole wave wait for a second, dependent L2 round trip inside
// S3.  With keys that hash like random numbers (SURVEY 8d's) about 1 % of the rows are like that, i.e. nearly every
// wave in every trip (256 rows) stalled once: 8.4 ms against 6.4 ms for round 2's low-discrepancy keys.  Such a row is
// now parked in a small per-wave LDS queue {key, row, next slot, first match}; lanes 0..q-1 request the next slot of
// their queue entry right away and look at it in the NEXT trip, where the row is staged like a fifth row of the lane.
constexpr int PP_TAGPAD = 16;  // bytes of tags of the NEXT sub-table kept behind a sub-table's own: 16-slot windows never leave the LDS
constexpr int PP_Q = 8;  // queue entries per wave (expected ~2 per trip; a full queue falls back to the in-place walk)
template <typename K>
struct alignas(8) PpDefer {
  K key;
  uint32_t row_m;  // row index | (first candidate matched) << 31
  uint32_t wb;     // absolute slot of the row's 16-slot tag window
  int32_t first;   // build row of the first match
  uint32_t ended;  // the chain ends inside the window
  uint64_t cand;   // tag candidates still to look at (bit 4i + 3 = slot i of the window), lowest first
};
typedef uint32_t pj_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t pj_u32x3 __attribute__((ext_vector_type(3)));
// REC (round 6): the probe rows are 12-byte records {key, row} (k_pj2_scatter_rec) behind `pkeys`: one 12-byte load per row instead of an
// 8-byte and a 4-byte one.  (Measured and dropped: four CONSECUTIVE rows per lane as three 16-byte loads -- 16 bytes per lane at a 48-byte
// stride touch every line of the wave's 3 KiB three times: probe 7.57 -> 8.8 ms, profiles/r6_run1_join_ab.txt.)
// ABL (measurement only, WRONG results; gx_join_set_experiment bits 4-6): 1 = tag lookup kept, no slot is read; 2 = matches are found but
// not staged / flushed; 3 = no tag lookup at all (rows streamed in, nothing else)
// LONG (round 6, second cut; record form, tables of <= 2^28 slots): every load of the probe waves gets a WHOLE trip between its issue and
// its first use.  Until now a trip ran S3 (compare the slots requested by last trip's S2) -> S2 (tags, request slots) -> S1 (request the
// next rows) -> X(t): the slot reads had S1 + the barrier + the flush to arrive, and all fifteen waves of the gang stalled on them at the
// top of S3 at the same time -- the L2's random reads (0.45 per row, 2.4 ms at the part's rate) added to the kernel instead of hiding
// under its ~140 VALU instructions per row (measured: no slot reads 3.98 ms, full kernel 7.13).  Now a trip is
//     request rows(t) -> S2(t-1): tags, request slots -> S3(t-2): compare the slots requested LAST trip -> X(t) -> flush(t-3)
// with two register sets for the rows and two for the S2 -> S3 state, alternated by unrolling the trip loop twice (a copy between sets
// would wait for the loads it copies).  The memory pipeline returns loads in order, so the rows are requested BEFORE the slots of the
// same trip and S3 waits with both of them still in flight.  Also cheaper per row: record addresses as 32-bit offsets from a scalar
// base, the hash's upper word only, ONE staging reservation per wave and trip instead of one per row, no per-row candidate masks
// carried to S3 (a row with a second tag candidate or a chain beyond its 16-slot window -- ~1 % -- walks its chain from the home slot).
template <typename K, bool EARLY, bool DEFER, bool REC = false, int ABL = 0, bool LONG = false>
__global__ void __launch_bounds__(PP_BT)
k_pj2_probe_pipe(const K* __restrict__ pkeys, const int32_t* __restrict__ pidx, PieceTable pt, int pbits,
                 const Slot<K>* __restrict__ slots, uint32_t log2cap, int left_outer, int32_t* __restrict__ out_probe,
                 int32_t* __restrict__ out_build, int64_t capacity, unsigned long long* cursor)
{
  typedef typename SlotRaw<K>::type Raw;
  constexpr uint32_t SUB = 1u << PJ_SUB_LOG2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t* s_tagw = reinterpret_cast<const uint32_t*>(smem);                       // 64 KiB of tags + the 32 tags that follow them
  int32_t* s_sidx        = reinterpret_cast<int32_t*>(smem + (SUB >> 1) + PP_TAGPAD);      // [3][PP_ROWS] staged probe rows
  int32_t* s_sfirst      = s_sidx + 3 * PP_ROWS;                                           // [3][PP_ROWS] staged build rows
  __shared__ unsigned int s_cnt[8];           // pairs staged by piece (it & 7)
  __shared__ unsigned long long s_base[3];    // output position of staging buffer (it % 3)
  __shared__ PpPiece s_piece[4];              // piece of trip (t & 3), resolved two trips ahead by the service wave
  __shared__ PpDefer<K> s_defer[DEFER ? PP_PW * PP_Q : 1];  // per-wave queues of rows that need another slot
  __shared__ unsigned int s_ovf;              // LONG: rows sent to the workgroup's overflow slice
  const int P         = 1 << pbits;
  const int LISTP     = P / PJ_NR;
  const uint64_t mask = (1ull << log2cap) - 1;
  const unsigned tid  = threadIdx.x;
  const unsigned lane = lane_id();
  const unsigned w    = tid / GX_WAVE;
  const uint8_t* gtags = reinterpret_cast<const uint8_t*>(slots + (mask + 1));

  if (tid < 8) s_cnt[tid] = 0;
  if (tid == 0) s_ovf = 0;

  if (w == PP_PW) {
    // ------------------------------------------------------------------ service wave: tickets and reservations
    const unsigned x0 = pj_xcc();
    unsigned ylist    = 0;  // lists tried so far
    auto take_piece = [&](PpPiece& pc) {
      pc.valid = 0;
      pc.c0 = pc.c1 = 0;
      pc.part = 0;
      unsigned int g = 0xFFFFFFFFu, y = 0;
      if (lane == 0) {
        while (ylist < PJ_NR) {
          y                      = (x0 + ylist) % PJ_NR;
          const unsigned int nch = pt.list_chunk0[y + 1] - pt.list_chunk0[y];
          if (nch) {
            const unsigned int t = atomicAdd(&pt.ticket[y].v, 1u);
            if (t < nch) {
              g = pt.list_chunk0[y] + t;
              break;
            }
          }
          ++ylist;
        }
      }
      g     = (unsigned int)__builtin_amdgcn_readfirstlane((int)g);
      y     = (unsigned int)__builtin_amdgcn_readfirstlane((int)y);
      ylist = (unsigned int)__builtin_amdgcn_readfirstlane((int)ylist);
      if (g == 0xFFFFFFFFu) return;
      // partition of piece g inside list y: the one whose piece interval contains it (LISTP <= 512 entries, 64 lanes)
      unsigned int part = 0;
      bool hit = false;
      for (int e = (int)lane; e < LISTP; e += GX_WAVE) {
        const int p           = (int)y * LISTP + e;
        const unsigned int lo = pt.chunk0[p * pt.nr], hi = pt.chunk0[(p + 1) * pt.nr];
        if (lo <= g && g < hi) {
          part = (unsigned int)p;
          hit  = true;
        }
      }
      uint64_t hb = ballot(hit);
      part        = shfl(part, __builtin_ctzll(hb));
      // its region inside the partition
      unsigned int reg = part * (unsigned int)pt.nr, loc = 0;
      hit = false;
      if ((int)lane < pt.nr) {
        const unsigned int e  = part * (unsigned int)pt.nr + lane;
        const unsigned int lo = pt.chunk0[e], hi = pt.chunk0[e + 1];
        if (lo <= g && g < hi) {
          reg = e;
          loc = g - lo;
          hit = true;
        }
      }
      hb           = ballot(hit);
      const int sr = __builtin_ctzll(hb);
      reg          = shfl(reg, sr);
      loc          = shfl(loc, sr);
      unsigned long long r0, r1;
      if (pt.start) {
        r0 = pt.start[reg];
        r1 = pt.start[reg + 1];
      } else {
        unsigned int c = pt.fill[reg];
        c              = c < pt.cap ? c : pt.cap;
        r0             = (unsigned long long)reg * pt.cap;
        r1             = r0 + c;
      }
      pc.c0    = r0 + (unsigned long long)loc * PP_ROWS;
      pc.c1    = pc.c0 + PP_ROWS < r1 ? pc.c0 + PP_ROWS : r1;
      pc.part  = part;
      pc.valid = 1;
    };
    // Round 6: the PIPELINED take (pt.fixedk > 0).  take_piece above is a chain of four dependent global round trips -- ticket atomic,
    // two searches through chunk0, the region's fill counter -- and the output reservation of the previous piece was waited for at the
    // top of the same trip: five round trips per trip on ONE wave, with fifteen probe waves waiting for it at X(t).  Measured
    // (profiles/r6_run2_join_ab.txt): with the slot reads, or the tag lookups, or both removed the kernel still takes 4.7 ms = 4.9 us
    // per 3840-row trip; every form of the probe that shares this service wave landed at the same 7.5 ms in round 5.  Now a ticket IS
    // its (region, piece): step C issues the ticket atomic, step B (next trip) turns it into a region and requests the fill counter,
    // step A (the trip after) builds the piece -- each step consumes what was issued a whole trip earlier, and the reservation is
    // read back behind them.  A list that runs out costs one empty piece (valid, no rows) while the next list's first ticket travels.
    const unsigned fixedk  = pt.fixedk;
    const bool gated       = fixedk && pt.gate && *pt.gate != 0;
    unsigned int tk_raw    = 0, fill_raw = 0;  // lane 0: ticket / fill counter in flight
    unsigned int tk_y      = 0, pe = 0, ploc = 0, last_part = 0;
    bool tk_pend = false, fill_pend = false;
    auto step_take = [&](PpPiece& pc) {
      pc.valid = 1;  // an empty piece unless step A has a region
      pc.c0 = pc.c1 = 0;
      pc.part = last_part;
      // ---- A: the fill counter requested last trip
      if (fill_pend) {
        unsigned int c = (unsigned int)__builtin_amdgcn_readfirstlane((int)fill_raw);
        c              = c < pt.cap ? c : pt.cap;
        const unsigned long long r0 = (unsigned long long)pe * pt.cap;
        pc.c0     = r0 + (unsigned long long)ploc * PP_ROWS;
        pc.c1     = pc.c0 + PP_ROWS < r0 + c ? pc.c0 + PP_ROWS : r0 + c;
        if (pc.c1 < pc.c0) pc.c1 = pc.c0;  // the region ends before this piece
        pc.part   = pe / (unsigned int)pt.nr;
        last_part = pc.part;
        fill_pend = false;
      } else if (!tk_pend && ylist >= PJ_NR) {
        pc.valid = 0;  // nothing in flight, no list left: the pipeline drains
      }
      // ---- B: the ticket requested last trip
      if (tk_pend) {
        const unsigned int tk  = (unsigned int)__builtin_amdgcn_readfirstlane((int)tk_raw);
        const unsigned int nch = (unsigned int)LISTP * (unsigned int)pt.nr * fixedk;
        tk_pend                = false;
        if (tk < nch) {
          pe        = tk_y * (unsigned int)LISTP * (unsigned int)pt.nr + tk / fixedk;
          ploc      = tk % fixedk;
          if (lane == 0) fill_raw = pt.fill[pe];
          fill_pend = true;
        } else {
          ++ylist;  // this list is exhausted
        }
      }
      // ---- C: the next ticket
      if (ylist < PJ_NR && !gated) {
        tk_y = (x0 + ylist) % PJ_NR;
        if (lane == 0) tk_raw = atomicAdd(&pt.ticket[tk_y].v, 1u);
        tk_pend = true;
      } else {
        ylist = PJ_NR;
      }
    };
    PpPiece pc;
    if (fixedk) {
      step_take(pc);  // (C)
      step_take(pc);  // (B, C)
      step_take(pc);
      if (lane == 0) s_piece[0] = pc;
      step_take(pc);
      if (lane == 0) s_piece[1] = pc;
    } else {
      take_piece(pc);
      if (lane == 0) s_piece[0] = pc;
      take_piece(pc);
      if (lane == 0) s_piece[1] = pc;
    }
    __syncthreads();  // prologue barrier
    unsigned long long pending = 0;
    int done_at = -1;
    unsigned int partC = 0;
    bool tags_loaded   = false;
    for (int t = 0;; ++t) {
      const int itf = t - 3, it3 = t - 2;
      if (t >= 1) {  // R(t): the probe waves reload their tags when the piece entering S2 belongs to another partition
        const PpPiece pa = s_piece[(t - 1) & 3];
        if (pa.valid && (!tags_loaded || pa.part != partC)) {
          partC       = pa.part;
          tags_loaded = true;
          __syncthreads();
        }
      }
      if (done_at < 0 && !s_piece[t & 3].valid) done_at = t;  // same test as the probe waves make
      if (fixedk) {
        step_take(pc);
      } else {
        if (lane == 0 && itf >= 0) s_base[(unsigned)itf % 3u] = pending;
        take_piece(pc);
      }
      if (lane == 0) s_piece[(t + 2) & 3] = pc;
      if (fixedk && lane == 0 && itf >= 0) s_base[(unsigned)itf % 3u] = pending;  // (its atomic left behind X(t - 1): read back last)
      __syncthreads();  // X(t)
      if (done_at >= 0 && t >= done_at + (DEFER ? 4 : 3)) break;  // DEFER: rows parked in the last S3 are staged one trip later
      if (lane == 0) {
        if (it3 >= 0) {
          unsigned int c = s_cnt[(unsigned)it3 & 7u];
          c              = c < (unsigned)PP_ROWS ? c : (unsigned)PP_ROWS;
          pending        = c ? atomicAdd(cursor, (unsigned long long)c) : 0ull;
        }
        s_cnt[(unsigned)(t + 2) & 7u] = 0;
      }
    }
    return;
  }

  // ---------------------------------------------------------------------- probe waves
  __syncthreads();  // prologue barrier: pieces 0 and 1 are resolved
  if constexpr (LONG) {
    static_assert(REC && sizeof(K) == 8 && !EARLY && !DEFER && (ABL == 0 || ABL == 4), "LONG: the record form of 8-byte keys");  // ABL 4 (measurement, WRONG results): no chain walks
    struct RowSet {    // rows of a piece, in flight from the top of trip t to S2 of trip t + 1
      pj_u32x3 r[PP_R];
    };
    struct SlotSet {   // S2 -> S3: the rows, the first candidate slot of each (in flight), per-row flags
      uint32_t klo[PP_R], khi[PP_R];
      int32_t idx[PP_R];
      Raw sv[PP_R];
      Raw sv2;         // the SECOND candidate slot of one of the lane's rows (~1 % of the rows have two tag candidates)
      uint32_t fl;     // bit j: live row; bit 4 + j: it has a first candidate; bit 8 + j: its chain must be walked (S3, rare: ~1e-4);
                       // bit 12: sv2 is the second candidate of row (fl >> 13) & 3
      uint32_t valid;
    };
    RowSet R0, R1;
    SlotSet M0, M1;
    M0.fl = M1.fl = 0;
    M0.valid = M1.valid = 0;
#pragma unroll
    for (int j = 0; j < PP_R; ++j) {
      R0.r[j] = R1.r[j] = pj_u32x3{0u, 0u, 0u};
      M0.klo[j] = M0.khi[j] = M1.klo[j] = M1.khi[j] = 0;
      M0.idx[j] = M1.idx[j] = 0;
      M0.sv[j] = M1.sv[j] = Raw{};
    }
    M0.sv2 = M1.sv2 = Raw{};
    const PjRec* recs  = reinterpret_cast<const PjRec*>(pkeys);
    const uint32_t woff = w * (PP_R * GX_WAVE) + lane;  // this thread's first row inside a piece
    uint32_t partA = 0, cntA = 0, partC = 0;
    bool validA = false, tags_loaded = false;
    int done_at = -1;
    // the whole chain of a row, from its home slot (rare rows only)
    auto walk = [&](K key, uint32_t& mm, int32_t& ff) {
      mm          = 0;
      ff          = NO_MATCH;
      uint64_t hh = slot_of<K>(key, log2cap);
      for (;;) {
        K k;
        int32_t r;
        load_slot<K>(&slots[hh], k, r);
        if (r == EMPTY_ROW) break;
        if (k == key) {
          if (mm == 0) ff = r;
          ++mm;
        }
        hh = (hh + 1) & mask;
      }
    };
    auto trip = [&](const int t, RowSet& Rl, RowSet& Ru, SlotSet& Mn, SlotSet& Mo) __attribute__((always_inline)) -> bool {
      // ---------------- rows of piece t: requested first, looked at by S2 of the next trip
      const PpPiece pcur = s_piece[t & 3];
      const uint32_t pvalid = (uint32_t)__builtin_amdgcn_readfirstlane((int)pcur.valid);
      // (UNCONDITIONAL loads, here and in S2: a load under a wave-uniform `if` does not count as "issued after" for the compiler's
      //  s_waitcnt placement -- it waits for the older loads as if the younger ones were not in flight.  A trip without a piece reads
      //  row 0 of the records / slot 0 of the table and ignores them.)
      uint32_t cntN = 0, partN = 0;
      {
        uint32_t c0l = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pcur.c0);
        uint32_t c0h = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pcur.c0 >> 32));
        cntN         = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pcur.c1 - pcur.c0));
        partN        = (uint32_t)__builtin_amdgcn_readfirstlane((int)pcur.part);
        if (!pvalid) c0l = c0h = cntN = partN = 0;
        const char* base = reinterpret_cast<const char*>(recs + (((unsigned long long)c0h << 32) | c0l));
#pragma unroll
        for (int j = 0; j < PP_R; ++j) {
          const uint32_t o = woff + (uint32_t)j * GX_WAVE;
          const uint32_t b = (o < cntN ? o : 0u) * (uint32_t)sizeof(PjRec);
          Rl.r[j]          = __builtin_nontemporal_load(reinterpret_cast<const pj_u32x3*>(base + b));
        }
      }
      // ---------------- tags of the partition S2 is about to probe
      if (validA && (!tags_loaded || partA != partC)) {
        const uint4* src = reinterpret_cast<const uint4*>(gtags + (((uint64_t)partA << PJ_SUB_LOG2) >> 1));
        uint4* dst       = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = tid; i < SUB / 2 / 16; i += PP_PW * GX_WAVE) dst[i] = src[i];
        if (tid == 0) {  // the 32 tags behind the sub-table's own; behind the LAST sub-table the table wraps to slot 0
          const bool last = (((uint64_t)partA + 1) << PJ_SUB_LOG2) > mask;
          dst[SUB / 2 / 16] = last ? *reinterpret_cast<const uint4*>(gtags) : src[SUB / 2 / 16];
        }
        partC       = partA;
        tags_loaded = true;
        __syncthreads();  // R(t)
      }
      // ---------------- S2(t-1): chain heads on the LDS tags, first candidate slot requested
      Mn.fl    = 0;
      Mn.valid = validA ? 1u : 0u;
      {  // (cntA = 0 without a piece: no live row)
        const uint32_t sub_lo   = partA << PJ_SUB_LOG2;                      // (log2cap <= 28: a slot number is 32 bits)
        const bool last         = (((uint64_t)partA + 1) << PJ_SUB_LOG2) > mask;
        const char* sbase       = reinterpret_cast<const char*>(slots + ((uint64_t)partA << PJ_SUB_LOG2));
        const uint32_t sh_slot  = 32u - log2cap, sh_tag = 28u - log2cap;
        uint32_t soff[PP_R];
        uint32_t soff2 = 0;
#pragma unroll
        for (int j = 0; j < PP_R; ++j) {
          // (a COPY the compiler cannot see through: the rows' registers are free for the next request from here on, and the copy is made
          //  HERE, where the rows are needed anyway.  Left to itself the compiler keeps one value and rotates the register sets with
          //  copies at the loop's back edge -- copies of registers whose loads are still in flight: a full wait every second trip.)
          asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5"
                       : "=&v"(Mn.klo[j]), "=&v"(Mn.khi[j]), "=&v"(Mn.idx[j])
                       : "v"(Ru.r[j].x), "v"(Ru.r[j].y), "v"(Ru.r[j].z));
          soff[j]   = 0;
          if (woff + (uint32_t)j * GX_WAVE < cntA) {
            const uint64_t key = ((uint64_t)Mn.khi[j] << 32) | Mn.klo[j];
            const uint32_t hi  = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 32);
            const uint32_t li  = (hi >> sh_slot) - sub_lo;
            uint32_t tg        = (hi >> sh_tag) & 15u;
            tg                 = tg ? tg : 8u;
            const uint32_t tagpat = tg * 0x11111111u;
            const uint32_t w0 = s_tagw[li >> 3], w1 = s_tagw[(li >> 3) + 1], w2 = s_tagw[(li >> 3) + 2];
            const uint32_t sh = (li & 7u) * 4u;
            const uint32_t x0 = __builtin_amdgcn_alignbit(w1, w0, sh), x1 = __builtin_amdgcn_alignbit(w2, w1, sh);
            const uint32_t y0 = x0 ^ tagpat, y1 = x1 ^ tagpat;
            const uint32_t z0 = ~(((x0 & 0x77777777u) + 0x77777777u) | x0) & 0x88888888u;  // empty slots
            const uint32_t z1 = ~(((x1 & 0x77777777u) + 0x77777777u) | x1) & 0x88888888u;
            const uint32_t m0 = ~(((y0 & 0x77777777u) + 0x77777777u) | y0) & 0x88888888u;  // tag matches
            const uint32_t m1 = ~(((y1 & 0x77777777u) + 0x77777777u) | y1) & 0x88888888u;
            const uint32_t c0 = m0 & ((z0 & (0u - z0)) - 1u);                               // ... below the first empty one
            const uint32_t c1 = z0 ? 0u : (m1 & ((z1 & (0u - z1)) - 1u));
            const uint32_t nc = (uint32_t)__builtin_popcount(c0) + (uint32_t)__builtin_popcount(c1);
            uint32_t f        = 1u << j;
            if (nc) {
              f |= 1u << (4 + j);
              const uint32_t first = c0 ? ((uint32_t)__builtin_ctz(c0) >> 2) : 8u + ((uint32_t)__builtin_ctz(c1) >> 2);
              soff[j]              = li + first;
            }
            // More than the first candidate.  A dependent read in S3 would wait for EVERY load in flight (they return in order),
            // i.e. empty the pipeline this loop exists to keep full: 92 % of the wave-trips hold such a row.  So the common case --
            // exactly two candidates, the chain ends inside the window, the lane's first such row of the trip -- gets its second
            // slot requested here as well; what remains (~1e-4 of the rows) walks its chain in S3.
            if (nc > 1 || (z0 | z1) == 0) {
              if (nc == 2 && (z0 | z1) != 0 && !((Mn.fl >> 12) & 1u)) {
                const uint32_t d0 = c0 & (c0 - 1u);                 // (c1:c0) without its lowest candidate
                const uint32_t d1 = c0 ? c1 : (c1 & (c1 - 1u));
                soff2 = li + (d0 ? ((uint32_t)__builtin_ctz(d0) >> 2) : 8u + ((uint32_t)__builtin_ctz(d1) >> 2));
                f |= (1u << 12) | ((uint32_t)j << 13);
              } else {
                f |= 1u << (8 + j);
              }
            }
            Mn.fl |= f;
          }
        }
        if (!last) {
#pragma unroll
          for (int j = 0; j < PP_R; ++j) Mn.sv[j] = *reinterpret_cast<const Raw*>(sbase + soff[j] * (uint32_t)sizeof(Slot<K>));
          Mn.sv2 = *reinterpret_cast<const Raw*>(sbase + soff2 * (uint32_t)sizeof(Slot<K>));
        } else {  // the table's last sub-table: a window may wrap to slot 0
#pragma unroll
          for (int j = 0; j < PP_R; ++j) Mn.sv[j] = *reinterpret_cast<const Raw*>(&slots[((uint64_t)sub_lo + soff[j]) & mask]);
          Mn.sv2 = *reinterpret_cast<const Raw*>(&slots[((uint64_t)sub_lo + soff2) & mask]);
        }
      }
      // ---------------- S3(t-2): compare the slots requested last trip, stage the matches
      const int it3 = t - 2;
      {  // (Mo.fl = 0 without a piece)
        const unsigned buf = (unsigned)(it3 + 3) % 3u;
        uint32_t m[PP_R];
        int32_t first[PP_R];
        bool multi = false;
#pragma unroll
        for (int j = 0; j < PP_R; ++j) {
          // (compared as two words: a 64-bit compare wants an aligned register pair, and the pair was put together by copies at the
          //  loop's back edge -- of registers whose loads are still in flight)
          const K key = ((uint64_t)Mo.khi[j] << 32) | Mo.klo[j];
          m[j]        = ((Mo.fl >> (4 + j)) & 1u) && Mo.sv[j].x == Mo.klo[j] && Mo.sv[j].y == Mo.khi[j] ? 1u : 0u;
          first[j]    = m[j] ? (int32_t)Mo.sv[j].z : NO_MATCH;
          if (((Mo.fl & 0x7000u) == (0x1000u | ((uint32_t)j << 13))) & (Mo.sv2.x == Mo.klo[j]) & (Mo.sv2.y == Mo.khi[j])) {  // (no short cut: sv2 is LOOKED AT on every path)
            if (m[j] == 0) first[j] = (int32_t)Mo.sv2.z;
            ++m[j];
          }
          bool settled = true;
          if (ABL != 4 && ((Mo.fl >> (8 + j)) & 1u)) {  // rare (~1e-3 of the rows): three or more candidates, a chain beyond the window, ...
            // NOT walked here: a dependent read waits for every load in flight (this trip's rows and slots included) -- with these rows
            // walked in place the kernel took 8.9 ms, without them 5.0 (profiles/r6_run36_join_ab.txt).  The row goes to the workgroup's
            // slice of the overflow list (position from an LDS counter, a plain store) and k_pj2_probe_rare settles it afterwards.
            const unsigned int op = pt.ovf ? atomicAdd(&s_ovf, 1u) : 0xFFFFFFFFu;
            if (op < pt.ovf_cap) {
              PjRec rr;
              rr.klo = Mo.klo[j];
              rr.khi = Mo.khi[j];
              rr.row = Mo.idx[j];
              pt.ovf[(size_t)blockIdx.x * pt.ovf_cap + op] = rr;
              m[j]     = 0;
              first[j] = NO_MATCH;
              settled  = false;
            } else {  // the slice is full (or there is none): in place
              walk(key, m[j], first[j]);
            }
          }
          if (left_outer && settled && ((Mo.fl >> j) & 1u) && m[j] == 0) m[j] = 1;  // (row, JoinNoMatch); first[j] is NO_MATCH
          multi |= m[j] > 1;
        }
        if (ballot(multi) == 0) {  // the common case: at most one pair per row -- one reservation for the wave's four rows
          uint64_t bb[PP_R];
          uint32_t tot = 0;
#pragma unroll
          for (int j = 0; j < PP_R; ++j) {
            bb[j] = ballot(m[j] != 0);
            tot += (uint32_t)__builtin_popcountll(bb[j]);
          }
          if (tot) {
            uint32_t wbase = 0;
            if (lane == 0) wbase = atomicAdd(&s_cnt[(unsigned)it3 & 7u], tot);
            wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
#pragma unroll
            for (int j = 0; j < PP_R; ++j) {
              const uint32_t pos = wbase + (uint32_t)__builtin_popcountll(bb[j] & lanemask_lt());
              wbase += (uint32_t)__builtin_popcountll(bb[j]);
              if (m[j]) {
                if (pos < (uint32_t)PP_ROWS) {
                  s_sidx[buf * PP_ROWS + pos]   = Mo.idx[j];
                  s_sfirst[buf * PP_ROWS + pos] = first[j];
                } else {  // staging full (another wave staged duplicates): reserve and write directly.  (an atomic store: an ordinary one is
                          // merged with the LDS store of the other branch into ONE flat store through a selected pointer -- and a pending FLAT
                          // access makes the compiler wait with vmcnt(0) for EVERY load of the kernel: no load stays in flight across a use)
                  const unsigned long long gp = atomicAdd(cursor, 1ull);
                  if ((int64_t)gp < capacity) {
                    __hip_atomic_store(&out_probe[gp], (int32_t)(Mo.idx[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&out_build[gp], (int32_t)(first[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  }
                }
              }
            }
          }
        } else {  // duplicate build keys somewhere in the wave: per-row reservations, chains walked again
#pragma unroll
          for (int j = 0; j < PP_R; ++j) {
            if (ballot(m[j] != 0) == 0) continue;
            const uint32_t sc  = wave_inclusive_scan(m[j], SumOp());
            const uint32_t off = sc - m[j];
            const uint32_t tot = shfl(sc, GX_WAVE - 1);
            uint32_t wbase     = 0;
            if (lane == 0) wbase = atomicAdd(&s_cnt[(unsigned)it3 & 7u], tot);
            wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
            if (m[j] == 0) continue;
            const uint32_t pos = wbase + off;
            const K rkey       = ((uint64_t)Mo.khi[j] << 32) | Mo.klo[j];
            const int32_t ridx = Mo.idx[j];
            if (m[j] == 1) {
              if (pos < (uint32_t)PP_ROWS) {
                s_sidx[buf * PP_ROWS + pos]   = ridx;
                s_sfirst[buf * PP_ROWS + pos] = first[j];
              } else {
                const unsigned long long gp = atomicAdd(cursor, 1ull);
                if ((int64_t)gp < capacity) {
                  __hip_atomic_store(&out_probe[gp], (int32_t)(ridx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  __hip_atomic_store(&out_build[gp], (int32_t)(first[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
              }
            } else {
              const uint32_t room   = pos < (uint32_t)PP_ROWS ? (uint32_t)PP_ROWS - pos : 0u;
              const uint32_t staged = room < m[j] ? room : m[j];
              unsigned long long gp = 0;
              if (staged < m[j]) gp = atomicAdd(cursor, (unsigned long long)(m[j] - staged));
              uint32_t seen = 0;
              uint64_t hh   = slot_of<K>(rkey, log2cap);
              for (;;) {
                K k;
                int32_t r;
                load_slot<K>(&slots[hh], k, r);
                if (r == EMPTY_ROW) break;
                if (k == rkey) {
                  if (seen < staged) {
                    s_sidx[buf * PP_ROWS + pos + seen]   = ridx;
                    s_sfirst[buf * PP_ROWS + pos + seen] = r;
                  } else {
                    if ((int64_t)gp < capacity) {
                      __hip_atomic_store(&out_probe[gp], (int32_t)(ridx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                      __hip_atomic_store(&out_build[gp], (int32_t)(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    ++gp;
                  }
                  ++seen;
                }
                hh = (hh + 1) & mask;
              }
            }
          }
        }
      }
      // ---------------- the piece that enters S2 next trip
      validA = pvalid != 0;
      if (done_at < 0 && !validA) done_at = t;
      partA = validA ? partN : partA;  // (an empty trip keeps the tags it has)
      cntA  = cntN;
      __syncthreads();  // X(t): staging of piece t-2 complete, s_base of piece t-3 visible
      if (done_at >= 0 && t >= done_at + 3) return true;
      // ---------------- flush of piece t-3: two coalesced streams
      const int itf = t - 3;
      if (itf >= 0) {
        const unsigned buf          = (unsigned)itf % 3u;
        unsigned int c              = s_cnt[(unsigned)itf & 7u];
        c                           = c < (unsigned)PP_ROWS ? c : (unsigned)PP_ROWS;
        const unsigned long long gb = s_base[buf];
        for (unsigned int i = tid; i < c; i += PP_PW * GX_WAVE) {
          const unsigned long long gp = gb + i;
          if ((int64_t)gp < capacity) {
            __builtin_nontemporal_store(s_sidx[buf * PP_ROWS + i], &out_probe[gp]);
            __builtin_nontemporal_store(s_sfirst[buf * PP_ROWS + i], &out_build[gp]);
          }
        }
      }
      return false;
    };
    for (int t = 0;; t += 2) {
      if (trip(t, R0, R1, M0, M1)) break;      // rows(t) -> R0; S2 reads R1 (rows t-1) -> M0; S3 reads M1 (piece t-2)
      if (trip(t + 1, R1, R0, M1, M0)) break;
    }
    // (every S3 lies before the barrier the loop was left behind)
    if (tid == 0 && pt.ovf_count) pt.ovf_count[blockIdx.x] = s_ovf < pt.ovf_cap ? s_ovf : pt.ovf_cap;
    return;
  }
  K kN[PP_R];                // EARLY: loads of trip t, issued at its top
  int32_t iN[PP_R];
  K kA[PP_R];                // S1 -> S2
  int32_t iA[PP_R];
  K kB[PP_R];                // S2 -> S3
  int32_t iB[PP_R];
  uint32_t li[PP_R];
  uint64_t cand[PP_R];       // tag candidates of the row's 16-slot window (nibble top bits)
  Raw sv[PP_R];
  uint32_t fl = 0;           // per row j: bit j = live row, bit 4+j = chain ends inside the window
  uint32_t partA = 0, partB = 0, partC = 0;
  bool tags_loaded = false;
  unsigned long long cA0 = 0, cA1 = 0;
  bool validA = false, validB = false;
  PpDefer<K>* myq = &s_defer[DEFER ? w * PP_Q : 0];
  uint32_t qn = 0;           // entries of this wave's queue whose next slot is in flight in `qsv` (wave-uniform)
  Raw qsv = Raw{};
#pragma unroll
  for (int j = 0; j < PP_R; ++j) {
    kA[j] = kB[j] = kN[j] = K(0);
    iA[j] = iB[j] = iN[j] = 0;
    li[j] = cand[j] = 0;
    sv[j] = Raw{};
  }
  int done_at = -1;

  for (int t = 0;; ++t) {
    // ---------------- S1(t), EARLY form: request the piece's rows first; they are not looked at before the next trip
    PpPiece pcur = s_piece[t & 3];
    if (EARLY && pcur.valid) {
      const unsigned long long pb = pcur.c0 + (unsigned long long)w * (PP_R * GX_WAVE) + lane;
#pragma unroll
      for (int j = 0; j < PP_R; ++j) {
        const unsigned long long i  = pb + (unsigned long long)j * GX_WAVE;
        const unsigned long long ic = i < pcur.c1 ? i : pcur.c0;
        kN[j] = __builtin_nontemporal_load(&pkeys[ic]);
        iN[j] = __builtin_nontemporal_load(&pidx[ic]);
      }
    }
    // ---------------- S3(t-2): compare, settle or park unsettled rows, stage the matches
    const int it3 = t - 2;
    if (validB || (DEFER && qn)) {
      const unsigned buf      = (unsigned)it3 % 3u;
      const uint64_t sub_base = (uint64_t)partB << PJ_SUB_LOG2;
      const uint32_t* gtagw   = reinterpret_cast<const uint32_t*>(gtags);  // the tags of the WHOLE table, by absolute slot
      uint32_t m[PP_R + 1];
      int32_t first[PP_R + 1];
      // Finish a chain in place (every load waits): the remaining tag candidates of the 8-slot window at absolute slot
      // `wb`, then further windows on the tags in global memory.  Rare by construction: the first candidate of a row was
      // preloaded by S2, a parked row's second one by last trip's S3.
      auto finish = [&](K key, uint64_t wb, uint64_t cnd64, bool ended, uint32_t& mm, int32_t& ff) {
        const uint32_t tagpat = tag_of<K>(key, log2cap) * 0x11111111u;
        while (cnd64) {  // what is left of the row's first (16-slot) window
          K k;
          int32_t r;
          load_slot<K>(&slots[(wb + ((uint32_t)__builtin_ctzll(cnd64) >> 2)) & mask], k, r);
          if (k == key) {  // a tagged slot is never empty
            if (mm == 0) ff = r;
            ++mm;
          }
          cnd64 &= cnd64 - 1;
        }
        wb = (wb + 16) & mask;
        while (!ended) {  // further windows of 8 slots, tags from global memory
          if (wb + 8 > mask) {  // the window would wrap around the table's end: walk the slots themselves
            for (;;) {
              K k;
              int32_t r;
              load_slot<K>(&slots[wb & mask], k, r);
              if (r == EMPTY_ROW) break;
              if (k == key) {
                if (mm == 0) ff = r;
                ++mm;
              }
              ++wb;
            }
            break;
          }
          uint32_t cnd = 0;
          ended        = scan_tags8_global(gtagw, (uint32_t)wb, tagpat, cnd);
          while (cnd) {
            K k;
            int32_t r;
            load_slot<K>(&slots[wb + ((uint32_t)__builtin_ctz(cnd) >> 2)], k, r);
            if (k == key) {
              if (mm == 0) ff = r;
              ++mm;
            }
            cnd &= cnd - 1;
          }
          wb += 8;
        }
      };
      // ---- (0) the rows parked last trip: their second candidate slot has arrived
      K keyx       = K(0);
      int32_t idxx = 0;
      bool livex   = false;
      m[PP_R]      = 0;
      first[PP_R]  = NO_MATCH;
      if (DEFER && qn) {  // wave-uniform
        if (lane < qn) {
          const PpDefer<K> e = myq[lane];
          keyx               = e.key;
          idxx               = (int32_t)(e.row_m & 0x7FFFFFFFu);
          livex              = true;
          if (e.row_m >> 31) {
            m[PP_R]     = 1;
            first[PP_R] = e.first;
          }
          K k;
          int32_t r;
          unpack_slot(qsv, k, r);
          if (k == keyx) {
            if (m[PP_R] == 0) first[PP_R] = r;
            ++m[PP_R];
          }
          const uint64_t cnd = e.cand & (e.cand - 1);  // the second candidate is the slot in `qsv`
          const bool ended   = e.ended != 0;
          if (cnd || !ended) finish(keyx, e.wb, cnd, ended, m[PP_R], first[PP_R]);  // a third candidate / a 16+ chain: rare
        }
      }
      // ---- (1) this trip's rows
      uint32_t qnew   = 0;  // wave-uniform
      uint32_t parked = 0;  // rows of this lane that went to the queue: they emit next trip
      if (validB) {
#pragma unroll
        for (int j = 0; j < PP_R; ++j) {
          m[j]     = 0;
          first[j] = NO_MATCH;
          K sk;
          int32_t sr;
          unpack_slot(sv[j], sk, sr);
          bool more  = false;
          bool ended = true;
          if ((fl >> j) & 1u) {
            ended = (fl >> (4 + j)) & 1u;
            if (cand[j]) {
              if (sk == kB[j]) {  // a tagged slot is never empty
                m[j]     = 1;
                first[j] = sr;
              }
              cand[j] &= cand[j] - 1;
            }
            more = cand[j] != 0 || !ended;
          }
          const uint64_t mb = ballot(more);
          if (mb == 0) continue;  // wave-uniform: the common case
          // park the rows that have another tag candidate in their window (its slot is requested below and looked at next
          // trip); a chain that merely runs past the window (no candidate left) is finished in place -- rarer still
          const uint64_t pb = ballot(more && cand[j] != 0);
          const uint32_t nq = (uint32_t)__builtin_popcountll(pb);
          const bool fits   = DEFER && nq > 0 && qnew + nq <= (uint32_t)PP_Q;
          if (fits && more && cand[j] != 0) {
            PpDefer<K> e;
            e.key   = kB[j];
            e.row_m = (uint32_t)iB[j] | (m[j] << 31);
            e.wb    = (uint32_t)(sub_base + li[j]);
            e.first = first[j];
            e.ended = ended ? 1u : 0u;
            e.cand  = cand[j];
            myq[qnew + (uint32_t)__builtin_popcountll(pb & lanemask_lt())] = e;
            m[j] = 0;
            parked |= 1u << j;
          } else if (more) {
            finish(kB[j], sub_base + li[j], cand[j], ended, m[j], first[j]);
          }
          if (fits) qnew += nq;
        }
      } else {
#pragma unroll
        for (int j = 0; j < PP_R; ++j) {
          m[j]     = 0;
          first[j] = NO_MATCH;
        }
      }
      // ---- (2) request the second candidate slot of the rows just parked: looked at in the next trip
      if (DEFER) {
        __builtin_amdgcn_wave_barrier();
        qn = qnew;
        if (lane < qn) {
          const PpDefer<K> e = myq[lane];
          qsv = *reinterpret_cast<const Raw*>(&slots[((uint64_t)e.wb + ((uint32_t)__builtin_ctzll(e.cand) >> 2)) & mask]);
        }
      }
      // ---- (3) stage: pairs go to s_sidx / s_sfirst [buf] at positions handed out by an LDS counter
#pragma unroll
      for (int j = 0; j <= PP_R; ++j) {
        if (j == PP_R && !DEFER) break;
        const bool live    = j < PP_R ? (validB && ((fl >> j) & 1u) && !((parked >> j) & 1u)) : livex;
        const K rkey       = j < PP_R ? kB[j < PP_R ? j : 0] : keyx;
        const int32_t ridx = j < PP_R ? iB[j < PP_R ? j : 0] : idxx;
        if (left_outer && live && m[j] == 0) m[j] = 1;  // (row, JoinNoMatch); first[j] is NO_MATCH
        if (ABL == 2) {
          asm volatile("" ::"v"(m[j]), "v"(first[j]));
          m[j] = 0;
        }
        uint32_t off, tot;
        if (ballot(m[j] > 1) == 0) {
          const uint64_t bb = ballot(m[j] == 1);
          if (bb == 0) continue;
          off = (uint32_t)__builtin_popcountll(bb & lanemask_lt());
          tot = (uint32_t)__builtin_popcountll(bb);
        } else {
          const uint32_t sc = wave_inclusive_scan(m[j], SumOp());
          off               = sc - m[j];
          tot               = shfl(sc, GX_WAVE - 1);
        }
        uint32_t wbase = 0;
        if (lane == 0) wbase = atomicAdd(&s_cnt[(unsigned)it3 & 7u], tot);
        wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
        if (m[j] == 0) continue;
        const uint32_t pos = wbase + off;
        if (m[j] == 1) {
          if (pos < (uint32_t)PP_ROWS) {
            s_sidx[buf * PP_ROWS + pos]   = ridx;
            s_sfirst[buf * PP_ROWS + pos] = first[j];
          } else {  // staging full (duplicate build keys): reserve and write directly
            const unsigned long long gp = atomicAdd(cursor, 1ull);
            if ((int64_t)gp < capacity) {
              __hip_atomic_store(&out_probe[gp], (int32_t)(ridx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(&out_build[gp], (int32_t)(first[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        } else {  // duplicate build keys: walk the chain again (lines are cache-resident)
          const uint32_t room   = pos < (uint32_t)PP_ROWS ? (uint32_t)PP_ROWS - pos : 0u;
          const uint32_t staged = room < m[j] ? room : m[j];
          unsigned long long gp = 0;
          if (staged < m[j]) gp = atomicAdd(cursor, (unsigned long long)(m[j] - staged));
          uint32_t seen = 0;
          uint64_t hh   = slot_of<K>(rkey, log2cap);
          for (;;) {
            K k;
            int32_t r;
            load_slot<K>(&slots[hh], k, r);
            if (r == EMPTY_ROW) break;
            if (k == rkey) {
              if (seen < staged) {
                s_sidx[buf * PP_ROWS + pos + seen]   = ridx;
                s_sfirst[buf * PP_ROWS + pos + seen] = r;
              } else {
                if ((int64_t)gp < capacity) {
                  __hip_atomic_store(&out_probe[gp], (int32_t)(ridx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  __hip_atomic_store(&out_build[gp], (int32_t)(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                ++gp;
              }
              ++seen;
            }
            hh = (hh + 1) & mask;
          }
        }
      }
    }
    // ---------------- tags of the partition S2 is about to probe
    if (validA && (!tags_loaded || partA != partC)) {
      const uint4* src = reinterpret_cast<const uint4*>(gtags + (((uint64_t)partA << PJ_SUB_LOG2) >> 1));
      uint4* dst       = reinterpret_cast<uint4*>(smem);
      for (uint32_t i = tid; i < SUB / 2 / 16; i += PP_PW * GX_WAVE) dst[i] = src[i];
      if (tid == 0) {  // the 32 tags behind the sub-table's own; behind the LAST sub-table the table wraps to slot 0
        const bool last = (((uint64_t)partA + 1) << PJ_SUB_LOG2) > mask;
        dst[SUB / 2 / 16] = last ? *reinterpret_cast<const uint4*>(gtags) : src[SUB / 2 / 16];
      }
      partC       = partA;
      tags_loaded = true;
      __syncthreads();  // R(t)
    }
    // ---------------- S2(t-1): chain heads on the LDS tags, first candidate slot in flight
    fl     = 0;
    validB = validA;
    partB  = partA;
    if (validA) {
      const uint64_t sub_base = (uint64_t)partA << PJ_SUB_LOG2;
      const Slot<K>* dummy    = slots + sub_base;
      const unsigned long long pb = cA0 + (unsigned long long)w * (PP_R * GX_WAVE) + lane;
#pragma unroll
      for (int j = 0; j < PP_R; ++j) {
        kB[j]   = kA[j];
        iB[j]   = iA[j];
        cand[j] = 0;
        li[j]   = 0;
        if (pb + (unsigned long long)j * GX_WAVE < cA1) {
          fl |= 1u << j;
          const uint64_t prod = (uint64_t)kB[j] * 0x9E3779B97F4A7C15ull;
          li[j]               = (uint32_t)((prod >> (64 - log2cap)) - sub_base);
          uint32_t tg         = (uint32_t)(prod >> (60 - log2cap)) & 15u;
          tg                  = tg ? tg : 8u;
          // the 16 tags of local slots [li, li + 16): three words, two funnel shifts.  A chain of the table (load <= 0.5)
          // that is not over after 16 slots is a 1e-5 event; after 8 it is not (3e-3 per row: one stalled wave per trip).
          const uint32_t tagpat = tg * 0x11111111u;
          const uint32_t w0 = ABL == 3 ? li[j] : s_tagw[li[j] >> 3], w1 = ABL == 3 ? tg : s_tagw[(li[j] >> 3) + 1], w2 = ABL == 3 ? 0u : s_tagw[(li[j] >> 3) + 2];
          const uint32_t sh = (li[j] & 7u) * 4u;
          const uint32_t x0 = __builtin_amdgcn_alignbit(w1, w0, sh), x1 = __builtin_amdgcn_alignbit(w2, w1, sh);
          const uint32_t y0 = x0 ^ tagpat, y1 = x1 ^ tagpat;
          const uint32_t z0 = ~(((x0 & 0x77777777u) + 0x77777777u) | x0) & 0x88888888u;  // empty slots
          const uint32_t z1 = ~(((x1 & 0x77777777u) + 0x77777777u) | x1) & 0x88888888u;
          const uint32_t m0 = ~(((y0 & 0x77777777u) + 0x77777777u) | y0) & 0x88888888u;  // tag matches
          const uint32_t m1 = ~(((y1 & 0x77777777u) + 0x77777777u) | y1) & 0x88888888u;
          const uint32_t c0 = m0 & ((z0 & (0u - z0)) - 1u);                               // ... below the first empty one
          const uint32_t c1 = z0 ? 0u : (m1 & ((z1 & (0u - z1)) - 1u));
          cand[j]           = (uint64_t)c0 | ((uint64_t)c1 << 32);
          if (z0 | z1) fl |= 1u << (4 + j);
          if (ABL == 1 || ABL == 3) {
            uint32_t a0 = c0, a1 = c1;
            asm volatile("" ::"v"(a0), "v"(a1));
            cand[j] = 0;
            fl |= 1u << (4 + j);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < PP_R; ++j) {
        const Slot<K>* sp = cand[j] ? slots + ((sub_base + li[j] + ((uint32_t)__builtin_ctzll(cand[j]) >> 2)) & mask) : dummy;
        sv[j]             = *reinterpret_cast<const Raw*>(sp);  // (measured and dropped: nontemporal slot reads -- the L1 does help: probe 7.55 -> 9.34 ms, profiles/r6_run16_join_ab.txt)
      }
    }
    // ---------------- S1(t): the piece that enters S2 next trip
    validA = pcur.valid != 0;
    if (done_at < 0 && !validA) done_at = t;
    if (validA) {
      partA = pcur.part;
      cA0   = pcur.c0;
      cA1   = pcur.c1;
      if constexpr (REC) {
        static_assert(sizeof(K) == 8 && !EARLY, "record form: 8-byte keys, rows requested at the end of a trip");
        const PjRec* recs           = reinterpret_cast<const PjRec*>(pkeys);
        const unsigned long long pb = cA0 + (unsigned long long)w * (PP_R * GX_WAVE) + lane;
#pragma unroll
        for (int j = 0; j < PP_R; ++j) {  // one 12-byte load per row: a wave reads 768 contiguous bytes per instruction
          const unsigned long long i  = pb + (unsigned long long)j * GX_WAVE;
          const unsigned long long ic = i < cA1 ? i : cA0;
          const pj_u32x3 r = __builtin_nontemporal_load(reinterpret_cast<const pj_u32x3*>(recs + ic));
          kA[j] = ((uint64_t)r.y << 32) | r.x;
          iA[j] = (int32_t)r.z;
        }
      } else if (!EARLY) {
        const unsigned long long pb = cA0 + (unsigned long long)w * (PP_R * GX_WAVE) + lane;
#pragma unroll
        for (int j = 0; j < PP_R; ++j) {
          const unsigned long long i  = pb + (unsigned long long)j * GX_WAVE;
          const unsigned long long ic = i < cA1 ? i : cA0;
          kA[j] = __builtin_nontemporal_load(&pkeys[ic]);
          iA[j] = __builtin_nontemporal_load(&pidx[ic]);
        }
      }
    }
    __syncthreads();  // X(t): staging of piece t-2 complete, s_base of piece t-3 visible
    if (done_at >= 0 && t >= done_at + (DEFER ? 4 : 3)) break;
    // ---------------- flush of piece t-3: two coalesced streams
    const int itf = t - 3;
    if (itf >= 0) {
      const unsigned buf          = (unsigned)itf % 3u;
      unsigned int c              = s_cnt[(unsigned)itf & 7u];
      c                           = c < (unsigned)PP_ROWS ? c : (unsigned)PP_ROWS;
      const unsigned long long gb = s_base[buf];
      for (unsigned int i = tid; i < c; i += PP_PW * GX_WAVE) {
        const unsigned long long gp = gb + i;
        if ((int64_t)gp < capacity) {
          __builtin_nontemporal_store(s_sidx[buf * PP_ROWS + i], &out_probe[gp]);
          __builtin_nontemporal_store(s_sfirst[buf * PP_ROWS + i], &out_build[gp]);
        }
      }
    }
    if (EARLY && validA) {  // the rows requested at the top of the trip: into the S1 -> S2 registers
#pragma unroll
      for (int j = 0; j < PP_R; ++j) {
        kA[j] = kN[j];
        iA[j] = iN[j];
      }
    }
  }
}

// The rows k_pj2_probe_pipe<LONG> did not settle (PieceTable::ovf): workgroup b walks the chains of slice b from their home slots and
// emits the pairs -- a count walk, ONE reservation per wave, a write walk.  ~1e-3 of the probe rows, read from wherever the table lies.
constexpr unsigned int PR_SUB = 8;  // workgroups per slice
constexpr int PR_U            = 4;  // rows per thread and round
template <typename K>
__global__ void __launch_bounds__(256) k_pj2_probe_rare(const PjRec* __restrict__ ovf, unsigned int ovf_cap, const unsigned int* __restrict__ ovf_count,
                                                        const Slot<K>* __restrict__ slots, uint32_t log2cap, int left_outer,
                                                        int32_t* __restrict__ out_probe, int32_t* __restrict__ out_build, int64_t capacity,
                                                        unsigned long long* cursor)
{
  // PR_SUB workgroups per slice: the walks are chains of dependent reads from HBM -- what hides them is waves in flight
  const uint64_t mask    = (1ull << log2cap) - 1;
  const unsigned int sl  = blockIdx.x / PR_SUB, sub = blockIdx.x % PR_SUB;
  const unsigned int cnt = ovf_count[sl];
  const PjRec* mine      = ovf + (size_t)sl * ovf_cap;
  // A chain is a run of adjacent 16-byte slots: FOUR slots are requested at once (one or two lines).  Each thread walks PR_U rows and keeps
  // their match counts and first matches in registers; the workgroup then makes ONE reservation for all of them (with one returning atomic
  // per wave and row -- 6e4 on the cursor's address inside 0.2 ms of work -- the kernel took 0.8 ms whatever its walks cost) and only rows
  // with several matches (duplicate build keys) walk a second time.
  auto walk4 = [&](K key, auto&& on_match) {
    uint64_t hh = slot_of<K>(key, log2cap);
    for (;;) {
      K k[4];
      int32_t br[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) load_slot<K>(&slots[(hh + q) & mask], k[q], br[q]);
      bool end = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!end && br[q] == EMPTY_ROW) end = true;
        if (!end && k[q] == key) on_match(br[q]);
      }
      if (end) break;
      hh = (hh + 4) & mask;
    }
  };
  __shared__ unsigned long long s_tmp[256 / GX_WAVE];
  __shared__ unsigned long long s_gbase;
  for (unsigned int r0 = sub * (256u * PR_U); r0 < cnt; r0 += 256u * PR_U * PR_SUB) {  // (block-uniform)
    PjRec rec[PR_U];
    uint32_t m[PR_U];
    int32_t first[PR_U];
    unsigned long long mine_tot = 0;
#pragma unroll
    for (int u = 0; u < PR_U; ++u) {
      const unsigned int i = r0 + (unsigned int)u * 256u + threadIdx.x;
      m[u]     = 0;
      first[u] = NO_MATCH;
      rec[u]   = PjRec{0u, 0u, 0};
      if (i < cnt) {
        rec[u]      = mine[i];
        const K key = (K)(((uint64_t)rec[u].khi << 32) | rec[u].klo);
        walk4(key, [&](int32_t br) {
          if (m[u] == 0) first[u] = br;
          ++m[u];
        });
        if (left_outer && m[u] == 0) m[u] = 1;  // (row, JoinNoMatch)
        mine_tot += m[u];
      }
    }
    unsigned long long total = 0;
    unsigned long long gp    = block_exclusive_scan<256>(mine_tot, 0ull, SumOp(), s_tmp, &total);
    if (threadIdx.x == 0) s_gbase = total ? atomicAdd(cursor, total) : 0ull;
    __syncthreads();
    gp += s_gbase;
    __syncthreads();  // (s_gbase is rewritten by the next round)
#pragma unroll
    for (int u = 0; u < PR_U; ++u) {
      if (m[u] == 1) {
        if ((int64_t)gp < capacity) {
          out_probe[gp] = rec[u].row;
          out_build[gp] = first[u];
        }
        ++gp;
      } else if (m[u] > 1) {
        const K key = (K)(((uint64_t)rec[u].khi << 32) | rec[u].klo);
        walk4(key, [&](int32_t br) {
          if ((int64_t)gp < capacity) {
            out_probe[gp] = rec[u].row;
            out_build[gp] = br;
          }
          ++gp;
        });
      }
    }
  }
}

// ================================================================================================
// Round 5: the L2-RESIDENT direct probe (track B) -- selectable, gx_join_set_probe_kernel(2 | 3).
//
// k_pj2_probe_pipe keeps a sub-table's 4-bit tags in LDS so that only a tag match costs a slot read; that is ~128 VALU
// instructions per row (SWAR scans of a 16-slot window, three pipeline stages of 64-bit keys) at ONE 16-wave gang per CU
// (64 KiB of tags + 92 KiB of staging fill the LDS): half instruction issue, half exposed latency, 1.9 TB/s.  This kernel
// asks the opposite question: the partition pass hands the pieces of XCD y's partitions to the workgroups of XCD y IN ORDER
// (per-XCD ticket lists), so at any moment the 32 CUs of an XCD probe the same one or two 2-MiB sub-tables -- which the
// XCD's 4-MiB L2 holds.  So: no tags, no LDS tables, no staging; every row reads its home slot straight from the L2 and walks
// its chain there (1.3 - 1.8 slots at load 0.37, nearly always inside one 128-B line), ~45 VALU per row, and with the LDS
// free the CU holds 24 - 32 waves instead of 16 to cover the ~200-cycle L2 latency.  Output positions: the single matches
// of a piece are numbered through one LDS counter, ONE returning atomic per piece reserves their run, and the pairs wait in
// registers for one iteration until that atomic has come back (the wave never waits for it); rows with duplicate build keys
// (rare) reserve and write on their own.  Same PieceTable, same scatter, same fallback gating as k_pj2_probe_pipe.
// ================================================================================================
constexpr int PD_BT = 512;
template <typename K, int R, int WPE>
__global__ void __launch_bounds__(PD_BT, WPE)
k_pj3_probe_direct(const K* __restrict__ pkeys, const int32_t* __restrict__ pidx, PieceTable pt, int pbits,
                   const Slot<K>* __restrict__ slots, uint32_t log2cap, int left_outer, int32_t* __restrict__ out_probe,
                   int32_t* __restrict__ out_build, int64_t capacity, unsigned long long* cursor)
{
  typedef typename SlotRaw<K>::type Raw;
  __shared__ PpPiece s_piece[2];
  __shared__ unsigned int s_cnt[2];
  __shared__ unsigned long long s_base[2];
  const unsigned tid  = threadIdx.x;
  const unsigned lane = lane_id();
  const unsigned w    = tid / GX_WAVE;
  const int P         = 1 << pbits;
  const int LISTP     = P / PJ_NR;
  const uint32_t mask = log2cap >= 32 ? 0xFFFFFFFFu : ((1u << log2cap) - 1u);  // (rows are int32: a table has at most 2^32 slots)
  // ---- wave 0 resolves pieces (the service wave's search of k_pj2_probe_pipe): ticket of the XCD's list, partition, region
  const unsigned x0 = pj_xcc();
  unsigned ylist    = 0;
  auto take_piece = [&](PpPiece& pc) {
    pc.valid = 0;
    pc.c0 = pc.c1 = 0;
    pc.part = 0;
    unsigned int g = 0xFFFFFFFFu, y = 0;
    if (lane == 0) {
      while (ylist < PJ_NR) {
        y                      = (x0 + ylist) % PJ_NR;
        const unsigned int nch = pt.list_chunk0[y + 1] - pt.list_chunk0[y];
        if (nch) {
          const unsigned int t = atomicAdd(&pt.ticket[y].v, 1u);
          if (t < nch) {
            g = pt.list_chunk0[y] + t;
            break;
          }
        }
        ++ylist;
      }
    }
    g     = (unsigned int)__builtin_amdgcn_readfirstlane((int)g);
    y     = (unsigned int)__builtin_amdgcn_readfirstlane((int)y);
    ylist = (unsigned int)__builtin_amdgcn_readfirstlane((int)ylist);
    if (g == 0xFFFFFFFFu) return;
    unsigned int part = 0;
    bool hit = false;
    for (int e = (int)lane; e < LISTP; e += GX_WAVE) {
      const int p           = (int)y * LISTP + e;
      const unsigned int lo = pt.chunk0[p * pt.nr], hi = pt.chunk0[(p + 1) * pt.nr];
      if (lo <= g && g < hi) {
        part = (unsigned int)p;
        hit  = true;
      }
    }
    uint64_t hb = ballot(hit);
    part        = shfl(part, __builtin_ctzll(hb));
    unsigned int reg = part * (unsigned int)pt.nr, loc = 0;
    hit = false;
    if ((int)lane < pt.nr) {
      const unsigned int e  = part * (unsigned int)pt.nr + lane;
      const unsigned int lo = pt.chunk0[e], hi = pt.chunk0[e + 1];
      if (lo <= g && g < hi) {
        reg = e;
        loc = g - lo;
        hit = true;
      }
    }
    hb           = ballot(hit);
    const int sr = __builtin_ctzll(hb);
    reg          = shfl(reg, sr);
    loc          = shfl(loc, sr);
    unsigned long long r0, r1;
    if (pt.start) {
      r0 = pt.start[reg];
      r1 = pt.start[reg + 1];
    } else {
      unsigned int c = pt.fill[reg];
      c              = c < pt.cap ? c : pt.cap;
      r0             = (unsigned long long)reg * pt.cap;
      r1             = r0 + c;
    }
    constexpr unsigned long long ROWS = (unsigned long long)PD_BT * R;
    pc.c0    = r0 + (unsigned long long)loc * ROWS;
    pc.c1    = pc.c0 + ROWS < r1 ? pc.c0 + ROWS : r1;
    pc.part  = part;
    pc.valid = 1;
  };
  if (tid < 2) s_cnt[tid] = 0;
  if (w == 0) {
    PpPiece pc;
    take_piece(pc);
    if (lane == 0) s_piece[0] = pc;
  }
  __syncthreads();
  // the single matches of the PREVIOUS piece: written one iteration late, when the piece's reservation has come back
  int32_t h_idx[R], h_first[R];
  uint32_t h_pos[R];
  uint32_t h_live = 0;
#pragma unroll
  for (int j = 0; j < R; ++j) {
    h_idx[j] = h_first[j] = 0;
    h_pos[j] = 0;
  }
  unsigned long long pend = 0;  // thread 0: the returning atomic of the previous piece
  for (int t = 0;; ++t) {
    const PpPiece pc = s_piece[t & 1];
    const bool valid = pc.valid != 0;  // block-uniform
    K key[R];
    int32_t idx[R];
    uint32_t act = 0;
    if (valid) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const unsigned long long i  = pc.c0 + (unsigned long long)(j * PD_BT) + tid;
        const unsigned long long ic = i < pc.c1 ? i : pc.c0;
        key[j] = __builtin_nontemporal_load(&pkeys[ic]);
        idx[j] = __builtin_nontemporal_load(&pidx[ic]);
        if (i < pc.c1) act |= 1u << j;
      }
    }
    if (tid == 0) s_base[(t & 1) ^ 1] = pend;  // (read behind this iteration's barrier)
    if (w == 0 && valid) {                     // the next piece resolves under this piece's loads
      PpPiece nx;
      take_piece(nx);
      if (lane == 0) s_piece[(t + 1) & 1] = nx;
    }
    uint32_t m[R], pos[R];
    int32_t first[R];
    uint32_t live = 0;
    if (valid) {
      uint32_t sl[R];
      Raw sv[R];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        m[j]     = 0;
        first[j] = NO_MATCH;
        sl[j]    = (uint32_t)slot_of<K>(key[j], log2cap);
        sv[j]    = *reinterpret_cast<const Raw*>(&slots[sl[j]]);  // the home slot of every row, R loads in flight per lane
      }
      const uint32_t rows = act;
      for (;;) {
        uint32_t next = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
          if ((act >> j) & 1u) {
            K k;
            int32_t r;
            unpack_slot(sv[j], k, r);
            if (r != EMPTY_ROW) {
              if (k == key[j]) {
                if (m[j] == 0) first[j] = r;
                ++m[j];
              }
              sl[j] = (sl[j] + 1u) & mask;
              next |= 1u << j;
            }
          }
        }
        act = next;
        if (act == 0) break;
#pragma unroll
        for (int j = 0; j < R; ++j)
          if ((act >> j) & 1u) sv[j] = *reinterpret_cast<const Raw*>(&slots[sl[j]]);
      }
      // duplicate build keys: such a row reserves its own run and walks its chain again (the lines are L2 / L1 resident)
#pragma unroll
      for (int j = 0; j < R; ++j) {
        if (left_outer && ((rows >> j) & 1u) && m[j] == 0) m[j] = 1;  // (row, JoinNoMatch): first[j] is NO_MATCH
        if (m[j] > 1) {
          unsigned long long gp = atomicAdd(cursor, (unsigned long long)m[j]);
          uint32_t hh           = (uint32_t)slot_of<K>(key[j], log2cap);
          for (;;) {
            K k;
            int32_t r;
            load_slot<K>(&slots[hh], k, r);
            if (r == EMPTY_ROW) break;
            if (k == key[j]) {
              if ((int64_t)gp < capacity) {
                out_probe[gp] = idx[j];
                out_build[gp] = r;
              }
              ++gp;
            }
            hh = (hh + 1u) & mask;
          }
          m[j] = 0;
        }
      }
      // single matches: numbered inside the piece through one LDS counter
      uint32_t tot = 0;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const uint64_t bb = ballot(m[j] == 1);
        pos[j]            = tot + (uint32_t)__builtin_popcountll(bb & lanemask_lt());
        tot += (uint32_t)__builtin_popcountll(bb);
        if (m[j] == 1) live |= 1u << j;
      }
      uint32_t woff = 0;
      if (lane == 0 && tot) woff = atomicAdd(&s_cnt[t & 1], tot);
      woff = (uint32_t)__builtin_amdgcn_readfirstlane((int)woff);
#pragma unroll
      for (int j = 0; j < R; ++j) pos[j] += woff;
    }
    __syncthreads();
    // ---- the previous piece's pairs leave (its base sits in s_base); this piece's reservation is requested
    {
      const unsigned long long gb = s_base[(t & 1) ^ 1];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        if ((h_live >> j) & 1u) {
          const unsigned long long gp = gb + h_pos[j];
          if ((int64_t)gp < capacity) {
            __builtin_nontemporal_store(h_idx[j], &out_probe[gp]);
            __builtin_nontemporal_store(h_first[j], &out_build[gp]);
          }
        }
      }
    }
    if (!valid) break;  // (block-uniform: the pipeline has drained)
    if (tid == 0) {
      const unsigned int c = s_cnt[t & 1];
      pend                 = c ? atomicAdd(cursor, (unsigned long long)c) : 0ull;
      s_cnt[t & 1]         = 0;  // next used two iterations from here, behind the next barrier
    }
    h_live = live;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      h_idx[j]   = idx[j];
      h_first[j] = first[j];
      h_pos[j]   = pos[j];
    }
  }
}

// ================================================================================================
// Round 5, second form: the tag probe WITHOUT the software pipeline and WITHOUT the staging buffers (knob 4).
//
// Fresh counters of k_pj2_probe_pipe (profiles/r5_run1_pmc_sq_join_k0.txt): 571 VALU + 274 SALU wave-instructions per 256 rows,
// the VALU pipes ~58 % busy, 61 % of the wave cycles waiting -- at FOUR waves per SIMD, because 64 KiB of tags + 92 KiB of pair
// staging fill the LDS with one 16-wave gang.  The direct probe (k_pj3, above) showed that pairs need no staging: the single
// matches of a piece are numbered through one LDS counter, one returning atomic reserves their run, and the pairs wait in
// registers for one iteration.  Here the tags stay (they settle 9 of 10 misses without a slot read and bound every row's work:
// the direct probe's waves ran 16 chain steps per trip for their slowest lane), the staging goes, a workgroup is 1024 threads
// with 2 rows each in <= 64 registers, and TWO workgroups share a CU: 8 waves per SIMD cover the latencies the pipeline was
// built to hide, and the two workgroups' barriers and tag reloads overlap.
// ================================================================================================
constexpr int PT_BT = 1024;
constexpr int PT_PW = PT_BT / GX_WAVE - 1;  // 15 probe waves + the service wave
// LDS_TAGS = false (knob 6 / 7): the tag windows are read from GLOBAL memory, i.e. from the XCD's L2 -- nothing is staged in LDS,
// so the partition pass may cut the table into FEWER, larger sub-tables (2^20 slots: 512 KiB of tags, which the 4-MiB L2 holds
// while the XCD's workgroups probe that partition): P = 256 instead of 2048 partitions means 64-row runs and an eighth of the
// fill-counter atomics in k_pj2_scatter (the sort's level 0 moves 16 B/row at 4.4 TB/s with exactly that shape; the 2048-way
// scatter reaches 3.1 - 3.3).  Slot reads then come from the Infinity Cache / HBM instead of the L2.
template <typename K, int R, int WPE, bool LDS_TAGS>
__global__ void __launch_bounds__(PT_BT, WPE)
k_pj4_probe_tags(const K* __restrict__ pkeys, const int32_t* __restrict__ pidx, PieceTable pt, int pbits,
                 const Slot<K>* __restrict__ slots, uint32_t log2cap, int left_outer, int32_t* __restrict__ out_probe,
                 int32_t* __restrict__ out_build, int64_t capacity, unsigned long long* cursor)
{
  typedef typename SlotRaw<K>::type Raw;
  constexpr uint32_t SUB = 1u << PJ_SUB_LOG2;
  constexpr int ROWS     = PT_PW * GX_WAVE * R;  // rows per piece
  const int sub_log2     = (int)log2cap - pbits;  // slots per sub-table (LDS_TAGS: PJ_SUB_LOG2)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t* s_tagw = reinterpret_cast<const uint32_t*>(smem);  // 64 KiB of tags + the 32 tags that follow them
  __shared__ PpPiece s_piece[4];            // piece of iteration (t & 3), resolved two iterations ahead by the service wave
  __shared__ unsigned int s_cnt[2];         // single matches of piece (t & 1)
  __shared__ unsigned long long s_base[2];  // output position of piece (t & 1)
  const unsigned tid  = threadIdx.x;
  const unsigned lane = lane_id();
  const unsigned w    = tid / GX_WAVE;
  const int P         = 1 << pbits;
  const int LISTP     = P / PJ_NR;
  const uint64_t mask = (1ull << log2cap) - 1;
  const uint8_t* gtags = reinterpret_cast<const uint8_t*>(slots + (mask + 1));
  if (tid < 2) s_cnt[tid] = 0;

  if (w == PT_PW) {
    // ------------------------------------------------------------------ service wave: tickets, region search, reservations.
    // Its global round trips (ticket atomic -> region table -> fill counter; the reservation atomic) run two iterations ahead of /
    // one behind the probe waves and never sit between their barriers.
    const unsigned x0 = pj_xcc();
    unsigned ylist    = 0;
    auto take_piece = [&](PpPiece& pc) {
      pc.valid = 0;
      pc.c0 = pc.c1 = 0;
      pc.part = 0;
      unsigned int g = 0xFFFFFFFFu, y = 0;
      if (lane == 0) {
        while (ylist < PJ_NR) {
          y                      = (x0 + ylist) % PJ_NR;
          const unsigned int nch = pt.list_chunk0[y + 1] - pt.list_chunk0[y];
          if (nch) {
            const unsigned int t = atomicAdd(&pt.ticket[y].v, 1u);
            if (t < nch) {
              g = pt.list_chunk0[y] + t;
              break;
            }
          }
          ++ylist;
        }
      }
      g     = (unsigned int)__builtin_amdgcn_readfirstlane((int)g);
      y     = (unsigned int)__builtin_amdgcn_readfirstlane((int)y);
      ylist = (unsigned int)__builtin_amdgcn_readfirstlane((int)ylist);
      if (g == 0xFFFFFFFFu) return;
      unsigned int part = 0;
      bool hit = false;
      for (int e = (int)lane; e < LISTP; e += GX_WAVE) {
        const int p           = (int)y * LISTP + e;
        const unsigned int lo = pt.chunk0[p * pt.nr], hi = pt.chunk0[(p + 1) * pt.nr];
        if (lo <= g && g < hi) {
          part = (unsigned int)p;
          hit  = true;
        }
      }
      uint64_t hb = ballot(hit);
      part        = shfl(part, __builtin_ctzll(hb));
      unsigned int reg = part * (unsigned int)pt.nr, loc = 0;
      hit = false;
      if ((int)lane < pt.nr) {
        const unsigned int e  = part * (unsigned int)pt.nr + lane;
        const unsigned int lo = pt.chunk0[e], hi = pt.chunk0[e + 1];
        if (lo <= g && g < hi) {
          reg = e;
          loc = g - lo;
          hit = true;
        }
      }
      hb           = ballot(hit);
      const int sr = __builtin_ctzll(hb);
      reg          = shfl(reg, sr);
      loc          = shfl(loc, sr);
      unsigned long long r0, r1;
      if (pt.start) {
        r0 = pt.start[reg];
        r1 = pt.start[reg + 1];
      } else {
        unsigned int c = pt.fill[reg];
        c              = c < pt.cap ? c : pt.cap;
        r0             = (unsigned long long)reg * pt.cap;
        r1             = r0 + c;
      }
      pc.c0    = r0 + (unsigned long long)loc * ROWS;
      pc.c1    = pc.c0 + ROWS < r1 ? pc.c0 + ROWS : r1;
      pc.part  = part;
      pc.valid = 1;
    };
    PpPiece pc;
    take_piece(pc);
    if (lane == 0) s_piece[0] = pc;
    take_piece(pc);
    if (lane == 0) s_piece[1] = pc;
    __syncthreads();  // prologue barrier
    unsigned long long pend = 0;
    uint32_t cur_part       = 0xFFFFFFFFu;
    for (int t = 0;; ++t) {
      const PpPiece cur = s_piece[t & 3];
      const bool valid  = cur.valid != 0;
      if (lane == 0) s_base[(t & 1) ^ 1] = pend;  // the reservation of piece t - 1 (asked for behind last iteration's barrier)
      if (valid) {
        take_piece(pc);
        if (lane == 0) s_piece[(t + 2) & 3] = pc;
      }
      if (LDS_TAGS && valid && cur.part != cur_part) {  // the probe waves reload their tags: same (block-uniform) test, same barrier
        cur_part = cur.part;
        __syncthreads();
      }
      __syncthreads();  // X(t): the counts of piece t are in
      if (!valid) break;
      if (lane == 0) {
        const unsigned int c = s_cnt[t & 1];
        pend                 = c ? atomicAdd(cursor, (unsigned long long)c) : 0ull;
        s_cnt[t & 1]         = 0;  // next used two iterations from here, behind the next barrier
      }
    }
    return;
  }

  // ---------------------------------------------------------------------- probe waves
  __syncthreads();  // prologue barrier: pieces 0 and 1 are resolved
  int32_t h_idx[R], h_first[R];  // the single matches of the PREVIOUS piece: written when its reservation has come back
  uint32_t h_live = 0, h_woff = 0;
  K keyN[R];                     // rows of the NEXT piece, requested one iteration ahead
  int32_t idxN[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    h_idx[j] = h_first[j] = 0;
    keyN[j] = K(0);
    idxN[j] = 0;
  }
  auto request_rows = [&](const PpPiece& pc) {
    if (!pc.valid) return;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const unsigned long long i  = pc.c0 + (unsigned long long)(j * PT_PW * GX_WAVE) + tid;
      const unsigned long long ic = i < pc.c1 ? i : pc.c0;
      keyN[j] = __builtin_nontemporal_load(&pkeys[ic]);
      idxN[j] = __builtin_nontemporal_load(&pidx[ic]);
    }
  };
  request_rows(s_piece[0]);
  uint32_t cur_part = 0xFFFFFFFFu;  // the partition whose tags sit in LDS
  for (int t = 0;; ++t) {
    const PpPiece pc = s_piece[t & 3];
    const bool valid = pc.valid != 0;  // block-uniform
    K key[R];
    int32_t idx[R];
    uint32_t rows = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      key[j] = keyN[j];
      idx[j] = idxN[j];
      if (valid && pc.c0 + (unsigned long long)(j * PT_PW * GX_WAVE) + tid < pc.c1) rows |= 1u << j;
    }
    if (valid) request_rows(s_piece[(t + 1) & 3]);  // (resolved by the service wave before the last barrier)
    if (LDS_TAGS && valid && pc.part != cur_part) {  // block-uniform: the tags of the piece's partition (every earlier reader passed the last barrier)
      const uint4* src = reinterpret_cast<const uint4*>(gtags + (((uint64_t)pc.part << PJ_SUB_LOG2) >> 1));
      uint4* dst       = reinterpret_cast<uint4*>(smem);
      for (uint32_t i = tid; i < SUB / 2 / 16; i += PT_PW * GX_WAVE) dst[i] = src[i];
      if (tid == 0) {  // the 32 tags behind the sub-table's own; behind the LAST sub-table the table wraps to slot 0
        const bool last   = (((uint64_t)pc.part + 1) << PJ_SUB_LOG2) > mask;
        dst[SUB / 2 / 16] = last ? *reinterpret_cast<const uint4*>(gtags) : src[SUB / 2 / 16];
      }
      cur_part = pc.part;
      __syncthreads();
    }
    uint32_t m[R];
    int32_t first[R];
    uint32_t live = 0, woff = 0;
    if (valid) {
      const uint64_t sub_base = (uint64_t)pc.part << sub_log2;
      const uint32_t* gtagw0  = reinterpret_cast<const uint32_t*>(gtags);
      uint32_t li[R];
      uint64_t cand[R];
      uint32_t ended = 0, edge = 0;
      Raw sv[R];
      uint32_t tw[LDS_TAGS ? 1 : R][3];
      if (!LDS_TAGS) {  // the 16-slot tag windows of all R rows are requested together (three dwords each, from the L2)
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const uint64_t sl = ((uint64_t)key[j] * 0x9E3779B97F4A7C15ull) >> (64 - log2cap);
          // (a window that would run past the table's last tag word is read from the table's start instead and its row takes the
          //  slot walk below: 16 of 2^log2cap home slots)
          const uint64_t wi = (sl >> 3) + 2 <= (mask >> 3) ? (sl >> 3) : 0;
          tw[j][0] = gtagw0[wi];
          tw[j][1] = gtagw0[wi + 1];
          tw[j][2] = gtagw0[wi + 2];
        }
      }
#pragma unroll
      for (int j = 0; j < R; ++j) {
        m[j]     = 0;
        first[j] = NO_MATCH;
        cand[j]  = 0;
        li[j]    = 0;
        if ((rows >> j) & 1u) {
          const uint64_t prod = (uint64_t)key[j] * 0x9E3779B97F4A7C15ull;
          li[j]               = (uint32_t)((prod >> (64 - log2cap)) - sub_base);
          uint32_t tg         = (uint32_t)(prod >> (60 - log2cap)) & 15u;
          tg                  = tg ? tg : 8u;
          const uint32_t tagpat = tg * 0x11111111u;
          uint32_t w0, w1, w2;
          if (LDS_TAGS) {
            w0 = s_tagw[li[j] >> 3];
            w1 = s_tagw[(li[j] >> 3) + 1];
            w2 = s_tagw[(li[j] >> 3) + 2];
          } else {
            w0 = tw[LDS_TAGS ? 0 : j][0];
            w1 = tw[LDS_TAGS ? 0 : j][1];
            w2 = tw[LDS_TAGS ? 0 : j][2];
          }
          const uint32_t sh = (li[j] & 7u) * 4u;
          const uint32_t x0w = __builtin_amdgcn_alignbit(w1, w0, sh), x1w = __builtin_amdgcn_alignbit(w2, w1, sh);
          const uint32_t y0 = x0w ^ tagpat, y1 = x1w ^ tagpat;
          const uint32_t z0 = ~(((x0w & 0x77777777u) + 0x77777777u) | x0w) & 0x88888888u;  // empty slots
          const uint32_t z1 = ~(((x1w & 0x77777777u) + 0x77777777u) | x1w) & 0x88888888u;
          const uint32_t m0 = ~(((y0 & 0x77777777u) + 0x77777777u) | y0) & 0x88888888u;    // tag matches
          const uint32_t m1 = ~(((y1 & 0x77777777u) + 0x77777777u) | y1) & 0x88888888u;
          const uint32_t c0 = m0 & ((z0 & (0u - z0)) - 1u);                                 // ... below the first empty one
          const uint32_t c1 = z0 ? 0u : (m1 & ((z1 & (0u - z1)) - 1u));
          cand[j]           = (uint64_t)c0 | ((uint64_t)c1 << 32);
          if (z0 | z1) ended |= 1u << j;
          if (!LDS_TAGS && ((sub_base + li[j]) >> 3) + 2 > (mask >> 3)) {  // the window was not the row's: no candidates, chain open
            cand[j] = 0;
            ended &= ~(1u << j);
            edge |= 1u << j;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < R; ++j)  // the first candidate slot of every row that has one: R loads in flight per lane
        if (cand[j]) sv[j] = *reinterpret_cast<const Raw*>(&slots[(sub_base + li[j] + ((uint32_t)__builtin_ctzll(cand[j]) >> 2)) & mask]);
#pragma unroll
      for (int j = 0; j < R; ++j) {
        if (!((rows >> j) & 1u)) continue;
        if (cand[j]) {
          K sk;
          int32_t sr;
          unpack_slot(sv[j], sk, sr);
          if (sk == key[j]) {  // a tagged slot is never empty
            m[j]     = 1;
            first[j] = sr;
          }
          cand[j] &= cand[j] - 1;
        }
        // rare: more tag candidates in the window (1 in ~25 rows), or a chain that runs past 16 slots (1e-5): finished in place
        uint64_t cnd64 = cand[j];
        uint64_t wb    = sub_base + li[j];
        while (cnd64) {
          K k;
          int32_t r;
          load_slot<K>(&slots[(wb + ((uint32_t)__builtin_ctzll(cnd64) >> 2)) & mask], k, r);
          if (k == key[j]) {
            if (m[j] == 0) first[j] = r;
            ++m[j];
          }
          cnd64 &= cnd64 - 1;
        }
        if (!((ended >> j) & 1u)) {
          const uint32_t tagpat = tag_of<K>(key[j], log2cap) * 0x11111111u;
          const uint32_t* gtagw = reinterpret_cast<const uint32_t*>(gtags);
          wb                    = ((edge >> j) & 1u) ? wb : ((wb + 16) & mask);  // (edge rows: nothing of the chain has been looked at)
          bool done             = false;
          while (!done) {  // further windows of 8 slots, tags from global memory
            if (wb + 8 > mask) {  // the window would wrap around the table's end: walk the slots themselves
              for (;;) {
                K k;
                int32_t r;
                load_slot<K>(&slots[wb & mask], k, r);
                if (r == EMPTY_ROW) break;
                if (k == key[j]) {
                  if (m[j] == 0) first[j] = r;
                  ++m[j];
                }
                ++wb;
              }
              break;
            }
            uint32_t cnd = 0;
            done         = scan_tags8_global(gtagw, (uint32_t)wb, tagpat, cnd);
            while (cnd) {
              K k;
              int32_t r;
              load_slot<K>(&slots[wb + ((uint32_t)__builtin_ctz(cnd) >> 2)], k, r);
              if (k == key[j]) {
                if (m[j] == 0) first[j] = r;
                ++m[j];
              }
              cnd &= cnd - 1;
            }
            wb += 8;
          }
        }
      }
      // duplicate build keys: such a row reserves its own run and walks its chain again
#pragma unroll
      for (int j = 0; j < R; ++j) {
        if (left_outer && ((rows >> j) & 1u) && m[j] == 0) m[j] = 1;  // (row, JoinNoMatch): first[j] is NO_MATCH
        if (m[j] > 1) {
          unsigned long long gp = atomicAdd(cursor, (unsigned long long)m[j]);
          uint64_t hh           = slot_of<K>(key[j], log2cap);
          for (;;) {
            K k;
            int32_t r;
            load_slot<K>(&slots[hh], k, r);
            if (r == EMPTY_ROW) break;
            if (k == key[j]) {
              if ((int64_t)gp < capacity) {
                out_probe[gp] = idx[j];
                out_build[gp] = r;
              }
              ++gp;
            }
            hh = (hh + 1) & mask;
          }
          m[j] = 0;
        }
      }
      // the wave's single matches: one LDS add hands out its run inside the piece
      uint32_t tot = 0;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        tot += (uint32_t)__builtin_popcountll(ballot(m[j] == 1));
        if (m[j] == 1) live |= 1u << j;
      }
      if (lane == 0 && tot) woff = atomicAdd(&s_cnt[t & 1], tot);
      woff = (uint32_t)__builtin_amdgcn_readfirstlane((int)woff);
    }
    __syncthreads();  // X(t)
    {
      // the previous piece's pairs leave: position = the piece's reservation + the wave's run + the lane's rank among the matches
      const unsigned long long gb = s_base[(t & 1) ^ 1] + h_woff;
      uint32_t run = 0;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const uint64_t bb = ballot((h_live >> j) & 1u);
        if ((h_live >> j) & 1u) {
          const unsigned long long gp = gb + run + (uint32_t)__builtin_popcountll(bb & lanemask_lt());
          if ((int64_t)gp < capacity) {
            __builtin_nontemporal_store(h_idx[j], &out_probe[gp]);
            __builtin_nontemporal_store(h_first[j], &out_build[gp]);
          }
        }
        run += (uint32_t)__builtin_popcountll(bb);
      }
    }
    if (!valid) break;
    h_live = live;
    h_woff = woff;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      h_idx[j]   = idx[j];
      h_first[j] = first[j];
    }
  }
}

// optional per-kernel timing of the partitioned probe with HIP events on the caller's stream (bench.py)
struct JoinProfile {
  bool enabled = false, created = false, marked = false;
  hipEvent_t ev[4];  // before hist | before scatter | before probe | after probe
};
constexpr int JPROF_SLOTS = 64;  // event slots, as gx_sort_profile_slot: K timed calls, K slots, read after the last one
static thread_local JoinProfile g_jprofs[JPROF_SLOTS];
static thread_local int g_jprof_slot = 0;
#define g_jprof g_jprofs[g_jprof_slot]
static inline void jprof_mark(int i, hipStream_t s)
{
  if (g_jprof.enabled) (void)hipEventRecord(g_jprof.ev[i], s);
}
static thread_local int g_pj_probe = 0; // probe kernel: 0 = default (software-pipelined tag probe), 1 = round-1 tag probe (A/B knob)
static thread_local int g_pj_build = 0; // partitioned build: 0 = window build, table composed in LDS and written once (round 4b, default), 2 = sub-table build with
                                        // the tags in LDS (round 4a), 1 = round-2 kernel (global CAS + k_tags) (A/B knob)
// round 6 (gx_join_set_experiment; default 133 = bits 0, 2, 7): bit 0 = the partition pass writes 12-byte {key, row} records (k_pj2_scatter_rec) and the probe
// reads them (8-byte keys); bit 1 = with bit 0, 24576-row scatter tiles; bit 2 = pipelined service wave of k_pj2_probe_pipe on fixed
// pieces per region; bit 3 = without bit 0, the windowed scatter writing key / row arrays (A/B); bits 4-6 = ablations (wrong results)
static thread_local int g_pj_xp = 133;
static thread_local int g_pj_ovf_cap = 0;  // tests: rows per overflow slice of k_pj2_probe_pipe<LONG> (0 = n / 16 / 256)
static thread_local int g_pj_tile = 0;  // scatter tile rows: 0 = default, else 4096 / 8192 / 16384 (A/B knob)
static thread_local int g_pj_probe_early = 0;  // round-3 probe: 1 = rows of a piece requested at the top of the trip, 0 = at its end (default: with the
                                  // deferral queue the early form no longer fits 128 VGPRs) (A/B knob)
static thread_local int g_pj_defer = 0;        // round-3 probe: 1 = unsettled rows are parked in a per-wave queue for one trip, 0 = settled in place (default:
                                  // measured 0.3 ms SLOWER with the queue once the tag window covers 16 slots -- profiles/r3_run4_*) (A/B knob)

// the partition pass shared by the partitioned probe and build (F = TableTop) and by gx_partition_rows
template <typename K, typename F>
int pj_partition_fn(const K* keys, int64_t n, int pbits, PjPlan* plan, K* pkeys, int32_t* pidx, unsigned int chunk_rows, hipStream_t s,
                    F part_of, bool profile = false, int32_t row0 = 0, uint32_t spec_cap = 0, const int32_t* payload = nullptr)
{
  GX_HIP_TRY(hipMemsetAsync(plan, 0, sizeof(PjPlan), s));
  // tile: as many rows as the LDS holds next to the three P-entry arrays (160 KiB per CU)
  int tile_rows = g_pj_tile ? g_pj_tile : 16384;
  while (tile_rows > 4096 && (size_t)tile_rows * sizeof(K) + ((size_t)12 << pbits) + 256 > (size_t)160 * 1024) tile_rows /= 2;
  // few partitions (the rank split of a sharded sort / join, small join tables): runs are long whatever the tile, so
  // take the 8192-row tile whose 64 KiB let two workgroups share a CU and overlap each other's load and write phases
  if (!g_pj_tile && pbits <= 6 && tile_rows > 8192) tile_rows = 8192;
  if (n <= 4 * (int64_t)tile_rows * 256) tile_rows = 4096;  // small inputs: more, smaller workgroups
  const int64_t rrows = pj_range_rows(n, tile_rows);
  if (profile) jprof_mark(0, s);
  int64_t hb = div_up(n, 256 * 8 * 4 * PJ_NR);
  if (hb > 256) hb = 256;
  if (hb < 1) hb = 1;
  if (!spec_cap) {  // (the speculative form has no histogram: fixed slots, fill counters from zero)
    hipLaunchKernelGGL((k_pj_hist<K, F>), dim3((unsigned)(hb * PJ_NR)), dim3(256), 0, s, keys, n, plan, pbits, rrows, part_of, (const unsigned int*)nullptr);
    hipLaunchKernelGGL(k_pj_offsets, dim3(1), dim3(1024), 0, s, plan, pbits, chunk_rows, (const unsigned int*)nullptr);
  }
  if (profile) jprof_mark(1, s);
  const size_t lds = (size_t)tile_rows * sizeof(K) + ((size_t)12 << pbits);
  auto k4          = k_pj_scatter<K, 8, 512, F>;
  auto k8          = k_pj_scatter<K, 16, 512, F>;
  auto k16         = k_pj_scatter<K, 16, 1024, F>;
  static std::atomic<bool> attr_set{false};  // per instantiation
  if (!attr_set) {
    const int lds_max = 160 * 1024 - 256;
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k8), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k16), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    attr_set = true;
  }
  const unsigned grid = (unsigned)div_up(n, (int64_t)tile_rows);
  if (tile_rows == 16384) hipLaunchKernelGGL(k16, dim3(grid), dim3(1024), lds, s, keys, n, plan, pbits, rrows, pkeys, pidx, part_of, row0, spec_cap, payload);
  else if (tile_rows == 8192) hipLaunchKernelGGL(k8, dim3(grid), dim3(512), lds, s, keys, n, plan, pbits, rrows, pkeys, pidx, part_of, row0, spec_cap, payload);
  else hipLaunchKernelGGL(k4, dim3(grid), dim3(512), lds, s, keys, n, plan, pbits, rrows, pkeys, pidx, part_of, row0, spec_cap, payload);
  if (profile) jprof_mark(2, s);
  GX_LAUNCH_CHECK();
  return 0;
}
template <typename K>
int pj_partition(const K* keys, int64_t n, int pbits, PjPlan* plan, K* pkeys, int32_t* pidx, unsigned int chunk_rows, hipStream_t s,
                 bool profile = false, int32_t row0 = 0, const int32_t* payload = nullptr)
{
  return pj_partition_fn<K, TableTop<K>>(keys, n, pbits, plan, pkeys, pidx, chunk_rows, s, TableTop<K>{pbits}, profile, row0, 0u, payload);
}

// partition starts as int64, for the caller of gx_partition_rows
__global__ void k_pj_export_offsets(const PjPlan* plan, int nparts, long long* out)
{
  for (int i = threadIdx.x; i <= nparts; i += blockDim.x) out[i] = (long long)plan->offset[i];
}
// speculative form: rows per group (clamped to the slot), then the overflow flag
__global__ void k_pj_export_fills(const PjPlan* plan, int nparts, uint32_t cap, long long* out)
{
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
    const unsigned long long f = plan->cursor[0][i];
    out[i]                     = (long long)(f < cap ? f : cap);
  }
  if (threadIdx.x == 0) out[nparts] = (long long)plan->spec_overflow;
}

template <typename K>
int partition_rows_hash(const void* keys, int64_t n, int pbits, int nparts, void* out_keys, int32_t* out_rows, int64_t* offsets,
                        void* tmp, size_t* tmp_bytes, hipStream_t s, int32_t row0, uint32_t spec_cap = 0)
{
  Carver c(tmp);
  PjPlan* plan = c.take<PjPlan>(1);
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  int rc = pj_partition_fn<K, AltHash<K>>(static_cast<const K*>(keys), n, pbits < 3 ? 3 : pbits, plan, static_cast<K*>(out_keys), out_rows,
                                          PJ_CHUNK, s, AltHash<K>{(unsigned int)nparts}, false, row0, spec_cap);
  if (rc) return rc;
  if (spec_cap) hipLaunchKernelGGL(k_pj_export_fills, dim3(1), dim3(256), 0, s, plan, nparts, spec_cap, reinterpret_cast<long long*>(offsets));
  else hipLaunchKernelGGL(k_pj_export_offsets, dim3(1), dim3(256), 0, s, plan, nparts, reinterpret_cast<long long*>(offsets));
  GX_LAUNCH_CHECK();
  return 0;
}
template <typename K, int KIND>
int partition_rows_range(const void* keys, int64_t n, int nparts, const void* splitters_host, void* out_keys, int32_t* out_rows,
                         int64_t* offsets, void* tmp, size_t* tmp_bytes, hipStream_t s, int32_t row0, uint32_t spec_cap = 0)
{
  Carver c(tmp);
  PjPlan* plan = c.take<PjPlan>(1);
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  RangeSplit<K, KIND> f;
  f.nsplit = nparts - 1;
  for (int i = 0; i < PJ_MAX_SPLIT; ++i)
    f.split[i] = i < f.nsplit ? to_sortable<K, KIND>(static_cast<const K*>(splitters_host)[i], K(0)) : K(0);
  int pbits = 3;
  while ((1 << pbits) < nparts) ++pbits;
  int rc = pj_partition_fn<K, RangeSplit<K, KIND>>(static_cast<const K*>(keys), n, pbits, plan, static_cast<K*>(out_keys), out_rows, PJ_CHUNK,
                                                   s, f, false, row0, spec_cap);
  if (rc) return rc;
  if (spec_cap) hipLaunchKernelGGL(k_pj_export_fills, dim3(1), dim3(256), 0, s, plan, nparts, spec_cap, reinterpret_cast<long long*>(offsets));
  else hipLaunchKernelGGL(k_pj_export_offsets, dim3(1), dim3(256), 0, s, plan, nparts, reinterpret_cast<long long*>(offsets));
  GX_LAUNCH_CHECK();
  return 0;
}

static inline int pj_bits(uint32_t log2cap, int slot_bytes)
{
  // sub-table of about 2 MiB: P = table bytes / 2 MiB
  (void)slot_bytes;
  int pb = (int)log2cap - PJ_SUB_LOG2;  // sub-table = 2^17 slots: 2 MiB of 16-B slots, 64 KiB of tags
  if (pb < 3) pb = 0;  // small tables: the direct probe is already cache resident
  if (pb > 12) pb = 12;
  return pb;
}

// The round-3 probe: speculative hist-free partition (padded regions, persistent scatter) -> probe over the region table,
// with the exact sequence (histogram, offsets, exact scatter, probe) enqueued behind it and gated on `fallback`.
static thread_local int g_pj_spec = 1;  // A/B knob: 1 = this path where it applies (default), 0 = the round-2 path
template <typename K>
static bool pj2_applies(int64_t n, int pbits)
{
  const size_t lds = (size_t)16384 * sizeof(K) + ((size_t)8 << pbits) + 512;
  return g_pj_spec && g_pj_probe != 1 && pbits >= 3 && pbits <= 12 && lds <= (size_t)160 * 1024 && n > 0 &&
         (g_pj_spec == 2 || n > 4 * (int64_t)16384 * 256);  // 2: forced for any n (tests)
}
template <typename K>
int probe_partitioned_impl2(const K* keys, int64_t n, const Slot<K>* slots, uint32_t lg, int pbits, int left_outer, int32_t* out_probe,
                            int32_t* out_build, int64_t capacity, int64_t* cursor, void* tmp, size_t* tmp_bytes, hipStream_t s, int32_t row0,
                            const int32_t* payload)
{
  // round 6 (gx_join_set_experiment bit 0; 8-byte keys): 12-byte record runs (k_pj2_scatter_rec) + the probe's record loads; bit 1: 24576-row tiles
  const bool rec       = (g_pj_xp & 1) != 0 && sizeof(K) == 8 && !g_pj_defer && !g_pj_probe_early && g_pj_probe == 0;
  const bool rec24     = rec && (g_pj_xp & 2) != 0;
  const bool win_soa   = !rec && (g_pj_xp & 8) != 0 && sizeof(K) == 8;  // bit 3: the windowed scatter writing key and row ARRAYS (any probe kernel)
  const int TILE       = rec24 ? 24576 : 16384;
  const uint32_t cap   = pj2_cap(n, pbits);
  const size_t nslots  = ((size_t)PJ_NR << pbits) * cap;
  const size_t nbuf    = nslots > (size_t)n ? nslots : (size_t)n;
  Carver c(tmp);
  Pj2Plan* plan2 = c.take<Pj2Plan>(1);
  PjPlan* plan   = c.take<PjPlan>(1);
  K* pkeys       = c.take<K>(nbuf);          // (record form: the 12-byte records occupy pkeys and pidx, which are carved back to back
  int32_t* pidx  = c.take<int32_t>(nbuf);    //  -- 256-byte carving leaves pidx at or behind pkeys + 8 nbuf, so 12 nbuf bytes fit)
  // bit 7 (k_pj2_probe_pipe<LONG>): the overflow list -- one slice per probe workgroup, n / 16 rows in all (rows that need a dependent
  // read: 1e-3 of them with keys that hash like random numbers; a full slice sends its workgroup back to walking in place)
  const bool longp        = rec && (g_pj_xp & 128) != 0 && ((g_pj_xp >> 4) & 7) == 0 && lg <= 28 && sizeof(K) == 8;
  constexpr int OVF_WGS   = 1024;  // >= any probe grid
  int64_t ovf_cap64       = n / 16 / 256;
  if (ovf_cap64 < 64) ovf_cap64 = 64;
  if (g_pj_ovf_cap > 0) ovf_cap64 = g_pj_ovf_cap;  // (tests: tiny slices)
  const unsigned int ovf_cap = (unsigned int)ovf_cap64;
  PjRec* ovf              = longp ? c.take<PjRec>((size_t)ovf_cap * 256) : nullptr;
  unsigned int* ovf_count = longp ? c.take<unsigned int>(OVF_WGS) : nullptr;
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  typedef TableTop<K> F;
  const F part_of{pbits};
  auto kspec  = k_pj2_scatter<K, 16, 1024, false, F>;
  auto kexact = k_pj2_scatter<K, 16, 1024, true, F>;
  auto kprobe = g_pj_defer ? (g_pj_probe_early ? k_pj2_probe_pipe<K, true, true> : k_pj2_probe_pipe<K, false, true>)
                           : (g_pj_probe_early ? k_pj2_probe_pipe<K, true, false> : k_pj2_probe_pipe<K, false, false>);
  if constexpr (sizeof(K) == 8) {
    if (rec) {
      const int abl = (g_pj_xp >> 4) & 7;  // measurement only: the ablated kernels give WRONG results
      kprobe        = abl == 1 ? k_pj2_probe_pipe<K, false, false, true, 1> : abl == 2 ? k_pj2_probe_pipe<K, false, false, true, 2>
                      : abl == 3 ? k_pj2_probe_pipe<K, false, false, true, 3>
                      : k_pj2_probe_pipe<K, false, false, true>;
      if ((g_pj_xp & 128) != 0 && abl == 0 && lg <= 28) kprobe = k_pj2_probe_pipe<K, false, false, true, 0, true>;  // bit 7: every load a trip ahead of its use
      if ((g_pj_xp & 128) != 0 && abl == 4 && lg <= 28) kprobe = k_pj2_probe_pipe<K, false, false, true, 4, true>;
    }
  }
  constexpr size_t lds_p = ((size_t)1 << (PJ_SUB_LOG2 - 1)) + PP_TAGPAD + (size_t)6 * PP_ROWS * sizeof(int32_t);
  const size_t lds_s     = (rec || win_soa) ? (size_t)8192 * 12 + ((size_t)8 << pbits) : (size_t)16384 * sizeof(K) + ((size_t)8 << pbits);
  static std::atomic<bool> attr_set{false};  // per instantiation
  static int num_cus     = 0;
  if (!attr_set) {
    const int lds_max = 160 * 1024 - 256;
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kspec), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kexact), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    if constexpr (sizeof(K) == 8) {
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<16, false, false, true, false, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<16, false, false, true, true, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<16, false, true, true, false, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<16, false, true, true, true, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<16, true, false, true, false, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<16, true, false, true, true, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<16, true, true, true, false, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<16, true, true, true, true, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<16, false, false, false, false, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<16, false, false, false, true, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<16, false, true, false, false, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<16, false, true, false, true, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<16, true, false, false, false, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<16, true, false, false, true, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<16, true, true, false, false, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<16, true, true, false, true, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<24, false, false, true, false, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<24, false, false, true, true, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<24, false, true, true, false, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<24, false, true, true, true, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<24, true, false, true, false, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<24, true, false, true, true, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<24, true, true, true, false, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_scatter_rec<24, true, true, true, true, F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_probe_pipe<K, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_probe_pipe<K, false, false, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_probe_pipe<K, false, false, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_probe_pipe<K, false, false, true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_probe_pipe<K, false, false, true, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_probe_pipe<K, false, false, true, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
    }
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_probe_pipe<K, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_probe_pipe<K, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_probe_pipe<K, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj2_probe_pipe<K, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
    int dev = 0;
    GX_HIP_TRY(hipGetDevice(&dev));
    GX_HIP_TRY(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev));
    attr_set = true;
  }
  const int64_t ntiles = div_up(n, (int64_t)TILE);
  const int64_t rrows  = pj_range_rows(n, TILE);
  const int xmajor     = (g_pj_xp & (1 << 24)) ? 0 : 1;  // bit 24 (measurement): the fill-counter atomics on fill[] itself, as until round 6
  int64_t grid         = num_cus > 0 ? num_cus : 256;
  if (((g_pj_xp >> 8) & 255) != 0) grid = ((g_pj_xp >> 8) & 255) * 4;  // measurement: bits 8-15 = workgroups of the partition pass / 4
  grid                 = grid / PJ_NR * PJ_NR;  // v % 8 must stay the XCD of a workgroup over its whole walk
  if (grid < PJ_NR) grid = PJ_NR;
  if (grid > ntiles) grid = ntiles;
  unsigned long long* cur = reinterpret_cast<unsigned long long*>(cursor);
  jprof_mark(0, s);
  GX_HIP_TRY(hipMemsetAsync(plan2, 0, sizeof(Pj2Plan), s));
  GX_HIP_TRY(hipMemsetAsync(plan, 0, sizeof(PjPlan), s));
  jprof_mark(1, s);
  // ---- speculative pass
  // (record form: full tiles, then the ragged tail as one more workgroup)
  const int64_t nfull = n / TILE;
  auto launch_rec = [&](bool exact) {
    if constexpr (sizeof(K) == 8) {
      const uint64_t* k64 = reinterpret_cast<const uint64_t*>(keys);
      PjRec* precs        = reinterpret_cast<PjRec*>(pkeys);
      const unsigned gf   = (unsigned)(grid < nfull ? grid : nfull);
      const int64_t tail0 = nfull * TILE;
#define GX_PJ_REC2(RPT_, EX_, AOS_, PAY_)                                                                                                       \
  do {                                                                                                                                          \
    if (gf) hipLaunchKernelGGL((k_pj2_scatter_rec<RPT_, EX_, false, AOS_, PAY_, F>), dim3(gf), dim3(1024), lds_s, s, k64, n, plan2, plan, pbits, \
                               rrows, cap, nfull, (int64_t)0, precs, part_of, row0, payload, pidx, xmajor);                                             \
    if (tail0 < n) hipLaunchKernelGGL((k_pj2_scatter_rec<RPT_, EX_, true, AOS_, PAY_, F>), dim3(1), dim3(1024), lds_s, s, k64, n, plan2, plan,   \
                                      pbits, rrows, cap, (int64_t)1, tail0, precs, part_of, row0, payload, pidx, xmajor);                               \
  } while (0)
#define GX_PJ_REC(RPT_, EX_, AOS_)                              \
  do {                                                          \
    if (payload) GX_PJ_REC2(RPT_, EX_, AOS_, true);             \
    else GX_PJ_REC2(RPT_, EX_, AOS_, false);                    \
  } while (0)
      if (rec24) {
        if (exact) GX_PJ_REC(24, true, true); else GX_PJ_REC(24, false, true);
      } else if (rec) {
        if (exact) GX_PJ_REC(16, true, true); else GX_PJ_REC(16, false, true);
      } else {
        if (exact) GX_PJ_REC(16, true, false); else GX_PJ_REC(16, false, false);
      }
#undef GX_PJ_REC2
#undef GX_PJ_REC
    }
  };
  if (rec || win_soa) launch_rec(false);
  else hipLaunchKernelGGL(kspec, dim3((unsigned)grid), dim3(1024), lds_s, s, keys, n, plan2, plan, pbits, rrows, cap, ntiles, pkeys, pidx, part_of, row0, payload, xmajor);
  jprof_mark(2, s);
  // (6 / 7: the same kernel with the tag windows read from the L2 -- the partition pass then cuts 2^20-slot sub-tables, see pj_bits)
  // the probe kernel: 0 the pipelined LDS-tag gang probe; 2 / 3 the L2-resident direct probe with 4 / 2 rows per thread (round 5,
  // measured negative); 4 / 5 the tag probe in its occupancy form (round 5: no pipeline, no staging, two workgroups per CU), 2 / 4 rows
  typedef void (*ProbeK)(const K*, const int32_t*, PieceTable, int, const Slot<K>*, uint32_t, int, int32_t*, int32_t*, int64_t, unsigned long long*);
  const int pk        = g_pj_probe;
  const bool alt      = pk >= 2;
  ProbeK kalt         = pk == 2 ? (ProbeK)k_pj3_probe_direct<K, 4, 6> : pk == 3 ? (ProbeK)k_pj3_probe_direct<K, 2, 8>
                        : pk == 4 ? (ProbeK)k_pj4_probe_tags<K, 2, 8, true> : pk == 5 ? (ProbeK)k_pj4_probe_tags<K, 4, 4, true>
                        : pk == 6 ? (ProbeK)k_pj4_probe_tags<K, 2, 8, false> : (ProbeK)k_pj4_probe_tags<K, 4, 4, false>;
  const unsigned alt_bt     = (pk == 2 || pk == 3) ? (unsigned)PD_BT : (unsigned)PT_BT;
  const unsigned alt_rpt    = (pk == 2 || pk == 5 || pk == 7) ? 4u : 2u;
  const size_t alt_lds      = (pk == 4 || pk == 5) ? ((size_t)1 << (PJ_SUB_LOG2 - 1)) + PP_TAGPAD : 0;  // (6 / 7: no LDS tags)
  const unsigned piece_rows = !alt ? (unsigned)PP_ROWS : (pk >= 4 ? (unsigned)(PT_PW * GX_WAVE) * alt_rpt : alt_bt * alt_rpt);
  static std::atomic<int> alt_wgs[6];  // resident workgroups per CU of the alternative kernels (occupancy query, once each; zero-initialised)
  if (alt && alt_wgs[pk - 2] == 0) {
    if (alt_lds) GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kalt), hipFuncAttributeMaxDynamicSharedMemorySize, (int)alt_lds));
    int nb = 0;
    GX_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kalt), (int)alt_bt, alt_lds));
    alt_wgs[pk - 2] = nb > 0 ? nb : 1;
  }
  auto launch_probe = [&](const PieceTable& t) {
    if (alt) {
      const int64_t g = (int64_t)(num_cus > 0 ? num_cus : 256) * alt_wgs[pk - 2];
      hipLaunchKernelGGL(kalt, dim3((unsigned)g), dim3(alt_bt), alt_lds, s, pkeys, pidx, t, pbits, slots, lg, left_outer, out_probe, out_build, capacity, cur);
    } else {
      int64_t g = num_cus > 0 ? num_cus : 256;
      if (((g_pj_xp >> 16) & 255) != 0) g = ((g_pj_xp >> 16) & 255) * 4;  // measurement: bits 16-23 = workgroups of the probe / 4
      if (g > 256 && longp) g = 256;  // (the overflow list has 256 slices)
      hipLaunchKernelGGL(kprobe, dim3((unsigned)g), dim3(PP_BT), lds_p, s, pkeys, pidx, t, pbits, slots, lg, left_outer, out_probe, out_build, capacity, cur);
      if (t.ovf)
        hipLaunchKernelGGL((k_pj2_probe_rare<K>), dim3((unsigned)g * PR_SUB), dim3(256), 0, s, t.ovf, t.ovf_cap, t.ovf_count, slots, lg, left_outer, out_probe, out_build,
                           capacity, cur);
    }
  };
  hipLaunchKernelGGL(k_pj2_offsets, dim3(1), dim3(1024), 0, s, plan2, pbits, cap, piece_rows, xmajor);
  PieceTable pt{plan2->chunk0, plan2->list_chunk0, plan2->ticket, nullptr, plan2->fill, cap, PJ_NR, 0u, nullptr, nullptr, 0u, nullptr};
  if (longp && !alt) {
    pt.ovf       = ovf;
    pt.ovf_cap   = ovf_cap;
    pt.ovf_count = ovf_count;
  }
  if ((g_pj_xp & 4) != 0 && !alt) {  // round 6: pipelined service wave on fixed pieces per region
    pt.fixedk = (cap + piece_rows - 1) / piece_rows;
    pt.gate   = &plan2->fallback;
  }
  launch_probe(pt);
  // ---- exact sequence: every kernel returns at once unless plan2->fallback is set
  int64_t hb = div_up(n, 256 * 8 * 4 * PJ_NR);
  if (hb > 256) hb = 256;
  if (hb < 1) hb = 1;
  hipLaunchKernelGGL((k_pj_hist<K, F>), dim3((unsigned)(hb * PJ_NR)), dim3(256), 0, s, keys, n, plan, pbits, rrows, part_of, &plan2->fallback);
  hipLaunchKernelGGL(k_pj_offsets, dim3(1), dim3(1024), 0, s, plan, pbits, piece_rows, &plan2->fallback);
  if (rec || win_soa) launch_rec(true);
  else hipLaunchKernelGGL(kexact, dim3((unsigned)grid), dim3(1024), lds_s, s, keys, n, plan2, plan, pbits, rrows, cap, ntiles, pkeys, pidx, part_of, row0, payload, xmajor);
  PieceTable pe{plan->chunk0, plan->list_chunk0, plan2->ticket_exact, plan->offset, nullptr, 0u, 1, 0u, nullptr, pt.ovf, pt.ovf_cap, pt.ovf_count};
  launch_probe(pe);
  jprof_mark(3, s);
  g_jprof.marked = g_jprof.enabled;
  GX_LAUNCH_CHECK();
  return 0;
}

template <typename K>
int probe_partitioned_impl(const void* keys, int64_t n, const void* table, size_t table_bytes, uint32_t lg,
                           int left_outer, int32_t* out_probe, int32_t* out_build, int64_t capacity, int64_t* cursor,
                           void* tmp, size_t* tmp_bytes, hipStream_t s, int32_t row0 = 0, const int32_t* payload = nullptr)
{
  int pbits = pj_bits(lg, (int)sizeof(Slot<K>));
  // knob 6 / 7 (the tag windows come from the L2, nothing is staged in LDS): 2^20-slot sub-tables -- an eighth of the partitions
  if ((g_pj_probe == 6 || g_pj_probe == 7) && pbits >= 6) pbits -= 3;
  if (pj2_applies<K>(n, pbits)) {
    if (tmp) {
      const size_t need = sizeof(TableHeader) + (sizeof(Slot<K>) << lg) + ((size_t)1 << lg) / 2;
      if (table_bytes < need) return GX_ETMP;
    }
    return probe_partitioned_impl2<K>(static_cast<const K*>(keys), n,
                                      reinterpret_cast<const Slot<K>*>(static_cast<const char*>(table) + sizeof(TableHeader)), lg, pbits,
                                      left_outer, out_probe, out_build, capacity, cursor, tmp, tmp_bytes, s, row0, payload);
  }
  Carver c(tmp);
  PjPlan* plan   = c.take<PjPlan>(1);
  K* pkeys       = c.take<K>((size_t)n);
  int32_t* pidx  = c.take<int32_t>((size_t)n);
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  const size_t need = sizeof(TableHeader) + (sizeof(Slot<K>) << lg) + ((size_t)1 << lg) / 2;
  if (table_bytes < need) return GX_ETMP;
  if (n == 0) return 0;
  const Slot<K>* slots = reinterpret_cast<const Slot<K>*>(static_cast<const char*>(table) + sizeof(TableHeader));
  if (pbits == 0) return GX_EINVAL;  // caller should use gx_join_probe
  const bool use_tags = true;  // (the tag-less probe of round 1 is kept for reference: 14.3 ms against 9.7 ms with tags)
  const bool use_pipe = use_tags && g_pj_probe != 1;
  // the tag probe amortises its 64 KiB tag copy over several PJ_CHUNKs of the same partition
  unsigned int chunk_rows = PJ_CHUNK;
  if (use_tags) {
    const int mult = 8;
    chunk_rows     = PJ_CHUNK * (unsigned)(mult < 1 ? 1 : mult);
    while (chunk_rows > PJ_CHUNK && div_up(n, (int64_t)chunk_rows) < 4096) chunk_rows /= 2;  // keep >> 512 workgroups
    if (use_pipe) chunk_rows = PP_ROWS;  // the persistent probe takes tickets per 3840-row piece
  }
  {
    int rc = pj_partition<K>(static_cast<const K*>(keys), n, pbits, plan, pkeys, pidx, chunk_rows, s, true, row0, payload);
    if (rc) return rc;
  }
  const int64_t max_chunks = div_up(n, (int64_t)chunk_rows) + (1 << pbits);
  if (use_pipe) {
    constexpr size_t lds_p = ((size_t)1 << (PJ_SUB_LOG2 - 1)) + (size_t)6 * PP_ROWS * sizeof(int32_t);
    static std::atomic<bool> pattr_set{false};
    static int num_cus     = 0;
    if (!pattr_set) {
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj_probe_pipe<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
      int dev = 0;
      GX_HIP_TRY(hipGetDevice(&dev));
      GX_HIP_TRY(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev));
      pattr_set = true;
    }
    // persistent workgroups, one per CU (the LDS admits no second one); pieces come from per-XCD tickets
    int64_t grid = num_cus > 0 ? num_cus : 256;
    if (grid > max_chunks) grid = max_chunks;
    hipLaunchKernelGGL((k_pj_probe_pipe<K>), dim3((unsigned)grid), dim3(PP_BT), lds_p, s, pkeys, pidx, plan, pbits, slots, lg,
                       left_outer, out_probe, out_build, capacity, reinterpret_cast<unsigned long long*>(cursor));
  } else if (use_tags) {
    constexpr size_t lds_t = (size_t)1 << (PJ_SUB_LOG2 - 1);
    static std::atomic<bool> tattr_set{false};
    if (!tattr_set) {
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pj_probe_tags<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
      tattr_set = true;
    }
    hipLaunchKernelGGL((k_pj_probe_tags<K>), dim3((unsigned)max_chunks), dim3(PJ_BT), lds_t, s, pkeys, pidx, plan, pbits, slots, lg,
                       left_outer, out_probe, out_build, capacity, reinterpret_cast<unsigned long long*>(cursor), chunk_rows);
  } else {
    hipLaunchKernelGGL((k_pj_probe<K>), dim3((unsigned)max_chunks), dim3(PJ_BT), 0, s, pkeys, pidx, plan, pbits, slots, lg,
                       left_outer, out_probe, out_build, capacity, reinterpret_cast<unsigned long long*>(cursor));
  }
  jprof_mark(3, s);
  g_jprof.marked = g_jprof.enabled;
  GX_LAUNCH_CHECK();
  return 0;
}

template <typename K>
int build_partitioned_impl(const void* keys, int64_t n, void* table, size_t table_bytes, double load_factor, void* tmp,
                           size_t* tmp_bytes, hipStream_t s, const int32_t* payload = nullptr)
{
  const uint32_t lg = log2_capacity(n, load_factor);
  const int pbits   = pj_bits(lg, (int)sizeof(Slot<K>));
  Carver c(tmp);
  PjPlan* plan  = c.take<PjPlan>(1);
  K* pkeys      = c.take<K>((size_t)n);
  int32_t* pidx = c.take<int32_t>((size_t)n);
  BuildFix* fix = c.take<BuildFix>(1);
  const bool windows = g_pj_build == 0 && (int)lg - pbits == PJ_SUB_LOG2;  // the window build (round 4b)
  K* wkeys        = windows ? c.take<K>((size_t)n) : nullptr;
  int32_t* widx   = windows ? c.take<int32_t>((size_t)n) : nullptr;
  uint32_t* woffs = windows ? c.take<uint32_t>(((size_t)1 << pbits) * (BW_PER + 1)) : nullptr;
  BuildFix2* fix2 = windows ? c.take<BuildFix2>(1) : nullptr;
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  const size_t need = sizeof(TableHeader) + (sizeof(Slot<K>) << lg) + ((size_t)1 << lg) / 2;
  if (table_bytes < need) return GX_ETMP;
  if (pbits == 0) return GX_EINVAL;
  char* base = static_cast<char*>(table);
  if (!windows || n == 0) GX_HIP_TRY(hipMemsetAsync(base + sizeof(TableHeader), 0xFF, sizeof(Slot<K>) << lg, s));
  if (n == 0) return launch_tags<K>(table, lg, s);
  int rc = pj_partition<K>(static_cast<const K*>(keys), n, pbits, plan, pkeys, pidx, PJ_CHUNK, s, false, 0, payload);
  if (rc) return rc;
  if (windows) {  // every window of the table is composed in LDS and written once, in full lines: no pre-fill, no global atomics
    uint8_t* tags = reinterpret_cast<uint8_t*>(base + sizeof(TableHeader) + (sizeof(Slot<K>) << lg));
    auto* slots   = reinterpret_cast<Slot<K>*>(base + sizeof(TableHeader));
    GX_HIP_TRY(hipMemsetAsync(fix2, 0, 128, s));
    hipLaunchKernelGGL((k_bw_split<K>), dim3(1u << pbits), dim3(BW_BT), 0, s, (const K*)pkeys, (const int32_t*)pidx, (const PjPlan*)plan, wkeys, widx, woffs, lg);
    hipLaunchKernelGGL((k_bw_build<K>), dim3(1u << pbits), dim3(BW_BT), 0, s, (const K*)wkeys, (const int32_t*)widx, (const PjPlan*)plan, (const uint32_t*)woffs,
                       slots, tags, lg, fix2);
    hipLaunchKernelGGL((k_bw_fixup<K>), dim3(256), dim3(256), 0, s, slots, tags, lg, (const BuildFix2*)fix2);
    // gated on fix2->failed (no-ops otherwise): the round-2 build from the level-1 partition
    const unsigned int* gate = &fix2->failed;
    hipLaunchKernelGGL(k_bw_fill_empty, dim3(2048), dim3(256), 0, s, reinterpret_cast<uint4*>(slots), (sizeof(Slot<K>) << lg) / 16, gate);
    const int64_t max_chunks_f = div_up(n, PJ_CHUNK) + (1 << pbits);
    hipLaunchKernelGGL((k_pj_build<K>), dim3((unsigned)max_chunks_f), dim3(PJ_BT), 0, s, pkeys, pidx, plan, pbits, slots, lg, gate);
    GX_LAUNCH_CHECK();
    return launch_tags<K>(table, lg, s, gate);
  }
  if (g_pj_build == 2 && (int)lg - pbits == PJ_SUB_LOG2) {  // one workgroup per sub-table: tags in LDS are the occupancy map
    uint8_t* tags = reinterpret_cast<uint8_t*>(base + sizeof(TableHeader) + (sizeof(Slot<K>) << lg));
    auto* slots   = reinterpret_cast<Slot<K>*>(base + sizeof(TableHeader));
    GX_HIP_TRY(hipMemsetAsync(fix, 0, 128, s));
    const size_t lds = (size_t)(1u << PJ_SUB_LOG2) / 2;
    auto kb          = k_bs_build<K>;
    static std::atomic<bool> attr_set{false};  // per instantiation
    if (!attr_set) {
      GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_set = true;
    }
    hipLaunchKernelGGL(kb, dim3(1u << pbits), dim3(BS_BT), lds, s, pkeys, pidx, plan, slots, tags, lg, fix);
    hipLaunchKernelGGL((k_bs_fixup<K>), dim3(256), dim3(256), 0, s, pkeys, pidx, n, slots, tags, lg, fix, 0);
    hipLaunchKernelGGL((k_bs_fixup<K>), dim3(1024), dim3(256), 0, s, pkeys, pidx, n, slots, tags, lg, fix, 1);
    GX_LAUNCH_CHECK();
    return 0;
  }
  const int64_t max_chunks = div_up(n, PJ_CHUNK) + (1 << pbits);
  hipLaunchKernelGGL((k_pj_build<K>), dim3((unsigned)max_chunks), dim3(PJ_BT), 0, s, pkeys, pidx, plan, pbits,
                     reinterpret_cast<Slot<K>*>(base + sizeof(TableHeader)), lg);
  GX_LAUNCH_CHECK();
  return launch_tags<K>(table, lg, s);
}

template <typename K>
int build_impl(const void* keys, const uint32_t* valid, int64_t n, void* table, size_t table_bytes,
               double load_factor, hipStream_t s, const int32_t* payload = nullptr)
{
  const uint32_t lg = log2_capacity(n, load_factor);
  const size_t need = sizeof(TableHeader) + (sizeof(Slot<K>) << lg) + ((size_t)1 << lg) / 2;
  if (table_bytes < need) return GX_ETMP;
  char* base = static_cast<char*>(table);
  GX_HIP_TRY(hipMemsetAsync(base + sizeof(TableHeader), 0xFF, sizeof(Slot<K>) << lg, s));
  if (n > 0) {
    int64_t blocks = div_up(n, JBT * 4);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((k_build<K>), dim3((unsigned)blocks), dim3(JBT), 0, s, static_cast<const K*>(keys), valid, n,
                       reinterpret_cast<Slot<K>*>(base + sizeof(TableHeader)), lg, payload);
    GX_LAUNCH_CHECK();
  }
  return launch_tags<K>(table, lg, s);
}

template <typename K, bool WRITE>
int probe_impl(const void* keys, const uint32_t* valid, int64_t n, const void* table, size_t table_bytes,
               uint32_t lg, int left_outer, int32_t* out_probe, int32_t* out_build, int64_t capacity,
               int64_t* cursor, hipStream_t s)
{
  const size_t need = sizeof(TableHeader) + (sizeof(Slot<K>) << lg);
  if (table_bytes < need) return GX_ETMP;
  if (n == 0) return 0;
  int64_t blocks = div_up(n, JCHUNK);
  if (blocks > 256 * 16) blocks = 256 * 16;
  const char* base = static_cast<const char*>(table);
  hipLaunchKernelGGL((k_probe<K, WRITE>), dim3((unsigned)blocks), dim3(JBT), 0, s, static_cast<const K*>(keys), valid,
                     n, reinterpret_cast<const Slot<K>*>(base + sizeof(TableHeader)), lg, left_outer, out_probe,
                     out_build, capacity, reinterpret_cast<unsigned long long*>(cursor));
  GX_LAUNCH_CHECK();
  return 0;
}

}  // namespace join
}  // namespace gx

extern "C" {

size_t gx_join_table_bytes(int key_size, int64_t build_rows, double load_factor)
{
  if (key_size != 4 && key_size != 8) return 0;
  const uint32_t lg = gx::join::log2_capacity(build_rows, load_factor);
  return sizeof(gx::join::TableHeader) + ((size_t)(key_size == 8 ? 16 : 8) << lg) + ((size_t)1 << lg) / 2;  // slots + 4-bit tags
}

int gx_join_build(int key_size, const void* build_keys, const uint32_t* build_valid, int64_t build_rows,
                  void* table, size_t table_bytes, double load_factor, gx_stream_t s)
{
  return gx_join_build_pl(key_size, build_keys, nullptr, build_valid, build_rows, table, table_bytes, load_factor, s);
}
int gx_join_build_pl(int key_size, const void* build_keys, const int32_t* payload, const uint32_t* build_valid, int64_t build_rows,
                     void* table, size_t table_bytes, double load_factor, gx_stream_t s)
{
  if (build_rows < 0 || !table || (build_rows > 0 && !build_keys)) return GX_EINVAL;
  if (key_size == 8) return gx::join::build_impl<uint64_t>(build_keys, build_valid, build_rows, table, table_bytes, load_factor, s, payload);
  if (key_size == 4) return gx::join::build_impl<uint32_t>(build_keys, build_valid, build_rows, table, table_bytes, load_factor, s, payload);
  return GX_EDTYPE;
}

// capacity (log2) is recomputed from the table size so the table blob needs no host round trip
static uint32_t gx_join_log2_from_bytes(int key_size, size_t table_bytes)
{
  const size_t slot = key_size == 8 ? 16 : 8;
  size_t slots      = (table_bytes - sizeof(gx::join::TableHeader)) / slot;
  uint32_t lg       = 0;
  while ((2ull << lg) <= slots) ++lg;
  return lg;
}

int gx_join_count(int key_size, const void* probe_keys, const uint32_t* probe_valid, int64_t probe_rows,
                  const void* table, size_t table_bytes, int64_t* count_dev, gx_stream_t s)
{
  if (probe_rows < 0 || !table || !count_dev || (probe_rows > 0 && !probe_keys)) return GX_EINVAL;
  if (table_bytes <= sizeof(gx::join::TableHeader)) return GX_ETMP;
  GX_HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(int64_t), s));
  const uint32_t lg = gx_join_log2_from_bytes(key_size, table_bytes);
  if (key_size == 8) return gx::join::probe_impl<uint64_t, false>(probe_keys, probe_valid, probe_rows, table, table_bytes, lg, 0, nullptr, nullptr, 0, count_dev, s);
  if (key_size == 4) return gx::join::probe_impl<uint32_t, false>(probe_keys, probe_valid, probe_rows, table, table_bytes, lg, 0, nullptr, nullptr, 0, count_dev, s);
  return GX_EDTYPE;
}

int gx_join_probe(int key_size, const void* probe_keys, const uint32_t* probe_valid, int64_t probe_rows,
                  const void* table, size_t table_bytes, int left_outer, int32_t* out_probe_idx,
                  int32_t* out_build_idx, int64_t capacity, int64_t* cursor_dev, gx_stream_t s)
{
  if (probe_rows < 0 || capacity < 0 || !table || !cursor_dev || (probe_rows > 0 && !probe_keys)) return GX_EINVAL;
  if (capacity > 0 && (!out_probe_idx || !out_build_idx)) return GX_EINVAL;
  if (table_bytes <= sizeof(gx::join::TableHeader)) return GX_ETMP;
  const uint32_t lg = gx_join_log2_from_bytes(key_size, table_bytes);
  if (key_size == 8) return gx::join::probe_impl<uint64_t, true>(probe_keys, probe_valid, probe_rows, table, table_bytes, lg, left_outer, out_probe_idx, out_build_idx, capacity, cursor_dev, s);
  if (key_size == 4) return gx::join::probe_impl<uint32_t, true>(probe_keys, probe_valid, probe_rows, table, table_bytes, lg, left_outer, out_probe_idx, out_build_idx, capacity, cursor_dev, s);
  return GX_EDTYPE;
}


/* see gx.h */
int gx_join_probe_partitioned_at(int key_size, const void* probe_keys, int64_t probe_rows, int32_t row_base, const void* table,
                                 size_t table_bytes, int left_outer, int32_t* out_probe_idx, int32_t* out_build_idx,
                                 int64_t capacity, int64_t* cursor_dev, void* tmp, size_t* tmp_bytes, gx_stream_t s)
{
  return gx_join_probe_partitioned_pl(key_size, probe_keys, nullptr, probe_rows, row_base, table, table_bytes, left_outer, out_probe_idx,
                                      out_build_idx, capacity, cursor_dev, tmp, tmp_bytes, s);
}
int gx_join_probe_partitioned_pl(int key_size, const void* probe_keys, const int32_t* payload, int64_t probe_rows, int32_t row_base,
                                 const void* table, size_t table_bytes, int left_outer, int32_t* out_probe_idx, int32_t* out_build_idx,
                                 int64_t capacity, int64_t* cursor_dev, void* tmp, size_t* tmp_bytes, gx_stream_t s)
{
  if (probe_rows < 0 || capacity < 0 || !table || !tmp_bytes || (probe_rows > 0 && !probe_keys)) return GX_EINVAL;
  if (tmp && (!cursor_dev || (capacity > 0 && (!out_probe_idx || !out_build_idx)))) return GX_EINVAL;
  if (table_bytes <= sizeof(gx::join::TableHeader)) return GX_ETMP;
  const uint32_t lg = gx_join_log2_from_bytes(key_size, table_bytes);
  if (key_size == 8)
    return gx::join::probe_partitioned_impl<uint64_t>(probe_keys, probe_rows, table, table_bytes, lg, left_outer, out_probe_idx,
                                                      out_build_idx, capacity, cursor_dev, tmp, tmp_bytes, s, row_base, payload);
  if (key_size == 4)
    return gx::join::probe_partitioned_impl<uint32_t>(probe_keys, probe_rows, table, table_bytes, lg, left_outer, out_probe_idx,
                                                      out_build_idx, capacity, cursor_dev, tmp, tmp_bytes, s, row_base, payload);
  return GX_EDTYPE;
}
int gx_join_probe_partitioned(int key_size, const void* probe_keys, int64_t probe_rows, const void* table,
                              size_t table_bytes, int left_outer, int32_t* out_probe_idx, int32_t* out_build_idx,
                              int64_t capacity, int64_t* cursor_dev, void* tmp, size_t* tmp_bytes, gx_stream_t s)
{
  return gx_join_probe_partitioned_at(key_size, probe_keys, probe_rows, 0, table, table_bytes, left_outer, out_probe_idx, out_build_idx,
                                      capacity, cursor_dev, tmp, tmp_bytes, s);
}

int gx_join_lookup(int key_size, const void* probe_keys, const uint32_t* probe_valid, int64_t probe_rows,
                   const void* table, size_t table_bytes, int32_t* out_build_idx, gx_stream_t s)
{
  if (probe_rows < 0 || (probe_rows > 0 && (!probe_keys || !out_build_idx)) || !table) return GX_EINVAL;
  if (table_bytes <= sizeof(gx::join::TableHeader)) return GX_ETMP;
  if (probe_rows == 0) return 0;
  const uint32_t lg = gx_join_log2_from_bytes(key_size, table_bytes);
  int64_t blocks    = gx::div_up(probe_rows, (int64_t)256 * 4);
  if (blocks > 16384) blocks = 16384;
  const char* base = static_cast<const char*>(table) + sizeof(gx::join::TableHeader);
  if (key_size == 8)
    hipLaunchKernelGGL((gx::join::k_lookup<uint64_t>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s,
                       static_cast<const uint64_t*>(probe_keys), probe_valid, probe_rows,
                       reinterpret_cast<const gx::join::Slot<uint64_t>*>(base), lg, out_build_idx);
  else if (key_size == 4)
    hipLaunchKernelGGL((gx::join::k_lookup<uint32_t>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s,
                       static_cast<const uint32_t*>(probe_keys), probe_valid, probe_rows,
                       reinterpret_cast<const gx::join::Slot<uint32_t>*>(base), lg, out_build_idx);
  else
    return GX_EDTYPE;
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_join_filter(int key_size, const void* probe_keys, const uint32_t* probe_valid, int64_t probe_rows,
                   const void* table, size_t table_bytes, int anti, int null_matches, int32_t* out_probe_idx,
                   int64_t* count_dev, void* tmp, size_t* tmp_bytes, gx_stream_t s)
{
  if (!tmp_bytes || probe_rows < 0) return GX_EINVAL;
  if (tmp && (!table || !count_dev || (probe_rows > 0 && (!probe_keys || !out_probe_idx)))) return GX_EINVAL;
  if (tmp && table_bytes <= sizeof(gx::join::TableHeader)) return GX_ETMP;
  const uint32_t lg = tmp ? gx_join_log2_from_bytes(key_size, table_bytes) : 0;
  if (key_size == 8)
    return gx::join::filter_impl<uint64_t>(probe_keys, probe_valid, probe_rows, table, table_bytes, lg, anti, null_matches,
                                           out_probe_idx, count_dev, tmp, tmp_bytes, (hipStream_t)s);
  if (key_size == 4)
    return gx::join::filter_impl<uint32_t>(probe_keys, probe_valid, probe_rows, table, table_bytes, lg, anti, null_matches,
                                           out_probe_idx, count_dev, tmp, tmp_bytes, (hipStream_t)s);
  return GX_EDTYPE;
}

/* see gx.h */
int gx_partition_rows(int key_dtype, const void* keys, int64_t n, int mode, int nparts, const void* splitters_host, void* out_keys,
                      int32_t* out_rows, int64_t* offsets_dev, void* tmp, size_t* tmp_bytes, gx_stream_t s)
{
  return gx_partition_rows_at(key_dtype, keys, n, 0, mode, nparts, splitters_host, out_keys, out_rows, offsets_dev, tmp, tmp_bytes, s);
}

int gx_partition_rows_at(int key_dtype, const void* keys, int64_t n, int32_t row0, int mode, int nparts, const void* splitters_host,
                         void* out_keys, int32_t* out_rows, int64_t* offsets_dev, void* tmp, size_t* tmp_bytes, gx_stream_t s)
{
  return gx_partition_rows_spec_at(key_dtype, keys, n, row0, mode, nparts, splitters_host, 0, out_keys, out_rows, offsets_dev, tmp, tmp_bytes, s);
}

int gx_partition_rows_spec_at(int key_dtype, const void* keys, int64_t n, int32_t row0, int mode, int nparts, const void* splitters_host,
                              int64_t cap_rows, void* out_keys, int32_t* out_rows, int64_t* offsets_dev, void* tmp, size_t* tmp_bytes,
                              gx_stream_t s)
{
  if (cap_rows < 0 || (int64_t)nparts * cap_rows > 0xFFFFFFFFll) return GX_EINVAL;
  const uint32_t cap = (uint32_t)cap_rows;
  using namespace gx;
  using namespace gx::join;
  if (n < 0 || nparts < 1 || nparts > PJ_MAX_SPLIT + 1 || !tmp_bytes || (mode != 0 && mode != 1)) return GX_EINVAL;
  if (tmp && (!offsets_dev || (n > 0 && (!keys || !out_keys)) || (mode == 1 && nparts > 1 && !splitters_host))) return GX_EINVAL;
  if (tmp && n == 0) {  // nothing to place: every partition is empty
    GX_HIP_TRY(hipMemsetAsync(offsets_dev, 0, sizeof(int64_t) * (size_t)(nparts + 1), (hipStream_t)s));
    return 0;
  }
  if (mode == 0) {
    int pbits = 0;
    while ((1 << pbits) < nparts) ++pbits;
    switch (gx_dtype_size(key_dtype)) {
      case 8: return partition_rows_hash<uint64_t>(keys, n, pbits, nparts, out_keys, out_rows, offsets_dev, tmp, tmp_bytes, s, row0, cap);
      case 4: return partition_rows_hash<uint32_t>(keys, n, pbits, nparts, out_keys, out_rows, offsets_dev, tmp, tmp_bytes, s, row0, cap);
      default: return GX_EDTYPE;
    }
  }
  switch (key_dtype) {
    case GX_INT64: return partition_rows_range<uint64_t, K_SIGNED>(keys, n, nparts, splitters_host, out_keys, out_rows, offsets_dev, tmp, tmp_bytes, s, row0, cap);
    case GX_UINT64: return partition_rows_range<uint64_t, K_UNSIGNED>(keys, n, nparts, splitters_host, out_keys, out_rows, offsets_dev, tmp, tmp_bytes, s, row0, cap);
    case GX_FLOAT64: return partition_rows_range<uint64_t, K_FLOAT>(keys, n, nparts, splitters_host, out_keys, out_rows, offsets_dev, tmp, tmp_bytes, s, row0, cap);
    case GX_INT32: return partition_rows_range<uint32_t, K_SIGNED>(keys, n, nparts, splitters_host, out_keys, out_rows, offsets_dev, tmp, tmp_bytes, s, row0, cap);
    case GX_UINT32: return partition_rows_range<uint32_t, K_UNSIGNED>(keys, n, nparts, splitters_host, out_keys, out_rows, offsets_dev, tmp, tmp_bytes, s, row0, cap);
    case GX_FLOAT32: return partition_rows_range<uint32_t, K_FLOAT>(keys, n, nparts, splitters_host, out_keys, out_rows, offsets_dev, tmp, tmp_bytes, s, row0, cap);
    default: return GX_EDTYPE;
  }
}

int gx_join_count_rows(int key_size, const void* probe_keys, const uint32_t* probe_valid, int64_t probe_rows, const void* table,
                       size_t table_bytes, int32_t min_count, int32_t* counts, gx_stream_t s)
{
  if (probe_rows < 0 || !table || (probe_rows > 0 && (!probe_keys || !counts))) return GX_EINVAL;
  if (table_bytes <= sizeof(gx::join::TableHeader)) return GX_ETMP;
  if (probe_rows == 0) return 0;
  const uint32_t lg = gx_join_log2_from_bytes(key_size, table_bytes);
  int64_t blocks    = gx::div_up(probe_rows, (int64_t)256 * 4);
  if (blocks > 16384) blocks = 16384;
  const char* base = static_cast<const char*>(table) + sizeof(gx::join::TableHeader);
  if (key_size == 8)
    hipLaunchKernelGGL((gx::join::k_count_rows<uint64_t>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s,
                       static_cast<const uint64_t*>(probe_keys), probe_valid, probe_rows,
                       reinterpret_cast<const gx::join::Slot<uint64_t>*>(base), lg, min_count, counts);
  else if (key_size == 4)
    hipLaunchKernelGGL((gx::join::k_count_rows<uint32_t>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s,
                       static_cast<const uint32_t*>(probe_keys), probe_valid, probe_rows,
                       reinterpret_cast<const gx::join::Slot<uint32_t>*>(base), lg, min_count, counts);
  else
    return GX_EDTYPE;
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_add_i32(int32_t* data, int64_t n, int32_t value, gx_stream_t s)
{
  if (n < 0 || (n > 0 && !data)) return GX_EINVAL;
  if (n == 0 || value == 0) return 0;
  int64_t blocks = gx::div_up(n, (int64_t)256 * 8);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gx::join::k_add_i32, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, data, n, value);
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_join_profile(int enable)
{
  for (auto& p : gx::join::g_jprofs) {
    if (enable && !p.created) {
      for (auto& e : p.ev) GX_HIP_TRY(hipEventCreate(&e));
      p.created = true;
    }
    p.enabled = enable != 0;
  }
  return 0;
}

int gx_join_profile_slot(int slot)
{
  if (slot < 0 || slot >= gx::join::JPROF_SLOTS) return GX_EINVAL;
  gx::join::g_jprof_slot = slot;
  return 0;
}

int gx_join_profile_read(float* ms3)
{
  auto& p = gx::join::g_jprofs[gx::join::g_jprof_slot];
  if (!p.created || !p.marked || !ms3) return GX_EINVAL;
  GX_HIP_TRY(hipEventSynchronize(p.ev[3]));
  for (int i = 0; i < 3; ++i) GX_HIP_TRY(hipEventElapsedTime(&ms3[i], p.ev[i], p.ev[i + 1]));
  return 0;
}

void gx_join_set_probe_kernel(int which) { gx::join::g_pj_probe = (which >= 1 && which <= 7) ? which : 0; }
void gx_join_set_partition_mode(int speculative, int early_loads)
{
  gx::join::g_pj_spec        = speculative == 2 ? 2 : (speculative ? 1 : 0);
  gx::join::g_pj_probe_early = early_loads & 1;
  gx::join::g_pj_defer       = (early_loads & 2) ? 1 : 0;  // bit 1 of early_loads: park unsettled rows in the per-wave queue (A/B)
}

void gx_join_set_build_kernel(int which) { gx::join::g_pj_build = (which == 1 || which == 2) ? which : 0; }
void gx_join_set_experiment(int bits) { gx::join::g_pj_xp = bits; }
void gx_join_set_overflow_slice(int rows) { gx::join::g_pj_ovf_cap = rows > 0 ? rows : 0; }
void gx_join_set_scatter_tile(int rows)
{
  gx::join::g_pj_tile = (rows == 4096 || rows == 8192 || rows == 16384) ? rows : 0;
}

int gx_join_partition_bits(int key_size, size_t table_bytes)
{
  if ((key_size != 4 && key_size != 8) || table_bytes <= sizeof(gx::join::TableHeader)) return 0;
  return gx::join::pj_bits(gx_join_log2_from_bytes(key_size, table_bytes), key_size == 8 ? 16 : 8);
}


/* see gx.h */
int gx_join_build_partitioned(int key_size, const void* build_keys, int64_t build_rows, void* table, size_t table_bytes,
                              double load_factor, void* tmp, size_t* tmp_bytes, gx_stream_t s)
{
  return gx_join_build_partitioned_pl(key_size, build_keys, nullptr, build_rows, table, table_bytes, load_factor, tmp, tmp_bytes, s);
}
int gx_join_build_partitioned_pl(int key_size, const void* build_keys, const int32_t* payload, int64_t build_rows, void* table,
                                 size_t table_bytes, double load_factor, void* tmp, size_t* tmp_bytes, gx_stream_t s)
{
  if (build_rows < 0 || !table || !tmp_bytes || (build_rows > 0 && !build_keys)) return GX_EINVAL;
  if (key_size == 8) return gx::join::build_partitioned_impl<uint64_t>(build_keys, build_rows, table, table_bytes, load_factor, tmp, tmp_bytes, s, payload);
  if (key_size == 4) return gx::join::build_partitioned_impl<uint32_t>(build_keys, build_rows, table, table_bytes, load_factor, tmp, tmp_bytes, s, payload);
  return GX_EDTYPE;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// full join complement: build rows that no probe row matched, as (JoinNoMatch, build_row) pairs
// appended behind a left-join result (cpp/src/join/join_utils.cu:86-157 get_left_join_indices_complement).
// ------------------------------------------------------------------------------------------------
namespace gx {
namespace join {

__global__ void __launch_bounds__(256) k_mark_matched(const int32_t* __restrict__ build_idx, int64_t n,
                                                      uint32_t* matched_bits)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int32_t r = build_idx[i];
    if (r >= 0) atomicOr(&matched_bits[r >> 5], 1u << (r & 31));
  }
}

__global__ void __launch_bounds__(256) k_emit_unmatched(const uint32_t* __restrict__ matched_bits, int64_t build_rows,
                                                        int32_t* out_probe, int32_t* out_build, int64_t capacity,
                                                        unsigned long long* cursor)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < build_rows; i0 += stride) {
    const int64_t i  = i0 + threadIdx.x;
    const bool un    = i < build_rows && !((matched_bits[i >> 5] >> (i & 31)) & 1u);
    const uint64_t b = ballot(un);
    if (b == 0) continue;
    unsigned long long base = 0;
    if (lane_id() == (unsigned)__builtin_ctzll(b)) base = atomicAdd(cursor, (unsigned long long)__builtin_popcountll(b));
    base = shfl(base, __builtin_ctzll(b));
    if (un) {
      const unsigned long long pos = base + (unsigned long long)__builtin_popcountll(b & lanemask_lt());
      if ((int64_t)pos < capacity) {
        out_probe[pos] = NO_MATCH;
        out_build[pos] = (int32_t)i;
      }
    }
  }
}

}  // namespace join
}  // namespace gx

extern "C" {

/* tmp: (build_rows + 31) / 32 uint32 words (query with tmp == NULL).  *cursor_dev must hold the number
 * of pairs already in out_* (the left-join result); unmatched build rows are appended from there. */
int gx_join_complement(const int32_t* build_idx, int64_t n, int64_t build_rows, int32_t* out_probe_idx,
                       int32_t* out_build_idx, int64_t capacity, int64_t* cursor_dev, void* tmp, size_t* tmp_bytes,
                       gx_stream_t s)
{
  if (n < 0 || build_rows < 0 || !tmp_bytes) return GX_EINVAL;
  const size_t need = gx::align_up((size_t)((build_rows + 31) / 32) * 4 + 4, 256);
  if (!tmp) {
    *tmp_bytes = need;
    return 0;
  }
  if (*tmp_bytes < need) return GX_ETMP;
  if (!cursor_dev) return GX_EINVAL;
  GX_HIP_TRY(hipMemsetAsync(tmp, 0, need, s));
  if (n > 0) {
    int64_t blocks = gx::div_up(n, 256 * 8);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gx::join::k_mark_matched, dim3((unsigned)blocks), dim3(256), 0, s, build_idx, n, static_cast<uint32_t*>(tmp));
  }
  if (build_rows > 0) {
    int64_t blocks = gx::div_up(build_rows, 256 * 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gx::join::k_emit_unmatched, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint32_t*>(tmp),
                       build_rows, out_probe_idx, out_build_idx, capacity, reinterpret_cast<unsigned long long*>(cursor_dev));
  }
  GX_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
