/*
 * oracle.c -- plain-C CPU restatement of the cudf hot path (TEST INFRASTRUCTURE ONLY).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library
 * built from this file (oracle/liboracle.so); the product path (cudf_amd/) never does.
 * It is a second, faster restatement next to oracle/cudf_oracle.py (NumPy) so that parity can
 * be checked at 1e7-1e8 rows in seconds and so the CPU baseline is compiled code.
 *
 * Parity pinning: cross-checked against oracle/cudf_oracle.py and, through it, against the
 * reference's golden vectors (tests/golden/reference_vectors.py) in tests/test_oracle_golden.py.
 * The reference's digit passes and hash-table loops live in CCCL/cuCollections, which are not
 * vendored under /root/reference; what is restated here is their published algorithm at the
 * reference's call sites:
 *   - LSD radix sort, 8-bit digits, stable:  cpp/src/sort/sort_radix.cu:58-78 (SortKeys),
 *     cpp/src/sort/sorted_order_radix.cu:63-96 (SortPairs with iota payload)
 *   - MurmurHash3_x86_32 seed 0 of the element bytes:
 *     cpp/include/cudf/hashing/detail/murmurhash3_x86_32.cuh:22-46
 *   - open-addressing multiset of {hash,row} built on the right table, probed by the left:
 *     cpp/src/join/hash_join/hash_join.cu:62-99,112-148; retrieve_impl.cuh:28-113
 *   - groupby SUM/COUNT by key: cpp/src/groupby/hash/compute_global_memory_aggs.cuh:123-157
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ------------------------------------------------------------------ murmur3 */
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static inline uint32_t fmix32(uint32_t h)
{
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}
static inline uint32_t mm_block(uint32_t h, uint32_t k)
{
  k *= 0xcc9e2d51u; k = rotl32(k, 15); k *= 0x1b873593u;
  h ^= k; h = rotl32(h, 13); h = h * 5u + 0xe6546b64u; return h;
}
uint32_t orc_murmur3_u32(uint32_t v, uint32_t seed) { return fmix32(mm_block(seed, v) ^ 4u); }
uint32_t orc_murmur3_u64(uint64_t v, uint32_t seed)
{
  uint32_t h = mm_block(seed, (uint32_t)v);
  h = mm_block(h, (uint32_t)(v >> 32));
  return fmix32(h ^ 8u);
}
void orc_murmur3_u32_array(const uint32_t* in, int64_t n, uint32_t seed, uint32_t* out)
{
  for (int64_t i = 0; i < n; ++i) out[i] = orc_murmur3_u32(in[i], seed);
}
void orc_murmur3_u64_array(const uint64_t* in, int64_t n, uint32_t seed, uint32_t* out)
{
  for (int64_t i = 0; i < n; ++i) out[i] = orc_murmur3_u64(in[i], seed);
}

/* ------------------------------------------------------------------ radix sort */
/* keys are "sortable bits" (unsigned order == cudf order); caller applies the sign flip. */
static inline uint64_t flip_i64(int64_t v, int descending)
{
  uint64_t u = (uint64_t)v ^ 0x8000000000000000ull;
  return descending ? ~u : u;
}

/* stable LSD radix sort of int64 keys, ascending or descending; tmp has n elements */
void orc_sort_i64(const int64_t* in, int64_t* out, int64_t* tmp, int64_t n, int descending)
{
  const int64_t* src = in;
  int64_t* bufs[2] = {tmp, out}; /* 8 passes: in->tmp->out->tmp->...->out */
  for (int pass = 0; pass < 8; ++pass) {
    int64_t* dst = bufs[pass & 1];
    int64_t hist[256];
    memset(hist, 0, sizeof(hist));
    const int shift = pass * 8;
    for (int64_t i = 0; i < n; ++i) hist[(flip_i64(src[i], descending) >> shift) & 0xff]++;
    int64_t sum = 0;
    for (int b = 0; b < 256; ++b) { int64_t c = hist[b]; hist[b] = sum; sum += c; }
    for (int64_t i = 0; i < n; ++i) dst[hist[(flip_i64(src[i], descending) >> shift) & 0xff]++] = src[i];
    src = dst;
  }
}

/* 32-bit integer keys: the same contract on four digit passes (cub::DeviceRadixSort::SortKeys over begin_bit = 0,
 * end_bit = 32 as cudf::sort calls it for an INT32 / UINT32 column: cpp/src/sort/sort_radix.cu:66-78; the reference's own
 * sort benchmark is typed on int32, cpp/benchmarks/sort/sort.cpp:51).  is_signed = 0: UINT32 (no sign flip). */
static inline uint32_t flip_32(uint32_t v, int is_signed, int descending)
{
  uint32_t u = is_signed ? (v ^ 0x80000000u) : v;
  return descending ? ~u : u;
}
void orc_sort_32(const uint32_t* in, uint32_t* out, uint32_t* tmp, int64_t n, int is_signed, int descending)
{
  const uint32_t* src = in;
  uint32_t* bufs[2] = {tmp, out}; /* 4 passes: in->tmp->out->tmp->out */
  for (int pass = 0; pass < 4; ++pass) {
    uint32_t* dst = bufs[pass & 1];
    int64_t hist[256];
    memset(hist, 0, sizeof(hist));
    const int shift = pass * 8;
    for (int64_t i = 0; i < n; ++i) hist[(flip_32(src[i], is_signed, descending) >> shift) & 0xff]++;
    int64_t sum = 0;
    for (int b = 0; b < 256; ++b) { int64_t c = hist[b]; hist[b] = sum; sum += c; }
    for (int64_t i = 0; i < n; ++i) dst[hist[(flip_32(src[i], is_signed, descending) >> shift) & 0xff]++] = src[i];
    src = dst;
  }
}

/* stable argsort (sorted_order): keys int64, out_idx int32; scratch: 2*n uint64 + n int32 */
void orc_sorted_order_i64(const int64_t* in, int32_t* out_idx, int64_t n, int descending)
{
  uint64_t* ka = (uint64_t*)malloc((size_t)n * 8);
  uint64_t* kb = (uint64_t*)malloc((size_t)n * 8);
  int32_t* ib  = (int32_t*)malloc((size_t)n * 4);
  int32_t* ia  = out_idx;
  for (int64_t i = 0; i < n; ++i) { ka[i] = flip_i64(in[i], descending); ia[i] = (int32_t)i; }
  for (int pass = 0; pass < 8; ++pass) {
    int64_t hist[256];
    memset(hist, 0, sizeof(hist));
    const int shift = pass * 8;
    for (int64_t i = 0; i < n; ++i) hist[(ka[i] >> shift) & 0xff]++;
    int64_t sum = 0;
    for (int b = 0; b < 256; ++b) { int64_t c = hist[b]; hist[b] = sum; sum += c; }
    for (int64_t i = 0; i < n; ++i) {
      int64_t d = hist[(ka[i] >> shift) & 0xff]++;
      kb[d] = ka[i]; ib[d] = ia[i];
    }
    uint64_t* tk = ka; ka = kb; kb = tk;
    int32_t* ti = ia; ia = ib; ib = ti;
  }
  /* an even number of swaps: ia is out_idx again, ib the scratch buffer */
  free(ka); free(kb); free(ib);
}

/* ------------------------------------------------------------------ hash join */
/* Build an open-addressing multiset of {hash32,row} over `right` (capacity = 2*nr rounded to
 * pow2, i.e. load factor <= 0.5 as CUCO_DESIRED_LOAD_FACTOR), probe with `left`.
 * Pass out_l == NULL to only count.  Returns the number of pairs. */
int64_t orc_inner_join_i64(const int64_t* left, int64_t nl, const int64_t* right, int64_t nr,
                           int32_t* out_l, int32_t* out_r, int64_t cap_out)
{
  if (nl == 0 || nr == 0) return 0;
  uint64_t cap = 1; while (cap < (uint64_t)nr * 2) cap <<= 1;
  const uint64_t mask = cap - 1;
  int32_t* slot_row = (int32_t*)malloc(cap * 4);
  uint32_t* slot_h  = (uint32_t*)malloc(cap * 4);
  for (uint64_t i = 0; i < cap; ++i) slot_row[i] = INT32_MIN; /* empty sentinel = JoinNoMatch */
  for (int64_t r = 0; r < nr; ++r) {
    uint32_t h = orc_murmur3_u64((uint64_t)right[r], 0);
    uint64_t s = h & mask;
    while (slot_row[s] != INT32_MIN) s = (s + 1) & mask;
    slot_row[s] = (int32_t)r; slot_h[s] = h;
  }
  int64_t count = 0;
  for (int64_t l = 0; l < nl; ++l) {
    uint32_t h = orc_murmur3_u64((uint64_t)left[l], 0);
    uint64_t s = h & mask;
    while (slot_row[s] != INT32_MIN) {
      if (slot_h[s] == h && right[slot_row[s]] == left[l]) {
        if (out_l && count < cap_out) { out_l[count] = (int32_t)l; out_r[count] = slot_row[s]; }
        ++count;
      }
      s = (s + 1) & mask;
    }
  }
  free(slot_row); free(slot_h);
  return count;
}

/* ------------------------------------------------------------------ groupby sum/count */
/* keys int32 in [0, ngroups) (dense: the BASELINE config-4 shape); exact double sum via
 * Neumaier compensation in long double -- well inside 1 ulp of the correctly rounded sum for the
 * group sizes used (checked against math.fsum in tests/test_oracle_golden.py). */
void orc_groupby_dense_sum_count(const int32_t* keys, const double* vals, int64_t n, int32_t ngroups,
                                 double* out_sum, int32_t* out_count)
{
  long double* s = (long double*)calloc((size_t)ngroups, sizeof(long double));
  long double* c = (long double*)calloc((size_t)ngroups, sizeof(long double));
  memset(out_count, 0, (size_t)ngroups * 4);
  for (int64_t i = 0; i < n; ++i) {
    const int32_t g = keys[i];
    const long double x = vals[i];
    const long double t = s[g] + x;
    if (fabsl(s[g]) >= fabsl(x)) c[g] += (s[g] - t) + x; else c[g] += (x - t) + s[g];
    s[g] = t;
    out_count[g]++;
  }
  for (int32_t g = 0; g < ngroups; ++g) out_sum[g] = (double)(s[g] + c[g]);
  free(s); free(c);
}

/* inclusive prefix sum, int64 wrap-around (cudf::scan keeps the input type:
 * cpp/src/reductions/scan/scan_inclusive.cu:72-74) */
void orc_inclusive_sum_i64(const int64_t* in, int64_t* out, int64_t n)
{
  uint64_t acc = 0;
  for (int64_t i = 0; i < n; ++i) { acc += (uint64_t)in[i]; out[i] = (int64_t)acc; }
}
