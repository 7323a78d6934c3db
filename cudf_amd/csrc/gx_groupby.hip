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
  uint64_t h = gb_hash(s, log2cap);
  for (uint64_t probes = 0; probes < cap; ++probes) {
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

static inline uint32_t log2_cap(int64_t max_groups)
{
  uint32_t lg = 6;
  while ((1ull << lg) < (unsigned long long)(max_groups < 1 ? 1 : max_groups) * 2ull) ++lg;
  return lg;
}

template <typename K, typename V, bool IS_FLOAT>
int groupby_impl(const void* keys, const uint32_t* kvalid, const void* vals, const uint32_t* vvalid, int64_t n,
                 int64_t max_groups, void* out_keys, void* out_sum, int32_t* out_cv, int32_t* out_ca,
                 int64_t* ngroups, void* tmp, size_t* tmp_bytes, hipStream_t s)
{
  const uint32_t lg  = log2_cap(max_groups);
  const uint64_t cap = 1ull << lg;
  Carver c(tmp);
  GbState* st               = c.take<GbState>(1);
  unsigned long long* table = c.take<unsigned long long>(cap);
  double* sum               = c.take<double>(cap + 1);
  double* comp              = c.take<double>(cap + 1);
  uint32_t* cv              = c.take<uint32_t>(cap + 1);
  uint32_t* ca              = c.take<uint32_t>(cap + 1);
  uint32_t* pos             = c.take<uint32_t>(cap + 1);
  uint32_t* partials        = c.take<uint32_t>(scan::partials_count(cap + 1));
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  // one memset covers state, table and accumulators (they are contiguous up to `pos`)
  GX_HIP_TRY(hipMemsetAsync(tmp, 0, (size_t)(reinterpret_cast<char*>(pos) - static_cast<char*>(tmp)), s));
  if (n > 0) {
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

}  // extern "C"
