// gx_groups.hip -- group structure of rows that are (or have been brought) in key order: run heads, labels,
// offsets; and the per-group row operations built on it (segmented shift, null replacement, ranks).
//
// Replaces the sort-based helpers of the reference: compute_group_offsets / label_segments
// (cpp/src/groupby/sort/sort_helper.cu:151-214), segmented_shift (cpp/src/copying/segmented_shift.cu via
// groupby.cu:306-346), group_replace_nulls (cpp/src/groupby/sort/group_replace_nulls.cu), and the rank scans of
// cpp/src/sort/rank.cu:60-330.  All streaming, one pass each over the rows.
#include "gx_common.hpp"
#include "gx_scan.hpp"

namespace gx {
namespace grp {

// ---- run heads: heads[i] = 1 iff row i differs from row i - 1 in THIS column (rows taken through `order` when
// given); null == null, NaN == NaN, -0.0 == +0.0 (row equality: detail/row_operator/common_utils.cuh:215-220).
template <typename U, int KIND>
__global__ void __launch_bounds__(256) k_heads(const U* __restrict__ col, const uint32_t* __restrict__ valid,
                                               const int32_t* __restrict__ order, int64_t n, int combine,
                                               uint8_t* __restrict__ heads)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    uint8_t h = 1;
    if (i > 0) {
      const int64_t a = order ? order[i] : i, b = order ? order[i - 1] : i - 1;
      const bool va = !valid || bit_is_set(valid, a), vb = !valid || bit_is_set(valid, b);
      if (va != vb) h = 1;
      else if (!va) h = 0;
      else h = to_sortable<U, KIND>(col[a], U(0)) != to_sortable<U, KIND>(col[b], U(0)) ? 1 : 0;
    }
    heads[i] = combine ? (uint8_t)(heads[i] | h) : h;
  }
}

template <typename U, int KIND>
int heads_launch(const void* col, const uint32_t* valid, const int32_t* order, int64_t n, int combine, uint8_t* heads, hipStream_t s)
{
  int64_t blocks = div_up(n, 256 * 8);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL((k_heads<U, KIND>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const U*>(col), valid, order, n,
                     combine, heads);
  GX_LAUNCH_CHECK();
  return 0;
}

struct HeadLoader {
  const uint8_t* heads;
  __device__ __forceinline__ int32_t operator()(int64_t i) const { return heads[i]; }
};

// labels hold the inclusive count of heads; turn them into 0-based labels and scatter the group starts
__global__ void __launch_bounds__(256) k_offsets(const uint8_t* __restrict__ heads, int32_t* __restrict__ labels, int64_t n,
                                                 int32_t* __restrict__ offsets, long long* ngroups)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int32_t g = labels[i] - 1;
    labels[i]       = g;
    if (heads[i]) offsets[g] = (int32_t)i;
    if (i == n - 1) {
      offsets[g + 1] = (int32_t)n;
      *ngroups       = g + 1;
    }
  }
}

__global__ void __launch_bounds__(256) k_sizes(const int32_t* __restrict__ offsets, int64_t g, int32_t* __restrict__ sizes)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < g; i += stride) sizes[i] = offsets[i + 1] - offsets[i];
}

// ---- segmented shift: out[i] = in[i - off] when row i - off belongs to the same group, else the fill value
template <typename T>
__global__ void __launch_bounds__(256) k_seg_shift(const T* __restrict__ in, const uint32_t* __restrict__ valid,
                                                   const int32_t* __restrict__ labels, int64_t n, int64_t off, T fill,
                                                   int fill_valid, T* __restrict__ out, uint32_t* __restrict__ out_valid)
{
  // one wave writes one 64-bit chunk of validity: grid-stride over rows of 64
  const int64_t nrows64 = div_up(n, (int64_t)GX_WAVE);
  const int64_t wstride = (int64_t)gridDim.x * (256 / GX_WAVE);
  const unsigned lane   = lane_id();
  for (int64_t r = (int64_t)blockIdx.x * (256 / GX_WAVE) + threadIdx.x / GX_WAVE; r < nrows64; r += wstride) {
    const int64_t i = r * GX_WAVE + lane;
    bool ok         = false;
    if (i < n) {
      const int64_t j = i - off;
      T v             = fill;
      ok              = fill_valid != 0;
      if (j >= 0 && j < n && labels[j] == labels[i]) {
        v  = in[j];
        ok = !valid || bit_is_set(valid, j);
      }
      out[i] = v;
    }
    const uint64_t b = ballot(ok);
    if (out_valid && lane == 0) {
      out_valid[2 * r] = (uint32_t)b;
      if ((2 * r + 1) * 32 < n) out_valid[2 * r + 1] = (uint32_t)(b >> 32);
    }
  }
}

// ---- replace nulls inside groups (replace_policy::PRECEDING / FOLLOWING): source[i] = the nearest valid row of the
// same group before (after) row i, or -1.  A segmented max (min) scan over row indices.
struct SrcSeg {
  int32_t v;
  uint32_t f;
};
struct SrcOp {
  __device__ __forceinline__ SrcSeg operator()(SrcSeg a, SrcSeg b) const
  {
    return SrcSeg{b.f ? b.v : (b.v >= 0 ? b.v : a.v), a.f | b.f};
  }
};
// forward: element k is row k; backward: element k is row n - 1 - k and a run starts where the NEXT row is a head
struct SrcLoader {
  const uint32_t* valid;
  const uint8_t* heads;
  int64_t n;
  int backward;
  __device__ __forceinline__ SrcSeg operator()(int64_t k) const
  {
    const int64_t i = backward ? n - 1 - k : k;
    const bool ok   = !valid || bit_is_set(valid, i);
    const uint32_t f = backward ? ((i == n - 1 || heads[i + 1]) ? 1u : 0u) : heads[i];
    return SrcSeg{ok ? (int32_t)i : -1, f};
  }
};
struct SrcOut {  // what the scan writes: just the row index
  int32_t v;
  __device__ SrcOut() = default;
  __device__ explicit SrcOut(SrcSeg s) : v(s.v) {}
};

template <typename T>
__global__ void __launch_bounds__(256) k_fill_from(const T* __restrict__ in, const SrcOut* __restrict__ src, int64_t n,
                                                   int backward, T* __restrict__ out, uint32_t* __restrict__ out_valid)
{
  const int64_t nrows64 = div_up(n, (int64_t)GX_WAVE);
  const int64_t wstride = (int64_t)gridDim.x * (256 / GX_WAVE);
  const unsigned lane   = lane_id();
  for (int64_t r = (int64_t)blockIdx.x * (256 / GX_WAVE) + threadIdx.x / GX_WAVE; r < nrows64; r += wstride) {
    const int64_t i = r * GX_WAVE + lane;
    bool ok         = false;
    if (i < n) {
      const int32_t j = src[backward ? n - 1 - i : i].v;
      ok              = j >= 0;
      out[i]          = ok ? in[j] : in[i];
    }
    const uint64_t b = ballot(ok);
    if (lane == 0) {
      out_valid[2 * r] = (uint32_t)b;
      if ((2 * r + 1) * 32 < n) out_valid[2 * r + 1] = (uint32_t)(b >> 32);
    }
  }
}

// ---- rank: row order[i] (sorted position i, group g = labels[i]) gets
//   FIRST i + 1 | MIN offsets[g] + 1 | MAX offsets[g + 1] | DENSE g + 1 | AVERAGE (MIN + MAX) / 2
// as int32 (out_f64 == NULL) or double; `scale` > 0 turns the rank into a percentage: rank / scale, or
// (rank - 1) / (scale - 1) when `one_normalized` (cpp/src/sort/rank.cu:246-330).
__global__ void __launch_bounds__(256) k_rank(const int32_t* __restrict__ order, const int32_t* __restrict__ labels,
                                              const int32_t* __restrict__ offsets, int64_t n, int method, double scale,
                                              int one_normalized, int32_t* __restrict__ out_i32, double* __restrict__ out_f64)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int32_t g = method == 0 ? 0 : labels[i];
    double r;
    switch (method) {
      case 0: r = (double)(i + 1); break;
      case 2: r = (double)(offsets[g] + 1); break;
      case 3: r = (double)offsets[g + 1]; break;
      case 4: r = (double)(g + 1); break;
      default: r = ((double)(offsets[g] + 1) + (double)offsets[g + 1]) * 0.5; break;  // 1: AVERAGE
    }
    if (scale > 0.0) r = one_normalized ? (scale > 1.0 ? (r - 1.0) / (scale - 1.0) : 0.0) : r / scale;
    const int32_t row = order ? order[i] : (int32_t)i;
    if (out_f64) out_f64[row] = r; else out_i32[row] = (int32_t)r;
  }
}

// ---- segment id of every row for a segmented sort (cpp/src/sort/segmented_sort_impl.cuh:178-203): rows of segment
// [offsets[j], offsets[j+1]) get offsets[j+1]; rows before the first / from the last offset on get unique ascending ids
// (their own index / index + 1), so they keep their place.
__global__ void __launch_bounds__(256) k_segment_ids(const int32_t* __restrict__ offsets, int64_t noff, int64_t n,
                                                     int32_t* __restrict__ ids)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    int64_t lo = 0, hi = noff;  // first offset > i
    while (lo < hi) {
      const int64_t mid = (lo + hi) / 2;
      if ((int64_t)offsets[mid] <= i) lo = mid + 1; else hi = mid;
    }
    int32_t id;
    if (lo == 0) id = (int32_t)i;                  // before the first segment (or no offsets at all)
    else if (lo < noff) id = offsets[lo];          // inside segment lo - 1
    else id = (int32_t)i + 1;                      // from the last offset on
    ids[i] = id;
  }
}

static inline unsigned grid_for(int64_t n, int per = 256 * 4)
{
  int64_t b = div_up(n, per);
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace grp
}  // namespace gx

extern "C" {

int gx_group_heads(int dtype, const void* col, const uint32_t* valid, const int32_t* order, int64_t n, int combine,
                   uint8_t* heads, gx_stream_t s)
{
  using namespace gx;
  using namespace gx::grp;
  if (n < 0 || (n > 0 && (!col || !heads))) return GX_EINVAL;
  if (n == 0) return 0;
  switch (dtype) {
    case GX_INT8: return heads_launch<uint8_t, K_SIGNED>(col, valid, order, n, combine, heads, s);
    case GX_BOOL8:
    case GX_UINT8: return heads_launch<uint8_t, K_UNSIGNED>(col, valid, order, n, combine, heads, s);
    case GX_INT16: return heads_launch<uint16_t, K_SIGNED>(col, valid, order, n, combine, heads, s);
    case GX_UINT16: return heads_launch<uint16_t, K_UNSIGNED>(col, valid, order, n, combine, heads, s);
    case GX_INT32: return heads_launch<uint32_t, K_SIGNED>(col, valid, order, n, combine, heads, s);
    case GX_UINT32: return heads_launch<uint32_t, K_UNSIGNED>(col, valid, order, n, combine, heads, s);
    case GX_FLOAT32: return heads_launch<uint32_t, K_FLOAT>(col, valid, order, n, combine, heads, s);
    case GX_INT64: return heads_launch<uint64_t, K_SIGNED>(col, valid, order, n, combine, heads, s);
    case GX_UINT64: return heads_launch<uint64_t, K_UNSIGNED>(col, valid, order, n, combine, heads, s);
    case GX_FLOAT64: return heads_launch<uint64_t, K_FLOAT>(col, valid, order, n, combine, heads, s);
    default: return GX_EDTYPE;
  }
}

int gx_group_offsets(const uint8_t* heads, int64_t n, int32_t* labels, int32_t* offsets, int32_t* sizes, int64_t* ngroups_dev,
                     void* tmp, size_t* tmp_bytes, gx_stream_t s)
{
  using namespace gx;
  using namespace gx::grp;
  if (n < 0 || !tmp_bytes) return GX_EINVAL;
  Carver c(tmp);
  int32_t* partials = c.take<int32_t>(scan::partials_count(n));
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  if (!ngroups_dev) return GX_EINVAL;
  if (n == 0) {
    GX_HIP_TRY(hipMemsetAsync(ngroups_dev, 0, sizeof(int64_t), s));
    if (offsets) GX_HIP_TRY(hipMemsetAsync(offsets, 0, sizeof(int32_t), s));
    return 0;
  }
  if (!heads || !labels || !offsets) return GX_EINVAL;
  int rc = scan::device_scan<int32_t, int32_t>(HeadLoader{heads}, n, 0, SumOp(), true, labels, partials, s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_offsets, dim3(grid_for(n)), dim3(256), 0, s, heads, labels, n, offsets, reinterpret_cast<long long*>(ngroups_dev));
  if (sizes) {
    // the group count is on the device: cover the worst case (n groups), rows beyond it read offsets that k_offsets wrote
    // for smaller indices only -- so bound the launch by reading the count on the device
    hipLaunchKernelGGL(k_sizes, dim3(grid_for(n)), dim3(256), 0, s, offsets, n, sizes);
  }
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_segmented_shift(int elem_size, const void* in, const uint32_t* in_valid, const int32_t* labels, int64_t n, int64_t offset,
                       uint64_t fill_bits, int fill_valid, void* out, uint32_t* out_valid, gx_stream_t s)
{
  using namespace gx::grp;
  if (n < 0 || (n > 0 && (!in || !labels || !out))) return GX_EINVAL;
  if (n == 0) return 0;
  const unsigned grid = grid_for(n, 256 * 4);
#define GX_SHIFT(T) hipLaunchKernelGGL((k_seg_shift<T>), dim3(grid), dim3(256), 0, s, static_cast<const T*>(in), in_valid, labels, n, offset, (T)fill_bits, fill_valid, static_cast<T*>(out), out_valid); break
  switch (elem_size) {
    case 1: GX_SHIFT(uint8_t);
    case 2: GX_SHIFT(uint16_t);
    case 4: GX_SHIFT(uint32_t);
    case 8: GX_SHIFT(uint64_t);
    default: return GX_EDTYPE;
  }
#undef GX_SHIFT
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_segmented_fill_nulls(int elem_size, const void* in, const uint32_t* in_valid, const uint8_t* heads, int64_t n, int backward,
                            void* out, uint32_t* out_valid, void* tmp, size_t* tmp_bytes, gx_stream_t s)
{
  using namespace gx;
  using namespace gx::grp;
  if (n < 0 || !tmp_bytes) return GX_EINVAL;
  Carver c(tmp);
  SrcOut* src      = c.take<SrcOut>((size_t)n);
  SrcSeg* partials = c.take<SrcSeg>(scan::partials_count(n));
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  if (n == 0) return 0;
  if (!in || !heads || !out || !out_valid) return GX_EINVAL;
  int rc = scan::device_scan<SrcSeg, SrcOut>(SrcLoader{in_valid, heads, n, backward}, n, SrcSeg{-1, 0u}, SrcOp(), true, src, partials, s);
  if (rc) return rc;
  const unsigned grid = grid_for(n, 256 * 4);
#define GX_FILL(T) hipLaunchKernelGGL((k_fill_from<T>), dim3(grid), dim3(256), 0, s, static_cast<const T*>(in), src, n, backward, static_cast<T*>(out), out_valid); break
  switch (elem_size) {
    case 1: GX_FILL(uint8_t);
    case 2: GX_FILL(uint16_t);
    case 4: GX_FILL(uint32_t);
    case 8: GX_FILL(uint64_t);
    default: return GX_EDTYPE;
  }
#undef GX_FILL
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_rank_from_groups(const int32_t* order, const int32_t* labels, const int32_t* offsets, int64_t n, int method, double scale,
                        int one_normalized, int32_t* out_i32, double* out_f64, gx_stream_t s)
{
  using namespace gx::grp;
  if (n < 0 || method < 0 || method > 4 || (n > 0 && ((method != 0 && (!labels || !offsets)) || (!out_i32 && !out_f64)))) return GX_EINVAL;
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_rank, dim3(grid_for(n)), dim3(256), 0, s, order, labels, offsets, n, method, scale, one_normalized, out_i32,
                     out_f64);
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_segment_ids(const int32_t* offsets, int64_t num_offsets, int64_t num_rows, int32_t* ids, gx_stream_t s)
{
  using namespace gx::grp;
  if (num_offsets < 0 || num_rows < 0 || (num_rows > 0 && !ids) || (num_offsets > 0 && !offsets)) return GX_EINVAL;
  if (num_rows == 0) return 0;
  hipLaunchKernelGGL(k_segment_ids, dim3(grid_for(num_rows)), dim3(256), 0, s, offsets, num_offsets, num_rows, ids);
  GX_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
