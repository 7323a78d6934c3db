// gx_rank.hip -- row-key encoding for multi-column join / groupby keys.
//
// The reference hashes and compares whole rows inside its hash tables
// (cpp/include/cudf/detail/row_operator/primitive_row_operators.cuh:95-163,207-274; equality.cuh), one
// dereference per column per probe.  Here the hash kernels keep ONE fixed-width key per row and the
// row is encoded into it first:
//   * gx_pack_keys   -- key columns whose widths sum to <= 8 bytes are concatenated into a uint64
//                       (exact; floats are normalised so that -0.0 == +0.0 and NaN == NaN, the row
//                       comparator's equality: detail/row_operator/common_utils.cuh:215-220);
//   * gx_dense_rank  -- any single column -> dense int32 ids (equal values share an id, null == null
//                       gets its own id, ids ascend with the value, nulls last), plus the first row
//                       of every id.  Wider rows are encoded by ranking each column and packing
//                       (id, id) pairs again -- cudf's own key_remapping (include/cudf/join/
//                       key_remapping.hpp) serves the same purpose.  Built on the radix sort:
//                       sorted_order -> adjacent-difference flags -> scan -> scatter.
#include "gx_common.hpp"

#include <cstring>

namespace gx {
namespace rank {

struct PackCols {
  const void* p[8];
  int size[8];
  int is_float[8];
  int shift[8];
  int ncols;
};

__device__ __forceinline__ uint64_t load_bits(const void* p, int size, int64_t i)
{
  switch (size) {
    case 1: return static_cast<const uint8_t*>(p)[i];
    case 2: return static_cast<const uint16_t*>(p)[i];
    case 4: return static_cast<const uint32_t*>(p)[i];
    default: return static_cast<const uint64_t*>(p)[i];
  }
}

// equality classes of the row comparator for floats: one zero, one NaN
__device__ __forceinline__ uint64_t normalise_float(uint64_t b, int size)
{
  if (size == 4) {
    const uint32_t x = (uint32_t)b;
    if ((x & 0x7FFFFFFFu) == 0) return 0;
    if ((x & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC00000u;
    return x;
  }
  if ((b & 0x7FFFFFFFFFFFFFFFull) == 0) return 0;
  if ((b & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull) return 0x7FF8000000000000ull;
  return b;
}

__global__ void __launch_bounds__(256) k_pack(PackCols c, int64_t n, uint64_t* __restrict__ out)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    uint64_t v = 0;
    for (int k = 0; k < c.ncols; ++k) {
      uint64_t b = load_bits(c.p[k], c.size[k], i);
      if (c.is_float[k]) b = normalise_float(b, c.size[k]);
      v |= b << c.shift[k];
    }
    out[i] = v;
  }
}

// flags[i] = 1 when sorted row i starts a new equality class
template <typename T>
__global__ void __launch_bounds__(256) k_differ(const T* __restrict__ keys, const uint32_t* __restrict__ valid,
                                                const int32_t* __restrict__ order, int64_t n, int is_float,
                                                int32_t* __restrict__ flags)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    int32_t f = 0;
    if (i > 0) {
      const int32_t a = order[i - 1], b = order[i];
      const bool va = !valid || bit_is_set(valid, a), vb = !valid || bit_is_set(valid, b);
      if (va != vb) {
        f = 1;
      } else if (va) {
        uint64_t x = keys[a], y = keys[b];
        if (is_float) {
          x = normalise_float(x, (int)sizeof(T));
          y = normalise_float(y, (int)sizeof(T));
        }
        f = x != y;
      }
    }
    flags[i] = f;
  }
}

// ids (inclusive scan of the flags, in sorted order) back to row order; first row of every id
__global__ void __launch_bounds__(256) k_rank_scatter(const int32_t* __restrict__ order, const int32_t* __restrict__ ids_sorted,
                                                      int64_t n, int32_t* __restrict__ out_ids, int32_t* __restrict__ out_rep,
                                                      int64_t* out_ngroups)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int32_t id = ids_sorted[i], row = order[i];
    out_ids[row]     = id;
    if (out_rep && (i == 0 || ids_sorted[i - 1] != id)) out_rep[id] = row;  // stable sort: the smallest row of the class
    if (i == n - 1 && out_ngroups) *out_ngroups = (int64_t)id + 1;
  }
}

// data[i] = value where row i is null (cudf::replace_nulls with a scalar, in place)
template <typename T>
__global__ void __launch_bounds__(256) k_fill_nulls(T* __restrict__ data, const uint32_t* __restrict__ valid, int64_t n, T value)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    if (!bit_is_set(valid, i)) data[i] = value;
}

static inline unsigned grid_for(int64_t n)
{
  int64_t b = div_up(n, (int64_t)256 * 4);
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (unsigned)b;
}


// inverse of k_pack: column k of row i = bits [shift_k, shift_k + 8 size_k) of packed[i]
struct UnpackCols {
  void* p[8];
  int size[8];
  int shift[8];
  int ncols;
};
__global__ void __launch_bounds__(256) k_unpack(UnpackCols c, int64_t n, const uint64_t* __restrict__ packed)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const uint64_t v = packed[i];
    for (int k = 0; k < c.ncols; ++k) {
      const uint64_t b = v >> c.shift[k];
      switch (c.size[k]) {
        case 1: static_cast<uint8_t*>(c.p[k])[i] = (uint8_t)b; break;
        case 2: static_cast<uint16_t*>(c.p[k])[i] = (uint16_t)b; break;
        case 4: static_cast<uint32_t*>(c.p[k])[i] = (uint32_t)b; break;
        default: static_cast<uint64_t*>(c.p[k])[i] = b; break;
      }
    }
  }
}

// ---- hashed row keys: rows wider than 8 bytes get a 64-bit hash as their join / groupby key, and the result is
// VERIFIED against the real columns afterwards (k_rows_mismatch): equal rows always hash equal, so a verified result
// is exact, and the caller falls back to the dense-rank encoding in the (2^-64 per pair) case of a collision.
// One pass over the key columns instead of one radix sort per column; the reference hashes the row once as well
// (cpp/include/cudf/detail/row_operator/primitive_row_operators.cuh:247-268) and compares on every probe (:95-163).
__device__ __forceinline__ uint64_t fmix64(uint64_t x)
{
  x ^= x >> 33;
  x *= 0xFF51AFD7ED558CCDull;
  x ^= x >> 33;
  x *= 0xC4CEB9FE1A85EC53ull;
  x ^= x >> 33;
  return x;
}
__global__ void __launch_bounds__(256) k_hash_rows(PackCols c, int64_t n, uint64_t seed, uint64_t* __restrict__ out)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    uint64_t h = seed;
    for (int k = 0; k < c.ncols; ++k) {
      uint64_t b = load_bits(c.p[k], c.size[k], i);
      if (c.is_float[k]) b = normalise_float(b, c.size[k]);
      h = fmix64(h + 0x9E3779B97F4A7C15ull + b) ^ (h << 1 | h >> 63);
    }
    out[i] = h;
  }
}
// pairs (lidx[i], ridx[i]) -- NULL = row i itself; a negative index = no row, skipped -- whose rows differ in a column
__global__ void __launch_bounds__(256) k_rows_mismatch(PackCols l, PackCols r, const int32_t* __restrict__ lidx,
                                                       const int32_t* __restrict__ ridx, int64_t npairs, unsigned long long* count)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  unsigned long long bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npairs; i += stride) {
    const int64_t a = lidx ? (int64_t)lidx[i] : i, b = ridx ? (int64_t)ridx[i] : i;
    if (a < 0 || b < 0) continue;
    bool diff = false;
    for (int k = 0; k < l.ncols; ++k) {
      uint64_t x = load_bits(l.p[k], l.size[k], a), y = load_bits(r.p[k], r.size[k], b);
      if (l.is_float[k]) {
        x = normalise_float(x, l.size[k]);
        y = normalise_float(y, l.size[k]);
      }
      diff = diff || x != y;
    }
    bad += diff ? 1ull : 0ull;
  }
  bad = wave_reduce(bad, SumOp());
  if (lane_id() == 0 && bad) atomicAdd(count, bad);
}
static inline int fill_cols(PackCols& c, int ncols, const void* const* cols, const int* dtypes, int64_t n)
{
  std::memset(&c, 0, sizeof(c));
  for (int k = 0; k < ncols; ++k) {
    const int sz = gx_dtype_size(dtypes[k]);
    if (sz <= 0) return GX_EDTYPE;
    if (n > 0 && !cols[k]) return GX_EINVAL;
    c.p[k]        = cols[k];
    c.size[k]     = sz;
    c.is_float[k] = dtypes[k] == GX_FLOAT32 || dtypes[k] == GX_FLOAT64;
  }
  c.ncols = ncols;
  return 0;
}

}  // namespace rank
}  // namespace gx

extern "C" {

int gx_pack_keys(int ncols, const void* const* cols, const int* dtypes, int64_t n, uint64_t* out, gx_stream_t s)
{
  if (ncols < 1 || ncols > 8 || !cols || !dtypes || n < 0 || (n > 0 && !out)) return GX_EINVAL;
  gx::rank::PackCols c;
  std::memset(&c, 0, sizeof(c));
  int bits = 0;
  for (int k = ncols - 1; k >= 0; --k) {  // first column in the most significant position
    const int sz = gx_dtype_size(dtypes[k]);
    if (sz <= 0) return GX_EDTYPE;
    if (n > 0 && !cols[k]) return GX_EINVAL;
    c.p[k]        = cols[k];
    c.size[k]     = sz;
    c.is_float[k] = dtypes[k] == GX_FLOAT32 || dtypes[k] == GX_FLOAT64;
    c.shift[k]    = bits;
    bits += sz * 8;
  }
  if (bits > 64) return GX_EINVAL;
  c.ncols = ncols;
  if (n == 0) return 0;
  hipLaunchKernelGGL(gx::rank::k_pack, dim3(gx::rank::grid_for(n)), dim3(256), 0, s, c, n, out);
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_unpack_keys(int ncols, void* const* out_cols, const int* dtypes, int64_t n, const uint64_t* packed, gx_stream_t s)
{
  if (ncols < 1 || ncols > 8 || !out_cols || !dtypes || n < 0 || (n > 0 && !packed)) return GX_EINVAL;
  gx::rank::UnpackCols c;
  std::memset(&c, 0, sizeof(c));
  int bits = 0;
  for (int k = ncols - 1; k >= 0; --k) {  // the layout of gx_pack_keys: first column most significant
    const int sz = gx_dtype_size(dtypes[k]);
    if (sz <= 0) return GX_EDTYPE;
    if (n > 0 && !out_cols[k]) return GX_EINVAL;
    c.p[k]     = out_cols[k];
    c.size[k]  = sz;
    c.shift[k] = bits;
    bits += sz * 8;
  }
  if (bits > 64) return GX_EINVAL;
  c.ncols = ncols;
  if (n == 0) return 0;
  hipLaunchKernelGGL(gx::rank::k_unpack, dim3(gx::rank::grid_for(n)), dim3(256), 0, s, c, n, packed);
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_hash_rows64(int ncols, const void* const* cols, const int* dtypes, int64_t n, uint64_t seed, uint64_t* out, gx_stream_t s)
{
  if (ncols < 1 || ncols > 8 || !cols || !dtypes || n < 0 || (n > 0 && !out)) return GX_EINVAL;
  gx::rank::PackCols c;
  if (int rc = gx::rank::fill_cols(c, ncols, cols, dtypes, n)) return rc;
  if (n == 0) return 0;
  hipLaunchKernelGGL(gx::rank::k_hash_rows, dim3(gx::rank::grid_for(n)), dim3(256), 0, s, c, n, seed, out);
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_rows_mismatch_count(int ncols, const void* const* lcols, const void* const* rcols, const int* dtypes, const int32_t* lidx,
                           const int32_t* ridx, int64_t npairs, int64_t* mismatch_dev, gx_stream_t s)
{
  if (ncols < 1 || ncols > 8 || !lcols || !rcols || !dtypes || npairs < 0 || !mismatch_dev) return GX_EINVAL;
  gx::rank::PackCols l, r;
  if (int rc = gx::rank::fill_cols(l, ncols, lcols, dtypes, npairs)) return rc;
  if (int rc = gx::rank::fill_cols(r, ncols, rcols, dtypes, npairs)) return rc;
  GX_HIP_TRY(hipMemsetAsync(mismatch_dev, 0, sizeof(int64_t), s));
  if (npairs == 0) return 0;
  hipLaunchKernelGGL(gx::rank::k_rows_mismatch, dim3(gx::rank::grid_for(npairs)), dim3(256), 0, s, l, r, lidx, ridx, npairs,
                     reinterpret_cast<unsigned long long*>(mismatch_dev));
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_fill_nulls(int elem_size, void* data, const uint32_t* valid, int64_t n, uint64_t value_bits, gx_stream_t s)
{
  if (n < 0 || (n > 0 && !data)) return GX_EINVAL;
  if (n == 0 || !valid) return 0;
  const unsigned g = gx::rank::grid_for(n);
  switch (elem_size) {
    case 1: hipLaunchKernelGGL((gx::rank::k_fill_nulls<uint8_t>), dim3(g), dim3(256), 0, s, static_cast<uint8_t*>(data), valid, n, (uint8_t)value_bits); break;
    case 2: hipLaunchKernelGGL((gx::rank::k_fill_nulls<uint16_t>), dim3(g), dim3(256), 0, s, static_cast<uint16_t*>(data), valid, n, (uint16_t)value_bits); break;
    case 4: hipLaunchKernelGGL((gx::rank::k_fill_nulls<uint32_t>), dim3(g), dim3(256), 0, s, static_cast<uint32_t*>(data), valid, n, (uint32_t)value_bits); break;
    case 8: hipLaunchKernelGGL((gx::rank::k_fill_nulls<uint64_t>), dim3(g), dim3(256), 0, s, static_cast<uint64_t*>(data), valid, n, (uint64_t)value_bits); break;
    default: return GX_EDTYPE;
  }
  GX_LAUNCH_CHECK();
  return 0;
}

int gx_dense_rank(int dtype, const void* keys, const uint32_t* valid, int64_t n, int64_t null_count, int32_t* out_ids,
                  int32_t* out_rep, int64_t* out_ngroups_dev, void* tmp, size_t* tmp_bytes, gx_stream_t s)
{
  if (!tmp_bytes || n < 0 || null_count < 0 || null_count > n) return GX_EINVAL;
  const int sz = gx_dtype_size(dtype);
  if (sz <= 0) return GX_EDTYPE;
  size_t sort_bytes = 0, scan_bytes = 0;
  int rc = gx_sorted_order(dtype, nullptr, nullptr, n, 0, 0, 0, nullptr, nullptr, &sort_bytes, s);
  if (rc) return rc;
  if (valid && null_count > 0) {
    size_t nb = 0;
    rc        = gx_sorted_order(dtype, nullptr, valid, n, null_count, 0, 0, nullptr, nullptr, &nb, s);
    if (rc) return rc;
    if (nb > sort_bytes) sort_bytes = nb;
  }
  rc = gx_scan(GX_INT32, nullptr, nullptr, n, GX_OP_SUM, 1, nullptr, nullptr, &scan_bytes, s);
  if (rc) return rc;
  gx::Carver c(tmp);
  int32_t* order = c.take<int32_t>((size_t)n + 1);
  int32_t* flags = c.take<int32_t>((size_t)n + 1);
  char* sub      = c.take<char>(sort_bytes > scan_bytes ? sort_bytes : scan_bytes);
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  if (!out_ids && n > 0) return GX_EINVAL;
  if (n == 0) {
    if (out_ngroups_dev) GX_HIP_TRY(hipMemsetAsync(out_ngroups_dev, 0, sizeof(int64_t), s));
    return 0;
  }
  if (!keys) return GX_EINVAL;
  const uint32_t* v = (valid && null_count > 0) ? valid : nullptr;
  rc = gx_sorted_order(dtype, keys, v, n, v ? null_count : 0, 0, /*nulls_before=*/0, order, sub, &sort_bytes, s);
  if (rc) return rc;
  const int is_float = dtype == GX_FLOAT32 || dtype == GX_FLOAT64;
  const unsigned g   = gx::rank::grid_for(n);
  switch (sz) {
    case 1: hipLaunchKernelGGL((gx::rank::k_differ<uint8_t>), dim3(g), dim3(256), 0, s, static_cast<const uint8_t*>(keys), v, order, n, 0, flags); break;
    case 2: hipLaunchKernelGGL((gx::rank::k_differ<uint16_t>), dim3(g), dim3(256), 0, s, static_cast<const uint16_t*>(keys), v, order, n, 0, flags); break;
    case 4: hipLaunchKernelGGL((gx::rank::k_differ<uint32_t>), dim3(g), dim3(256), 0, s, static_cast<const uint32_t*>(keys), v, order, n, is_float, flags); break;
    default: hipLaunchKernelGGL((gx::rank::k_differ<uint64_t>), dim3(g), dim3(256), 0, s, static_cast<const uint64_t*>(keys), v, order, n, is_float, flags); break;
  }
  rc = gx_scan(GX_INT32, flags, nullptr, n, GX_OP_SUM, 1, flags, sub, &scan_bytes, s);
  if (rc) return rc;
  hipLaunchKernelGGL(gx::rank::k_rank_scatter, dim3(g), dim3(256), 0, s, order, flags, n, out_ids, out_rep, out_ngroups_dev);
  GX_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
