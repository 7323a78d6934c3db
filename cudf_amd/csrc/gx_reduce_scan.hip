// gx_reduce_scan.hip -- cudf::reduce / cudf::scan / groupby::scan kernels' C ABI.
//
// reduce replaces cub::DeviceReduce::Reduce (cpp/include/cudf/reduction/detail/reduction.cuh:46-83),
// scan replaces thrust::inclusive_scan / exclusive_scan (cpp/src/reductions/scan/
// scan_inclusive.cu:76-89, scan_exclusive.cu), segmented scan replaces
// thrust::inclusive_scan_by_key (cpp/src/groupby/sort/group_scan_util.cuh:109-130).
//
// float SUM accumulates in double-double (error-free two_sum), so the rounded result is the
// correctly rounded exact sum in all but pathological cases and is bit-reproducible (fixed
// association order of gx_scan.hpp) -- north_star asks <= 1 ulp; integer SUM/PRODUCT wrap mod 2^64
// exactly like the reference's integer arithmetic.
#include <limits>
#include <type_traits>

#include "gx_common.hpp"
#include "gx_scan.hpp"

namespace gx {
namespace rs {

// ---------------------------------------------------------------- double-double accumulator
struct DD {
  double hi, lo;
  __host__ __device__ explicit operator double() const { return hi + lo; }
  __host__ __device__ explicit operator float() const { return (float)(hi + lo); }
};
struct DDSum {
  __device__ __forceinline__ DD operator()(DD a, DD b) const
  {
    // two_sum(a.hi, b.hi)
    const double s  = a.hi + b.hi;
    const double bb = s - a.hi;
    double e        = (a.hi - (s - bb)) + (b.hi - bb);
    e += a.lo + b.lo;
    // fast_two_sum(s, e)
    const double hi = s + e;
    const double lo = e - (hi - s);
    return DD{hi, lo};
  }
};

template <typename InT>
struct DDLoader {
  const InT* in;
  const uint32_t* valid;
  __device__ __forceinline__ DD operator()(int64_t i) const
  {
    if (valid && !bit_is_set(valid, i)) return DD{0.0, 0.0};
    return DD{(double)in[i], 0.0};
  }
};

// ---------------------------------------------------------------- segmented (by-key) element
template <typename T>
struct Seg {
  T v;
  uint32_t f;  // 1 = this element starts a new key run
  uint32_t pad;
  __host__ __device__ explicit operator double() const { return (double)v; }
  __host__ __device__ explicit operator float() const { return (float)v; }
  __host__ __device__ explicit operator long long() const { return (long long)v; }
  __host__ __device__ explicit operator long() const { return (long)v; }
  __host__ __device__ explicit operator int() const { return (int)v; }
  __host__ __device__ explicit operator short() const { return (short)v; }
  __host__ __device__ explicit operator signed char() const { return (signed char)v; }
  __host__ __device__ explicit operator unsigned long long() const { return (unsigned long long)v; }
  __host__ __device__ explicit operator unsigned long() const { return (unsigned long)v; }
  __host__ __device__ explicit operator unsigned int() const { return (unsigned int)v; }
  __host__ __device__ explicit operator unsigned short() const { return (unsigned short)v; }
  __host__ __device__ explicit operator unsigned char() const { return (unsigned char)v; }
};
template <typename T, typename Op>
struct SegOp {
  Op op;
  __device__ __forceinline__ Seg<T> operator()(Seg<T> a, Seg<T> b) const
  {
    return Seg<T>{b.f ? b.v : op(a.v, b.v), a.f | b.f, 0u};
  }
};
template <typename InT, typename AccT>
struct SegLoader {
  const InT* in;
  const uint32_t* valid;
  const uint8_t* heads;
  AccT identity;
  __device__ __forceinline__ Seg<AccT> operator()(int64_t i) const
  {
    const AccT v = (valid && !bit_is_set(valid, i)) ? identity : static_cast<AccT>(in[i]);
    return Seg<AccT>{v, heads[i], 0u};
  }
};

template <typename U, int KIND>
__global__ void __launch_bounds__(256) k_run_heads(const U* __restrict__ keys, int64_t n, uint8_t* __restrict__ heads)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    heads[i] = (i == 0 || to_sortable<U, KIND>(keys[i], U(0)) != to_sortable<U, KIND>(keys[i - 1], U(0))) ? 1 : 0;
  }
}

template <typename U, int KIND>
int heads_launch(const void* keys, int64_t n, uint8_t* heads, hipStream_t s)
{
  int64_t blocks = div_up(n, 256 * 8);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((k_run_heads<U, KIND>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const U*>(keys), n,
                     heads);
  GX_LAUNCH_CHECK();
  return 0;
}

static int heads_dispatch(int key_dtype, const void* keys, int64_t n, uint8_t* heads, hipStream_t s)
{
  switch (key_dtype) {
    case GX_INT8: return heads_launch<uint8_t, K_SIGNED>(keys, n, heads, s);
    case GX_BOOL8:
    case GX_UINT8: return heads_launch<uint8_t, K_UNSIGNED>(keys, n, heads, s);
    case GX_INT16: return heads_launch<uint16_t, K_SIGNED>(keys, n, heads, s);
    case GX_UINT16: return heads_launch<uint16_t, K_UNSIGNED>(keys, n, heads, s);
    case GX_INT32: return heads_launch<uint32_t, K_SIGNED>(keys, n, heads, s);
    case GX_UINT32: return heads_launch<uint32_t, K_UNSIGNED>(keys, n, heads, s);
    case GX_FLOAT32: return heads_launch<uint32_t, K_FLOAT>(keys, n, heads, s);
    case GX_INT64: return heads_launch<uint64_t, K_SIGNED>(keys, n, heads, s);
    case GX_UINT64: return heads_launch<uint64_t, K_UNSIGNED>(keys, n, heads, s);
    case GX_FLOAT64: return heads_launch<uint64_t, K_FLOAT>(keys, n, heads, s);
    default: return GX_EDTYPE;
  }
}

// ---------------------------------------------------------------- identities
template <typename T>
struct Limits;
#define GX_LIMITS(T, LO, HI)                                   \
  template <>                                                  \
  struct Limits<T> {                                           \
    static constexpr T lowest() { return LO; }                 \
    static constexpr T highest() { return HI; }                \
  };
GX_LIMITS(int64_t, INT64_MIN, INT64_MAX)
GX_LIMITS(uint64_t, 0ull, UINT64_MAX)
GX_LIMITS(double, -__builtin_huge_val(), __builtin_huge_val())
#undef GX_LIMITS

// result conversion: partials[nc] (AccT) -> *out in out_dtype
__global__ void k_set_i64(int64_t* p, int64_t v) { *p = v; }

template <typename AccT>
__device__ __forceinline__ void store_as(AccT r, int out_dtype, void* out)
{
  switch (out_dtype) {
    case GX_INT8: *static_cast<int8_t*>(out) = static_cast<int8_t>(r); break;
    case GX_BOOL8:
    case GX_UINT8: *static_cast<uint8_t*>(out) = static_cast<uint8_t>(r); break;
    case GX_INT16: *static_cast<int16_t*>(out) = static_cast<int16_t>(r); break;
    case GX_UINT16: *static_cast<uint16_t*>(out) = static_cast<uint16_t>(r); break;
    case GX_INT32: *static_cast<int32_t*>(out) = static_cast<int32_t>(r); break;
    case GX_UINT32: *static_cast<uint32_t*>(out) = static_cast<uint32_t>(r); break;
    case GX_INT64: *static_cast<int64_t*>(out) = static_cast<int64_t>(r); break;
    case GX_UINT64: *static_cast<uint64_t*>(out) = static_cast<uint64_t>(r); break;
    case GX_FLOAT32: *static_cast<float*>(out) = static_cast<float>(r); break;
    case GX_FLOAT64: *static_cast<double*>(out) = static_cast<double>(r); break;
    default: break;
  }
}
template <typename AccT>
__global__ void k_store_result(const AccT* res, int out_dtype, void* out)
{
  store_as<AccT>(*res, out_dtype, out);
}
template <>
__global__ void k_store_result<DD>(const DD* res, int out_dtype, void* out)
{
  store_as<double>(res->hi + res->lo, out_dtype, out);
}

template <typename InT, typename AccT, typename Op>
int reduce_typed(const void* in, const uint32_t* valid, int64_t n, AccT identity, Op op, int out_dtype, void* out,
                 void* tmp, size_t* tmp_bytes, hipStream_t s)
{
  Carver c(tmp);
  AccT* partials = c.take<AccT>(scan::partials_count(n));
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  scan::PlainLoader<InT, AccT> ld{static_cast<const InT*>(in), valid, identity};
  int rc = scan::device_reduce<AccT>(ld, n, identity, op, partials, s);
  if (rc) return rc;
  hipLaunchKernelGGL((k_store_result<AccT>), dim3(1), dim3(1), 0, s, partials + scan::reduce_blocks(n), out_dtype, out);
  GX_LAUNCH_CHECK();
  return 0;
}

// number of valid elements that are not zero (a NaN is not zero): what cudf::reduce ANY / ALL need -- the reference reduces
// static_cast<bool>(x) with max / min (src/reductions/any.cu:79-95, all.cu) -- as ONE streaming pass whatever the input type
template <typename InT>
struct NonZeroLoader {
  const InT* in;
  const uint32_t* valid;
  __device__ __forceinline__ uint64_t operator()(int64_t i) const
  {
    if (valid && !bit_is_set(valid, i)) return 0;
    return in[i] != InT(0) ? 1ull : 0ull;
  }
};
template <typename InT>
int reduce_nonzero(const void* in, const uint32_t* valid, int64_t n, int out_dtype, void* out, void* tmp, size_t* tmp_bytes, hipStream_t s)
{
  Carver c(tmp);
  uint64_t* partials = c.take<uint64_t>(scan::partials_count(n));
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  NonZeroLoader<InT> ld{static_cast<const InT*>(in), valid};
  int rc = scan::device_reduce<uint64_t>(ld, n, uint64_t(0), SumOp(), partials, s);
  if (rc) return rc;
  hipLaunchKernelGGL((k_store_result<uint64_t>), dim3(1), dim3(1), 0, s, partials + scan::reduce_blocks(n), out_dtype, out);
  GX_LAUNCH_CHECK();
  return 0;
}

template <typename InT>
int reduce_dd(const void* in, const uint32_t* valid, int64_t n, int out_dtype, void* out, void* tmp,
              size_t* tmp_bytes, hipStream_t s)
{
  Carver c(tmp);
  DD* partials = c.take<DD>(scan::partials_count(n));
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  DDLoader<InT> ld{static_cast<const InT*>(in), valid};
  int rc = scan::device_reduce<DD>(ld, n, DD{0.0, 0.0}, DDSum(), partials, s);
  if (rc) return rc;
  hipLaunchKernelGGL((k_store_result<DD>), dim3(1), dim3(1), 0, s, partials + scan::reduce_blocks(n), out_dtype, out);
  GX_LAUNCH_CHECK();
  return 0;
}

// integers: SUM/PRODUCT in uint64 (wrap == two's complement), MIN/MAX in int64 or uint64
template <typename InT, bool SIGNED>
int reduce_int(const void* in, const uint32_t* valid, int64_t n, int op, int out_dtype, void* out, void* tmp,
               size_t* tmp_bytes, hipStream_t s)
{
  using W = typename std::conditional<SIGNED, int64_t, uint64_t>::type;
  switch (op) {
    case GX_OP_SUM: return reduce_typed<InT, uint64_t>(in, valid, n, uint64_t(0), SumOp(), out_dtype, out, tmp, tmp_bytes, s);
    case GX_OP_PRODUCT: return reduce_typed<InT, uint64_t>(in, valid, n, uint64_t(1), ProdOp(), out_dtype, out, tmp, tmp_bytes, s);
    case GX_OP_MIN: return reduce_typed<InT, W>(in, valid, n, Limits<W>::highest(), MinOp(), out_dtype, out, tmp, tmp_bytes, s);
    case GX_OP_MAX: return reduce_typed<InT, W>(in, valid, n, Limits<W>::lowest(), MaxOp(), out_dtype, out, tmp, tmp_bytes, s);
    case GX_OP_COUNT_NONZERO: return reduce_nonzero<InT>(in, valid, n, out_dtype, out, tmp, tmp_bytes, s);
    default: return GX_EINVAL;
  }
}
template <typename InT>
int reduce_float(const void* in, const uint32_t* valid, int64_t n, int op, int out_dtype, void* out, void* tmp,
                 size_t* tmp_bytes, hipStream_t s)
{
  switch (op) {
    case GX_OP_SUM: return reduce_dd<InT>(in, valid, n, out_dtype, out, tmp, tmp_bytes, s);
    case GX_OP_PRODUCT: return reduce_typed<InT, double>(in, valid, n, 1.0, ProdOp(), out_dtype, out, tmp, tmp_bytes, s);
    case GX_OP_MIN: return reduce_typed<InT, double>(in, valid, n, Limits<double>::highest(), MinOp(), out_dtype, out, tmp, tmp_bytes, s);
    case GX_OP_MAX: return reduce_typed<InT, double>(in, valid, n, Limits<double>::lowest(), MaxOp(), out_dtype, out, tmp, tmp_bytes, s);
    case GX_OP_COUNT_NONZERO: return reduce_nonzero<InT>(in, valid, n, out_dtype, out, tmp, tmp_bytes, s);
    default: return GX_EINVAL;
  }
}

// ---------------------------------------------------------------- column scan
template <typename InT, typename AccT, typename Op>
int scan_typed(const void* in, const uint32_t* valid, int64_t n, AccT identity, Op op, int inclusive, void* out,
               void* tmp, size_t* tmp_bytes, hipStream_t s)
{
  Carver c(tmp);
  AccT* partials = c.take<AccT>(scan::partials_count(n));
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  scan::PlainLoader<InT, AccT> ld{static_cast<const InT*>(in), valid, identity};
  return scan::device_scan<AccT, InT>(ld, n, identity, op, inclusive != 0, static_cast<InT*>(out), partials, s);
}
template <typename InT>
int scan_dd(const void* in, const uint32_t* valid, int64_t n, int inclusive, void* out, void* tmp, size_t* tmp_bytes,
            hipStream_t s)
{
  Carver c(tmp);
  DD* partials = c.take<DD>(scan::partials_count(n));
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  DDLoader<InT> ld{static_cast<const InT*>(in), valid};
  return scan::device_scan<DD, InT>(ld, n, DD{0.0, 0.0}, DDSum(), inclusive != 0, static_cast<InT*>(out), partials, s);
}
// exact commutative operators: the single-pass look-back scan (16 B/row)
template <typename InT, typename AccT, typename Op>
int scan_lookback(const void* in, const uint32_t* valid, int64_t n, AccT identity, Op op, int inclusive, void* out, void* tmp,
                  size_t* tmp_bytes, hipStream_t s)
{
  Carver c(tmp);
  char* scratch = c.take<char>(scan::lookback_bytes(n));
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  scan::PlainLoader<InT, AccT> ld{static_cast<const InT*>(in), valid, identity};
  return scan::device_scan_lookback<AccT, InT>(ld, n, identity, op, inclusive != 0, static_cast<InT*>(out), scratch, s);
}
template <typename InT, bool SIGNED>
int scan_int(const void* in, const uint32_t* valid, int64_t n, int op, int inclusive, void* out, void* tmp,
             size_t* tmp_bytes, hipStream_t s)
{
  using W = typename std::conditional<SIGNED, int64_t, uint64_t>::type;
  switch (op) {
    case GX_OP_SUM: return scan_lookback<InT, uint64_t>(in, valid, n, uint64_t(0), SumOp(), inclusive, out, tmp, tmp_bytes, s);
    case GX_OP_PRODUCT: return scan_lookback<InT, uint64_t>(in, valid, n, uint64_t(1), ProdOp(), inclusive, out, tmp, tmp_bytes, s);
    case GX_OP_MIN: return scan_lookback<InT, W>(in, valid, n, (W)std::numeric_limits<InT>::max(), MinOp(), inclusive, out, tmp, tmp_bytes, s);
    case GX_OP_MAX: return scan_lookback<InT, W>(in, valid, n, (W)std::numeric_limits<InT>::lowest(), MaxOp(), inclusive, out, tmp, tmp_bytes, s);
    default: return GX_EINVAL;
  }
}
template <typename InT>
int scan_float(const void* in, const uint32_t* valid, int64_t n, int op, int inclusive, void* out, void* tmp,
               size_t* tmp_bytes, hipStream_t s)
{
  switch (op) {
    case GX_OP_SUM: return scan_dd<InT>(in, valid, n, inclusive, out, tmp, tmp_bytes, s);
    case GX_OP_PRODUCT: return scan_typed<InT, double>(in, valid, n, 1.0, ProdOp(), inclusive, out, tmp, tmp_bytes, s);
    case GX_OP_MIN: return scan_lookback<InT, double>(in, valid, n, Limits<double>::highest(), MinOp(), inclusive, out, tmp, tmp_bytes, s);
    case GX_OP_MAX: return scan_lookback<InT, double>(in, valid, n, Limits<double>::lowest(), MaxOp(), inclusive, out, tmp, tmp_bytes, s);
    default: return GX_EINVAL;
  }
}

// ---------------------------------------------------------------- segmented scan
// COUNT scans (cpp/src/groupby/sort/group_count_scan.cu:24-62): the scanned value is 1 per row (COUNT_ALL) or the
// row's validity (COUNT_VALID); the values themselves are never read
struct SegCountLoader {
  const uint32_t* valid;  // NULL: every row counts
  const uint8_t* heads;
  __device__ __forceinline__ Seg<int32_t> operator()(int64_t i) const
  {
    return Seg<int32_t>{(valid && !bit_is_set(valid, i)) ? 0 : 1, heads[i], 0u};
  }
};
static int seg_count(const uint32_t* valid, const uint8_t* heads, int64_t n, void* out, void* partials, hipStream_t s)
{
  SegCountLoader ld{valid, heads};
  return scan::device_scan<Seg<int32_t>, int32_t>(ld, n, Seg<int32_t>{0, 0u, 0u}, SegOp<int32_t, SumOp>{SumOp()}, true,
                                                  static_cast<int32_t*>(out), static_cast<Seg<int32_t>*>(partials), s);
}

template <typename InT, typename AccT, typename OutT, typename Op>
int seg_typed(const void* vals, const uint32_t* valid, const uint8_t* heads, int64_t n, AccT identity, Op op,
              void* out, Seg<AccT>* partials, hipStream_t s)
{
  SegLoader<InT, AccT> ld{static_cast<const InT*>(vals), valid, heads, identity};
  return scan::device_scan<Seg<AccT>, OutT>(ld, n, Seg<AccT>{identity, 0u, 0u}, SegOp<AccT, Op>{op}, true,
                                            static_cast<OutT*>(out), partials, s);
}

// integer SUM accumulates and returns int64 (cpp/src/groupby/sort/group_scan_util.cuh:85-95);
// MIN/MAX return the input type
template <typename InT, bool SIGNED>
int seg_int(const void* vals, const uint32_t* valid, const uint8_t* heads, int64_t n, int op, void* out,
            void* partials, hipStream_t s)
{
  using W = typename std::conditional<SIGNED, int64_t, uint64_t>::type;
  switch (op) {
    case GX_OP_SUM: return seg_typed<InT, int64_t, int64_t>(vals, valid, heads, n, int64_t(0), SumOp(), out, static_cast<Seg<int64_t>*>(partials), s);
    case GX_OP_MIN: return seg_typed<InT, W, InT>(vals, valid, heads, n, (W)std::numeric_limits<InT>::max(), MinOp(), out, static_cast<Seg<W>*>(partials), s);
    case GX_OP_MAX: return seg_typed<InT, W, InT>(vals, valid, heads, n, (W)std::numeric_limits<InT>::lowest(), MaxOp(), out, static_cast<Seg<W>*>(partials), s);
    default: return GX_EINVAL;
  }
}
template <typename InT>
int seg_float(const void* vals, const uint32_t* valid, const uint8_t* heads, int64_t n, int op, void* out,
              void* partials, hipStream_t s)
{
  switch (op) {
    case GX_OP_SUM: return seg_typed<InT, double, InT>(vals, valid, heads, n, 0.0, SumOp(), out, static_cast<Seg<double>*>(partials), s);
    case GX_OP_MIN: return seg_typed<InT, double, InT>(vals, valid, heads, n, Limits<double>::highest(), MinOp(), out, static_cast<Seg<double>*>(partials), s);
    case GX_OP_MAX: return seg_typed<InT, double, InT>(vals, valid, heads, n, Limits<double>::lowest(), MaxOp(), out, static_cast<Seg<double>*>(partials), s);
    default: return GX_EINVAL;
  }
}

// ---------------------------------------------------------------- segmented reduce (sort-path groupby)
// One result per key run instead of one per row: the same reduce-then-scan over (value, head flag) elements,
// carrying the number of valid values as a third field, with the last pass storing only where a run ENDS
// (at out[label]).  Replaces thrust::reduce_by_key of the reference's sort-based aggregations
// (cpp/src/groupby/sort/group_single_pass_reduction_util.cuh:133-200, group_count.cu:25-89).  Fixed
// association order; float SUM in double-double, so results are bit-reproducible and within 1 ulp.
template <typename T>
struct SegC {
  T v;
  uint32_t f;  // 1 = this element starts a new key run
  uint32_t c;  // valid values folded into v
};
template <typename T, typename Op>
struct SegCOp {
  Op op;
  __device__ __forceinline__ SegC<T> operator()(SegC<T> a, SegC<T> b) const
  {
    return SegC<T>{b.f ? b.v : op(a.v, b.v), a.f | b.f, b.f ? b.c : a.c + b.c};
  }
};
template <typename AccT>
struct AccFrom {
  template <typename X>
  static __device__ __forceinline__ AccT of(X x) { return static_cast<AccT>(x); }
};
template <>
struct AccFrom<DD> {
  template <typename X>
  static __device__ __forceinline__ DD of(X x) { return DD{(double)x, 0.0}; }
};
template <typename InT, typename AccT>
struct SegCLoader {
  const InT* in;
  const uint32_t* valid;
  const uint8_t* heads;
  AccT identity;
  __device__ __forceinline__ SegC<AccT> operator()(int64_t i) const
  {
    const bool ok = !valid || bit_is_set(valid, i);
    return SegC<AccT>{(ok && in) ? AccFrom<AccT>::of(in[i]) : identity, heads[i], ok ? 1u : 0u};
  }
};

template <typename AccT, typename OutT, typename Op, typename Loader>
__global__ void __launch_bounds__(scan::SCAN_BT) k_chunk_segreduce(Loader load, int64_t n, SegC<AccT> identity, SegCOp<AccT, Op> op,
                                                                   const SegC<AccT>* partials, const uint8_t* __restrict__ heads,
                                                                   const int32_t* __restrict__ labels, OutT* __restrict__ out,
                                                                   int32_t* __restrict__ out_cnt)
{
  using namespace scan;
  constexpr int NWV = SCAN_BT / GX_WAVE;
  __shared__ SegC<AccT> s_w[NWV];
  const unsigned l    = lane_id();
  const unsigned w    = threadIdx.x / GX_WAVE;
  const int64_t chunk = blockIdx.x;
  const int64_t base  = chunk * SCAN_CHUNK + (int64_t)w * (GX_WAVE * SCAN_IPT) + l;
  SegC<AccT> inc[SCAN_IPT];
  SegC<AccT> carry = identity;
#pragma unroll
  for (int k = 0; k < SCAN_IPT; ++k) {
    const int64_t i    = base + (int64_t)k * GX_WAVE;
    const SegC<AccT> v = (i < n) ? load(i) : identity;
    inc[k]             = wave_inclusive_scan(v, op);
    carry              = op(carry, shfl(inc[k], GX_WAVE - 1));
  }
  if (l == 0) s_w[w] = carry;
  __syncthreads();
  SegC<AccT> run = partials[chunk];
  for (unsigned k = 0; k < w; ++k) run = op(run, s_w[k]);
#pragma unroll
  for (int k = 0; k < SCAN_IPT; ++k) {
    const int64_t i = base + (int64_t)k * GX_WAVE;
    if (i < n && (i + 1 == n || heads[i + 1])) {  // last row of its run
      const SegC<AccT> r = op(run, inc[k]);
      const int32_t g    = labels[i];
      if (out) out[g] = static_cast<OutT>(r.v);
      if (out_cnt) out_cnt[g] = (int32_t)r.c;
    }
    run = op(run, shfl(inc[k], GX_WAVE - 1));
  }
}

template <typename InT, typename AccT, typename OutT, typename Op>
int segreduce_typed(const void* vals, const uint32_t* valid, const uint8_t* heads, const int32_t* labels, int64_t n,
                    AccT identity, Op op, void* out, int32_t* out_cnt, void* partials_raw, hipStream_t s)
{
  using E = SegC<AccT>;
  E* partials = static_cast<E*>(partials_raw);
  SegCLoader<InT, AccT> ld{static_cast<const InT*>(vals), valid, heads, identity};
  const E ident{identity, 0u, 0u};
  const SegCOp<AccT, Op> sop{op};
  const int64_t nc = scan::num_chunks(n);
  hipLaunchKernelGGL((scan::k_chunk_reduce<E, SegCOp<AccT, Op>, SegCLoader<InT, AccT>>), dim3((unsigned)nc), dim3(scan::SCAN_BT), 0, s,
                     ld, n, ident, sop, partials, (const int*)nullptr);
  hipLaunchKernelGGL((scan::k_partials_scan<E, SegCOp<AccT, Op>>), dim3(1), dim3(1024), 0, s, partials, nc, ident, sop,
                     (const int*)nullptr);
  hipLaunchKernelGGL((k_chunk_segreduce<AccT, OutT, Op, SegCLoader<InT, AccT>>), dim3((unsigned)nc), dim3(scan::SCAN_BT), 0, s, ld, n,
                     ident, sop, partials, heads, labels, static_cast<OutT*>(out), out_cnt);
  GX_LAUNCH_CHECK();
  return 0;
}

// integers: SUM / PRODUCT accumulate and return int64 (aggregation.hpp:949-970); MIN / MAX return the input type
template <typename InT, bool SIGNED>
int segreduce_int(const void* vals, const uint32_t* valid, const uint8_t* heads, const int32_t* labels, int64_t n, int op,
                  void* out, int32_t* out_cnt, void* partials, hipStream_t s)
{
  using W = typename std::conditional<SIGNED, int64_t, uint64_t>::type;
  switch (op) {
    case GX_OP_COUNT_VALID:
    case GX_OP_SUM: return segreduce_typed<InT, int64_t, int64_t>(vals, valid, heads, labels, n, int64_t(0), SumOp(), out, out_cnt, partials, s);
    case GX_OP_PRODUCT: return segreduce_typed<InT, uint64_t, int64_t>(vals, valid, heads, labels, n, uint64_t(1), ProdOp(), out, out_cnt, partials, s);
    case GX_OP_MIN: return segreduce_typed<InT, W, InT>(vals, valid, heads, labels, n, (W)std::numeric_limits<InT>::max(), MinOp(), out, out_cnt, partials, s);
    case GX_OP_MAX: return segreduce_typed<InT, W, InT>(vals, valid, heads, labels, n, (W)std::numeric_limits<InT>::lowest(), MaxOp(), out, out_cnt, partials, s);
    default: return GX_EINVAL;
  }
}
template <typename InT>
int segreduce_float(const void* vals, const uint32_t* valid, const uint8_t* heads, const int32_t* labels, int64_t n, int op,
                    void* out, int32_t* out_cnt, void* partials, hipStream_t s)
{
  switch (op) {
    case GX_OP_COUNT_VALID:
    case GX_OP_SUM: return segreduce_typed<InT, DD, InT>(vals, valid, heads, labels, n, DD{0.0, 0.0}, DDSum(), out, out_cnt, partials, s);
    case GX_OP_PRODUCT: return segreduce_typed<InT, double, InT>(vals, valid, heads, labels, n, 1.0, ProdOp(), out, out_cnt, partials, s);
    case GX_OP_MIN: return segreduce_typed<InT, double, InT>(vals, valid, heads, labels, n, Limits<double>::highest(), MinOp(), out, out_cnt, partials, s);
    case GX_OP_MAX: return segreduce_typed<InT, double, InT>(vals, valid, heads, labels, n, Limits<double>::lowest(), MaxOp(), out, out_cnt, partials, s);
    default: return GX_EINVAL;
  }
}

}  // namespace rs
}  // namespace gx

extern "C" {

int gx_reduce(int in_dtype, const void* in, const uint32_t* valid, int64_t n, int op, int out_dtype, void* out_dev,
              int64_t* valid_count_dev, void* tmp, size_t* tmp_bytes, gx_stream_t s)
{
  using namespace gx::rs;
  if (n < 0 || !tmp_bytes) return GX_EINVAL;
  if (tmp && (!out_dev || (n > 0 && !in))) return GX_EINVAL;
  if (tmp && valid_count_dev) {
    if (valid) {
      int rc = gx_bitmask_count(valid, 0, n, valid_count_dev, s);
      if (rc) return rc;
    } else {
      hipLaunchKernelGGL(gx::rs::k_set_i64, dim3(1), dim3(1), 0, s, valid_count_dev, n);
    }
  }
  switch (in_dtype) {
    case GX_INT8: return reduce_int<int8_t, true>(in, valid, n, op, out_dtype, out_dev, tmp, tmp_bytes, s);
    case GX_INT16: return reduce_int<int16_t, true>(in, valid, n, op, out_dtype, out_dev, tmp, tmp_bytes, s);
    case GX_INT32: return reduce_int<int32_t, true>(in, valid, n, op, out_dtype, out_dev, tmp, tmp_bytes, s);
    case GX_INT64: return reduce_int<int64_t, true>(in, valid, n, op, out_dtype, out_dev, tmp, tmp_bytes, s);
    case GX_BOOL8:
    case GX_UINT8: return reduce_int<uint8_t, false>(in, valid, n, op, out_dtype, out_dev, tmp, tmp_bytes, s);
    case GX_UINT16: return reduce_int<uint16_t, false>(in, valid, n, op, out_dtype, out_dev, tmp, tmp_bytes, s);
    case GX_UINT32: return reduce_int<uint32_t, false>(in, valid, n, op, out_dtype, out_dev, tmp, tmp_bytes, s);
    case GX_UINT64: return reduce_int<uint64_t, false>(in, valid, n, op, out_dtype, out_dev, tmp, tmp_bytes, s);
    case GX_FLOAT32: return reduce_float<float>(in, valid, n, op, out_dtype, out_dev, tmp, tmp_bytes, s);
    case GX_FLOAT64: return reduce_float<double>(in, valid, n, op, out_dtype, out_dev, tmp, tmp_bytes, s);
    default: return GX_EDTYPE;
  }
}

int gx_scan(int dtype, const void* in, const uint32_t* valid, int64_t n, int op, int inclusive, void* out, void* tmp,
            size_t* tmp_bytes, gx_stream_t s)
{
  using namespace gx::rs;
  if (n < 0 || !tmp_bytes) return GX_EINVAL;
  if (tmp && n > 0 && (!in || !out)) return GX_EINVAL;
  switch (dtype) {
    case GX_INT8: return scan_int<int8_t, true>(in, valid, n, op, inclusive, out, tmp, tmp_bytes, s);
    case GX_INT16: return scan_int<int16_t, true>(in, valid, n, op, inclusive, out, tmp, tmp_bytes, s);
    case GX_INT32: return scan_int<int32_t, true>(in, valid, n, op, inclusive, out, tmp, tmp_bytes, s);
    case GX_INT64: return scan_int<int64_t, true>(in, valid, n, op, inclusive, out, tmp, tmp_bytes, s);
    case GX_BOOL8:
    case GX_UINT8: return scan_int<uint8_t, false>(in, valid, n, op, inclusive, out, tmp, tmp_bytes, s);
    case GX_UINT16: return scan_int<uint16_t, false>(in, valid, n, op, inclusive, out, tmp, tmp_bytes, s);
    case GX_UINT32: return scan_int<uint32_t, false>(in, valid, n, op, inclusive, out, tmp, tmp_bytes, s);
    case GX_UINT64: return scan_int<uint64_t, false>(in, valid, n, op, inclusive, out, tmp, tmp_bytes, s);
    case GX_FLOAT32: return scan_float<float>(in, valid, n, op, inclusive, out, tmp, tmp_bytes, s);
    case GX_FLOAT64: return scan_float<double>(in, valid, n, op, inclusive, out, tmp, tmp_bytes, s);
    default: return GX_EDTYPE;
  }
}

int gx_segmented_scan(int key_dtype, const void* sorted_keys, int val_dtype, const void* vals,
                      const uint32_t* vals_valid, int64_t n, int op, void* out, void* tmp, size_t* tmp_bytes,
                      gx_stream_t s)
{
  using namespace gx::rs;
  if (n < 0 || !tmp_bytes) return GX_EINVAL;
  gx::Carver c(tmp);
  uint8_t* heads = c.take<uint8_t>((size_t)n);
  char* partials = c.take<char>(gx::scan::partials_count(n) * sizeof(Seg<double>));
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  if (n == 0) return 0;
  const bool counting = op == GX_OP_COUNT_VALID || op == GX_OP_COUNT_ALL;
  if (!sorted_keys || (!vals && !counting) || !out) return GX_EINVAL;
  int rc = heads_dispatch(key_dtype, sorted_keys, n, heads, s);
  if (rc) return rc;
  if (counting) return seg_count(op == GX_OP_COUNT_ALL ? nullptr : vals_valid, heads, n, out, partials, s);
  switch (val_dtype) {
    case GX_INT8: return seg_int<int8_t, true>(vals, vals_valid, heads, n, op, out, partials, s);
    case GX_INT16: return seg_int<int16_t, true>(vals, vals_valid, heads, n, op, out, partials, s);
    case GX_INT32: return seg_int<int32_t, true>(vals, vals_valid, heads, n, op, out, partials, s);
    case GX_INT64: return seg_int<int64_t, true>(vals, vals_valid, heads, n, op, out, partials, s);
    case GX_BOOL8:
    case GX_UINT8: return seg_int<uint8_t, false>(vals, vals_valid, heads, n, op, out, partials, s);
    case GX_UINT16: return seg_int<uint16_t, false>(vals, vals_valid, heads, n, op, out, partials, s);
    case GX_UINT32: return seg_int<uint32_t, false>(vals, vals_valid, heads, n, op, out, partials, s);
    case GX_UINT64: return seg_int<uint64_t, false>(vals, vals_valid, heads, n, op, out, partials, s);
    case GX_FLOAT32: return seg_float<float>(vals, vals_valid, heads, n, op, out, partials, s);
    case GX_FLOAT64: return seg_float<double>(vals, vals_valid, heads, n, op, out, partials, s);
    default: return GX_EDTYPE;
  }
}

/* see gx.h */
int gx_segmented_reduce(int val_dtype, const void* vals, const uint32_t* vals_valid, const uint8_t* heads, const int32_t* labels,
                        int64_t n, int op, void* out, int32_t* out_count_valid, void* tmp, size_t* tmp_bytes, gx_stream_t s)
{
  using namespace gx::rs;
  if (n < 0 || !tmp_bytes) return GX_EINVAL;
  gx::Carver c(tmp);
  char* partials = c.take<char>(gx::scan::partials_count(n) * sizeof(SegC<DD>));
  if (!tmp) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  if (n == 0) return 0;
  if (!heads || !labels || (!out && !out_count_valid)) return GX_EINVAL;
  if (op == GX_OP_COUNT_VALID) {
    out = nullptr;  // counts only; the values are not read
    if (!out_count_valid) return GX_EINVAL;
    return segreduce_int<int32_t, true>(nullptr, vals_valid, heads, labels, n, op, nullptr, out_count_valid, partials, s);
  }
  if (!vals) return GX_EINVAL;
  switch (val_dtype) {
    case GX_INT8: return segreduce_int<int8_t, true>(vals, vals_valid, heads, labels, n, op, out, out_count_valid, partials, s);
    case GX_INT16: return segreduce_int<int16_t, true>(vals, vals_valid, heads, labels, n, op, out, out_count_valid, partials, s);
    case GX_INT32: return segreduce_int<int32_t, true>(vals, vals_valid, heads, labels, n, op, out, out_count_valid, partials, s);
    case GX_INT64: return segreduce_int<int64_t, true>(vals, vals_valid, heads, labels, n, op, out, out_count_valid, partials, s);
    case GX_BOOL8:
    case GX_UINT8: return segreduce_int<uint8_t, false>(vals, vals_valid, heads, labels, n, op, out, out_count_valid, partials, s);
    case GX_UINT16: return segreduce_int<uint16_t, false>(vals, vals_valid, heads, labels, n, op, out, out_count_valid, partials, s);
    case GX_UINT32: return segreduce_int<uint32_t, false>(vals, vals_valid, heads, labels, n, op, out, out_count_valid, partials, s);
    case GX_UINT64: return segreduce_int<uint64_t, false>(vals, vals_valid, heads, labels, n, op, out, out_count_valid, partials, s);
    case GX_FLOAT32: return segreduce_float<float>(vals, vals_valid, heads, labels, n, op, out, out_count_valid, partials, s);
    case GX_FLOAT64: return segreduce_float<double>(vals, vals_valid, heads, labels, n, op, out, out_count_valid, partials, s);
    default: return GX_EDTYPE;
  }
}

}  // extern "C"
