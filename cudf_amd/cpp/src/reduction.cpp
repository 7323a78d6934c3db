// cudf::reduce / cudf::scan and the hashing entry points over the C ABI.
// reference: cpp/src/reductions/reductions.cpp:484-507, simple.cuh:47-85, scan/scan.cpp:13-54,
// scan/scan_inclusive.cu:36-240; cpp/src/hash/murmurhash3_x86_32.cu; cpp/src/partitioning/partitioning.cu:568-660.
#include "common.hpp"

#include <cudf/column/column_factories.hpp>
#include <cudf/copying.hpp>
#include <cudf/hashing.hpp>
#include <cudf/reduction.hpp>

#include <functional>
#include <optional>
#include <type_traits>

namespace cudf {
namespace {

double read_init_as_double(scalar const& s, rmm::cuda_stream_view stream);

int gx_op_of(aggregation::Kind k)
{
  switch (k) {
    case aggregation::SUM: return GX_OP_SUM;
    case aggregation::PRODUCT: return GX_OP_PRODUCT;
    case aggregation::MIN: return GX_OP_MIN;
    case aggregation::MAX: return GX_OP_MAX;
    default: CUDF_FAIL("aggregation kind not implemented on this path (SUM, PRODUCT, MIN, MAX, MEAN, COUNT_VALID, COUNT_ALL, ANY, ALL are)");
  }
}

// a valid numeric scalar of a runtime type holding a host value (count.cpp:20-34: static_cast<T>(count))
template <typename V>
std::unique_ptr<scalar> host_scalar(data_type type, V v, bool valid, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  switch (type.id()) {
    case type_id::INT8: return std::make_unique<numeric_scalar<int8_t>>(static_cast<int8_t>(v), valid, stream, mr);
    case type_id::INT16: return std::make_unique<numeric_scalar<int16_t>>(static_cast<int16_t>(v), valid, stream, mr);
    case type_id::INT32: return std::make_unique<numeric_scalar<int32_t>>(static_cast<int32_t>(v), valid, stream, mr);
    case type_id::INT64: return std::make_unique<numeric_scalar<int64_t>>(static_cast<int64_t>(v), valid, stream, mr);
    case type_id::UINT8: return std::make_unique<numeric_scalar<uint8_t>>(static_cast<uint8_t>(v), valid, stream, mr);
    case type_id::UINT16: return std::make_unique<numeric_scalar<uint16_t>>(static_cast<uint16_t>(v), valid, stream, mr);
    case type_id::UINT32: return std::make_unique<numeric_scalar<uint32_t>>(static_cast<uint32_t>(v), valid, stream, mr);
    case type_id::UINT64: return std::make_unique<numeric_scalar<uint64_t>>(static_cast<uint64_t>(v), valid, stream, mr);
    case type_id::FLOAT32: return std::make_unique<numeric_scalar<float>>(static_cast<float>(v), valid, stream, mr);
    case type_id::FLOAT64: return std::make_unique<numeric_scalar<double>>(static_cast<double>(v), valid, stream, mr);
    case type_id::BOOL8: return std::make_unique<numeric_scalar<bool>>(v != V(0), valid, stream, mr);
    default: CUDF_FAIL("reduce: unsupported output type");
  }
}

bool is_arithmetic_id(type_id id)
{
  switch (id) {
    case type_id::INT8: case type_id::INT16: case type_id::INT32: case type_id::INT64:
    case type_id::UINT8: case type_id::UINT16: case type_id::UINT32: case type_id::UINT64:
    case type_id::FLOAT32: case type_id::FLOAT64: case type_id::BOOL8: return true;
    default: return false;
  }
}

// one gx_reduce into a device scalar of `acc` type, read back (reduce ends in a host-visible scalar anyway: the reference's
// compound and bool reductions return through a device_scalar too, reduction.cuh:63-81)
template <typename T>
T reduce_to_host(column_view const& col, int op, int acc_dtype, rmm::cuda_stream_view stream)
{
  rmm::device_buffer holder;
  auto const* mask = col.has_nulls() ? detail::rebased_mask(col, holder, stream) : nullptr;
  rmm::device_buffer value{16, stream};
  detail::run_with_scratch(
    [&](void* t, std::size_t* b) {
      return gx_reduce(detail::gx_type(col.type()), detail::row0(col), mask, col.size(), op, acc_dtype, value.data(), nullptr, t, b,
                       detail::gxs(stream));
    },
    "reduce", stream);
  T h{};
  CUDF_CUDA_TRY(hipMemcpyAsync(&h, value.data(), sizeof(T), hipMemcpyDeviceToHost, stream.value()));
  CUDF_CUDA_TRY(hipStreamSynchronize(stream.value()));
  return h;
}

// COUNT_VALID / COUNT_ALL (reductions/count.cpp:37-46): size - null_count (EXCLUDE) or size, as the output type; any numeric type but
// bool; always valid, empty and all-null columns included (reductions.cpp:252-275: reduce_no_data == reduce)
std::unique_ptr<scalar> reduce_count(column_view const& col, bool include_nulls, data_type output_type, rmm::cuda_stream_view stream,
                                     rmm::device_async_resource_ref mr)
{
  if (!is_arithmetic_id(output_type.id()) || output_type.id() == type_id::BOOL8)
    throw std::invalid_argument{"COUNT is not supported for boolean or non-numeric types"};
  auto const count = col.size() - (include_nulls ? 0 : col.null_count());
  return host_scalar<int64_t>(output_type, count, true, stream, mr);
}

// MEAN (reductions/mean.cu, compound.cuh:41-84, reduction_operators.cuh:256-275): sum of the valid elements / valid count in the
// (floating) output type; no valid row -> an invalid scalar.  The sum is the double-double SUM of gx_reduce: within an ulp of the exact
// quotient, where the reference's order of additions is cub's
std::unique_ptr<scalar> reduce_mean(column_view const& col, data_type output_type, rmm::cuda_stream_view stream,
                                    rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(is_arithmetic_id(col.type().id()), "Reduction operator `mean` `var` `std` not supported for this type");
  CUDF_EXPECTS(output_type.id() == type_id::FLOAT32 || output_type.id() == type_id::FLOAT64, "Unsupported output data type");
  auto const valid_count = col.size() - col.null_count();
  if (valid_count == 0) return host_scalar<double>(output_type, 0.0, false, stream, mr);
  double const sum = reduce_to_host<double>(col, GX_OP_SUM, GX_FLOAT64, stream);
  return host_scalar<double>(output_type, sum / static_cast<double>(valid_count), true, stream, mr);
}

// ANY / ALL (reductions/any.cu:79-95, all.cu; simple.cuh:47-85, 238-259): max / min over static_cast<bool>(x), nulls skipped, the
// initial value cast to bool and folded in; BOOL8 output only.  No valid row: any = false, all = true, both VALID
// (reductions.cpp:163-186) -- with or without an initial value, which reduce_no_data never looks at
std::unique_ptr<scalar> reduce_any_all(column_view const& col, bool is_any, data_type output_type,
                                       std::optional<std::reference_wrapper<scalar const>> init, rmm::cuda_stream_view stream,
                                       rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(output_type == data_type{type_id::BOOL8},
               is_any ? "any() operation can be applied with output type `bool8` only" : "all() operation can be applied with output type `BOOL8` only");
  auto const valid_count = col.size() - col.null_count();
  if (valid_count == 0) return std::make_unique<numeric_scalar<bool>>(!is_any, true, stream, mr);
  CUDF_EXPECTS(is_arithmetic_id(col.type().id()), "Reduction operator not supported for this type");
  auto const nonzero = reduce_to_host<int64_t>(col, GX_OP_COUNT_NONZERO, GX_INT64, stream);
  bool result        = is_any ? nonzero > 0 : nonzero == static_cast<int64_t>(valid_count);
  bool valid         = true;
  if (init.has_value()) {
    if (!init.value().get().is_valid(stream)) valid = false;
    else {
      bool const iv = read_init_as_double(init.value().get(), stream) != 0.0;  // (NaN != 0: true, as static_cast<bool>)
      result        = is_any ? (result || iv) : (result && iv);
    }
  }
  return std::make_unique<numeric_scalar<bool>>(result, valid, stream, mr);
}

template <typename T>
std::unique_ptr<scalar> make_result(void const* dev_value, bool valid, rmm::cuda_stream_view stream,
                                    rmm::device_async_resource_ref mr)
{
  auto s = std::make_unique<numeric_scalar<T>>(detail::uninitialized_value_t{}, valid, stream, mr);
  CUDF_CUDA_TRY(hipMemcpyAsync(s->data(), dev_value, sizeof(T), hipMemcpyDeviceToDevice, stream.value()));
  return s;
}

}  // namespace

std::unique_ptr<scalar> reduce(column_view const& col, reduce_aggregation const& agg, data_type output_type,
                               rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  switch (agg.kind) {  // reductions.cpp:484-507 dispatches on (type, kind); these four are not cub reductions
    case aggregation::COUNT_VALID: return reduce_count(col, false, output_type, stream, mr);
    case aggregation::COUNT_ALL: return reduce_count(col, true, output_type, stream, mr);
    case aggregation::MEAN: return reduce_mean(col, output_type, stream, mr);
    case aggregation::ANY: return reduce_any_all(col, true, output_type, std::nullopt, stream, mr);
    case aggregation::ALL: return reduce_any_all(col, false, output_type, std::nullopt, stream, mr);
    default: break;
  }
  int const op = gx_op_of(agg.kind);
  CUDF_EXPECTS(is_fixed_width(col.type()), "reduce: only fixed-width columns are supported on this path", cudf::data_type_error);
  if (op == GX_OP_MIN || op == GX_OP_MAX)
    CUDF_EXPECTS(output_type == col.type(), "min/max reduction output type must match the input type", cudf::data_type_error);
  rmm::device_buffer holder;
  auto const* mask = col.has_nulls() ? detail::rebased_mask(col, holder, stream) : nullptr;
  rmm::device_buffer value{16, stream}, cnt{sizeof(int64_t), stream};
  detail::run_with_scratch(
    [&](void* t, std::size_t* b) {
      return gx_reduce(detail::gx_type(col.type()), detail::row0(col), mask, col.size(), op, detail::gx_type(output_type),
                       value.data(), static_cast<int64_t*>(cnt.data()), t, b, detail::gxs(stream));
    },
    "reduce", stream);
  bool const valid = detail::read_i64(static_cast<int64_t const*>(cnt.data()), stream) > 0;
  switch (output_type.id()) {
    case type_id::INT8: return make_result<int8_t>(value.data(), valid, stream, mr);
    case type_id::INT16: return make_result<int16_t>(value.data(), valid, stream, mr);
    case type_id::INT32: return make_result<int32_t>(value.data(), valid, stream, mr);
    case type_id::INT64: return make_result<int64_t>(value.data(), valid, stream, mr);
    case type_id::UINT8: return make_result<uint8_t>(value.data(), valid, stream, mr);
    case type_id::UINT16: return make_result<uint16_t>(value.data(), valid, stream, mr);
    case type_id::UINT32: return make_result<uint32_t>(value.data(), valid, stream, mr);
    case type_id::UINT64: return make_result<uint64_t>(value.data(), valid, stream, mr);
    case type_id::FLOAT32: return make_result<float>(value.data(), valid, stream, mr);
    case type_id::FLOAT64: return make_result<double>(value.data(), valid, stream, mr);
    default: CUDF_FAIL("reduce: unsupported output type");
  }
}

namespace {

// op(init, r) in the output type T, as the reference folds the initial value into the reduction (simple.cuh:56-77: the
// initial value is cast to the result type first; integers wrap like the device arithmetic does)
template <typename T>
T fold_init(int op, T r, T init)
{
  if constexpr (std::is_integral_v<T>) {
    using U = std::make_unsigned_t<T>;
    if (op == GX_OP_SUM) return static_cast<T>(static_cast<U>(static_cast<U>(r) + static_cast<U>(init)));
    if (op == GX_OP_PRODUCT) return static_cast<T>(static_cast<U>(static_cast<U>(r) * static_cast<U>(init)));
  }
  switch (op) {
    case GX_OP_SUM: return static_cast<T>(r + init);
    case GX_OP_PRODUCT: return static_cast<T>(r * init);
    case GX_OP_MIN: return init < r ? init : r;
    default: return init > r ? init : r;
  }
}

template <typename T>
T read_init(scalar const& s, rmm::cuda_stream_view stream)
{
  switch (s.type().id()) {
    case type_id::INT8: return static_cast<T>(static_cast<numeric_scalar<int8_t> const&>(s).value(stream));
    case type_id::INT16: return static_cast<T>(static_cast<numeric_scalar<int16_t> const&>(s).value(stream));
    case type_id::INT32: return static_cast<T>(static_cast<numeric_scalar<int32_t> const&>(s).value(stream));
    case type_id::INT64: return static_cast<T>(static_cast<numeric_scalar<int64_t> const&>(s).value(stream));
    case type_id::UINT8: return static_cast<T>(static_cast<numeric_scalar<uint8_t> const&>(s).value(stream));
    case type_id::UINT16: return static_cast<T>(static_cast<numeric_scalar<uint16_t> const&>(s).value(stream));
    case type_id::UINT32: return static_cast<T>(static_cast<numeric_scalar<uint32_t> const&>(s).value(stream));
    case type_id::UINT64: return static_cast<T>(static_cast<numeric_scalar<uint64_t> const&>(s).value(stream));
    case type_id::FLOAT32: return static_cast<T>(static_cast<numeric_scalar<float> const&>(s).value(stream));
    case type_id::FLOAT64: return static_cast<T>(static_cast<numeric_scalar<double> const&>(s).value(stream));
    case type_id::BOOL8: return static_cast<T>(static_cast<numeric_scalar<bool> const&>(s).value(stream));
    default: throw cudf::data_type_error{"reduce: unsupported initial value type"};
  }
}

double read_init_as_double(scalar const& s, rmm::cuda_stream_view stream) { return read_init<double>(s, stream); }

template <typename T>
void apply_init(int op, scalar& result, scalar const& init, rmm::cuda_stream_view stream)
{
  auto& r = static_cast<numeric_scalar<T>&>(result);
  r.set_value(fold_init<T>(op, r.value(stream), read_init<T>(init, stream)), stream);
}

}  // namespace

// reduce with an initial value (reduction.hpp:124-130; reductions.cpp:484-507; simple.cuh:47-85): result = op(init, reduce(col));
// valid iff the column has a valid row AND the initial value is valid.
std::unique_ptr<scalar> reduce(column_view const& col, reduce_aggregation const& agg, data_type output_type,
                               std::optional<std::reference_wrapper<scalar const>> init, rmm::cuda_stream_view stream,
                               rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(!init.has_value() || init.value().get().type() == col.type(), "column and initial value must be the same type",
               cudf::data_type_error);
  if (init.has_value() && !(agg.kind == aggregation::SUM || agg.kind == aggregation::PRODUCT || agg.kind == aggregation::MIN ||
                            agg.kind == aggregation::MAX || agg.kind == aggregation::ANY || agg.kind == aggregation::ALL))
    // (the reference also folds an initial value into SUM_WITH_OVERFLOW and HOST_UDF -- reductions.cpp:484-507; those two
    //  aggregations are not part of this hot path at all, with or without an initial value: INTEGRATION.md section 5)
    throw std::invalid_argument{"Initial value is only supported for SUM, PRODUCT, MIN, MAX, ANY and ALL on this path (the reference's SUM_OVERFLOW and HOST_UDF reductions are not implemented here)"};
  if (agg.kind == aggregation::ANY || agg.kind == aggregation::ALL)
    return reduce_any_all(col, agg.kind == aggregation::ANY, output_type, init, stream, mr);
  auto result = reduce(col, agg, output_type, stream, mr);
  if (!init.has_value()) return result;
  if (!init.value().get().is_valid(stream)) {
    result->set_valid_async(false, stream);
    return result;
  }
  if (!result->is_valid(stream)) return result;  // no valid row: reduce_no_data, invalid whatever the initial value
  int const op = gx_op_of(agg.kind);
  switch (output_type.id()) {
    case type_id::INT8: apply_init<int8_t>(op, *result, init.value().get(), stream); break;
    case type_id::INT16: apply_init<int16_t>(op, *result, init.value().get(), stream); break;
    case type_id::INT32: apply_init<int32_t>(op, *result, init.value().get(), stream); break;
    case type_id::INT64: apply_init<int64_t>(op, *result, init.value().get(), stream); break;
    case type_id::UINT8: apply_init<uint8_t>(op, *result, init.value().get(), stream); break;
    case type_id::UINT16: apply_init<uint16_t>(op, *result, init.value().get(), stream); break;
    case type_id::UINT32: apply_init<uint32_t>(op, *result, init.value().get(), stream); break;
    case type_id::UINT64: apply_init<uint64_t>(op, *result, init.value().get(), stream); break;
    case type_id::FLOAT32: apply_init<float>(op, *result, init.value().get(), stream); break;
    case type_id::FLOAT64: apply_init<double>(op, *result, init.value().get(), stream); break;
    default: CUDF_FAIL("reduce: unsupported output type");
  }
  return result;
}

std::unique_ptr<column> scan(column_view const& input, scan_aggregation const& agg, scan_type inclusive,
                             null_policy null_handling, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  int const op = gx_op_of(agg.kind);
  CUDF_EXPECTS(is_fixed_width(input.type()), "scan: only fixed-width columns are supported on this path", cudf::data_type_error);
  auto const n = input.size();
  auto out     = make_fixed_width_column(input.type(), n, mask_state::UNALLOCATED, stream, mr);
  if (n == 0) return out;
  rmm::device_buffer holder;
  auto const* mask = input.has_nulls() ? detail::rebased_mask(input, holder, stream) : nullptr;
  detail::run_with_scratch(
    [&](void* t, std::size_t* b) {
      return gx_scan(detail::gx_type(input.type()), detail::row0(input), mask, n, op, inclusive == scan_type::INCLUSIVE ? 1 : 0,
                     out->mutable_view().head<void>(), t, b, detail::gxs(stream));
    },
    "scan", stream);
  if (input.nullable()) {
    if (null_handling == null_policy::EXCLUDE) {  // nulls stay null (scan_inclusive.cu:198-216)
      rmm::device_buffer h2;
      auto const* m = detail::rebased_mask(input, h2, stream);
      out->set_null_mask(rmm::device_buffer{m, bitmask_allocation_size_bytes(n), stream, mr}, input.null_count());
    } else {  // the first null poisons the rest (mask_scan :36-61)
      size_type first = n;
      if (input.has_nulls()) {
        rmm::device_buffer pos{sizeof(int64_t), stream};
        detail::gx_check(gx_bitmask_first_unset(mask, n, static_cast<int64_t*>(pos.data()), detail::gxs(stream)), "first_unset");
        auto const p = detail::read_i64(static_cast<int64_t const*>(pos.data()), stream);
        first        = static_cast<size_type>(std::min<int64_t>(n, p + (inclusive == scan_type::INCLUSIVE ? 0 : 1)));
      }
      auto m = create_null_mask(n, mask_state::ALL_NULL, stream, mr);
      set_null_mask(static_cast<bitmask_type*>(m.data()), 0, first, true, stream);
      out->set_null_mask(std::move(m), n - first);
    }
  }
  return out;
}

// cudf::minmax (src/reductions/minmax.cu:214-257): the reference folds (min, max) pairs in one pass; here the two reductions of the
// hot path back to back -- the second reads what the first left in the Infinity Cache for columns up to its size
std::pair<std::unique_ptr<scalar>, std::unique_ptr<scalar>> minmax(column_view const& col, rmm::cuda_stream_view stream,
                                                                   rmm::device_async_resource_ref mr)
{
  auto lo = reduce(col, *make_min_aggregation<reduce_aggregation>(), col.type(), stream, mr);
  auto hi = reduce(col, *make_max_aggregation<reduce_aggregation>(), col.type(), stream, mr);
  return {std::move(lo), std::move(hi)};
}

namespace hashing {
std::unique_ptr<column> murmurhash3_x86_32(table_view const& input, uint32_t seed, rmm::cuda_stream_view stream,
                                           rmm::device_async_resource_ref mr)
{
  auto const n = input.num_rows();
  auto out     = make_numeric_column(data_type{type_id::UINT32}, n, mask_state::UNALLOCATED, stream, mr);
  if (n == 0 || input.num_columns() == 0) return out;
  int k = 0;
  for (auto const& c : input) {
    rmm::device_buffer holder;
    auto const* mask = c.has_nulls() ? detail::rebased_mask(c, holder, stream) : nullptr;
    detail::gx_check(gx_murmur3_32(detail::gx_type(c.type()), detail::row0(c), mask, n, seed, k > 0 ? 1 : 0,
                                   out->mutable_view().head<uint32_t>(), detail::gxs(stream)),
                     "murmurhash3_x86_32");
    ++k;
  }
  return out;
}
}  // namespace hashing

}  // namespace cudf
