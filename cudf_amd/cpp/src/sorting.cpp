// cudf::sorted_order / sort / sort_by_key (+ stable_ variants) and cudf::gather over the C ABI.
// reference: cpp/src/sort/{sort.cu:31-89, sort_impl.cuh:32-99, sort_column.cu:22-44, sort_radix.cu:52-161,
// sorted_order_radix.cu:56-179, stable_sort.cu}, cpp/src/copying/gather.cu.
#include "common.hpp"

#include <cudf/column/column_factories.hpp>
#include <cudf/copying.hpp>
#include <cudf/sorting.hpp>

namespace cudf {
namespace {

// Device-side protocol faults: a look-back wait that makes no progress for 30 s of wall-clock time is ABANDONED inside the kernel
// (gx_sort.hip, spin_guard; opted into by SortFaultMode below -- other callers of the C-ABI sorts keep the trap): the scratch's status
// word becomes 5 and every write stays inside the output.  Round 6: the sorts here stay STREAM-ORDERED like the reference's
// (cpp/src/sort/sort.cu:52-89 returns once the work is queued) -- the word travels to pinned host memory behind the sort
// (detail::post_sort_status) and the fault surfaces at the next sort call or at cudf_amd::check_device_faults(stream) as
// cudf::cuda_error, the way a sticky device error reaches a caller of the reference (utilities/error.hpp:63-86).  Round 5 read the
// word back after every sort: one host round trip per call, and callers pipelining sorts over several streams serialised on it.

void check_order_args(table_view const& input, std::vector<order> const& column_order,
                      std::vector<null_order> const& null_precedence)
{
  if (!column_order.empty())
    CUDF_EXPECTS(static_cast<std::size_t>(input.num_columns()) == column_order.size(),
                 "Mismatch between number of columns and column order.", std::invalid_argument);
  if (!null_precedence.empty())
    CUDF_EXPECTS(static_cast<std::size_t>(input.num_columns()) == null_precedence.size(),
                 "Mismatch between number of columns and null_precedence size.", std::invalid_argument);
}

// stable argsort of ONE column into `out` (device int32[n]); out may not alias the column
void column_sorted_order(column_view const& col, order ord, null_order nulls, int32_t* out, rmm::cuda_stream_view stream)
{
  rmm::device_buffer mask_holder;
  auto const* mask      = col.has_nulls() ? detail::rebased_mask(col, mask_holder, stream) : nullptr;
  int const dtype       = detail::gx_type(col.type());
  int const descending  = ord == order::DESCENDING ? 1 : 0;
  int const null_before = nulls == null_order::BEFORE ? 1 : 0;
  detail::SortFaultMode soft;
  auto const tmp = detail::run_with_scratch(
    [&](void* t, std::size_t* b) {
      return gx_sorted_order(dtype, detail::row0(col), mask, col.size(), mask ? col.null_count() : 0, descending,
                             null_before, out, t, b, detail::gxs(stream));
    },
    "sorted_order", stream);
  if (col.size() > 0 && !mask) detail::post_sort_status(tmp, stream);  // (nullable columns: the validity split's scratch has no plan header in front)
}

// numeric column without nulls: a key column of the table path (gx_sorted_order_table)
bool table_path_column(column_view const& c)
{
  auto const id = static_cast<int>(c.type().id());
  return !c.has_nulls() && id >= GX_INT8 && id <= GX_BOOL8;
}

// the lexicographic stable argsort of 1 - 8 such columns in ONE word sort on a nested rank of the tuple (cudf_amd/csrc/gx_order.hip):
// the comparator semantics of sort_impl.cuh:61-93 -- NaN equivalent and greatest in both directions, ties of the tuple by row
void table_sorted_order(table_view const& keys, std::vector<order> const& column_order, int32_t* out, rmm::cuda_stream_view stream)
{
  int const k = keys.num_columns();
  std::vector<int> dtypes(k), desc(k);
  std::vector<void const*> datas(k);
  for (int c = 0; c < k; ++c) {
    dtypes[c] = detail::gx_type(keys.column(c).type());
    datas[c]  = detail::row0(keys.column(c));
    desc[c]   = (!column_order.empty() && column_order[c] == order::DESCENDING) ? 1 : 0;
  }
  detail::SortFaultMode soft;
  auto const tmp = detail::run_with_scratch(
    [&](void* t, std::size_t* b) {
      return gx_sorted_order_table(k, dtypes.data(), datas.data(), desc.data(), keys.num_rows(), out, t, b, detail::gxs(stream));
    },
    "sorted_order (table)", stream);
  detail::post_sort_status(tmp, stream);
}

std::unique_ptr<column> gather_column(column_view const& src, int32_t const* map, size_type n, bool nullify,
                                      rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(is_fixed_width(src.type()), "gather: only fixed-width columns are supported on this path",
               cudf::data_type_error);
  rmm::device_buffer mask_holder;
  auto const* smask    = src.nullable() ? detail::rebased_mask(src, mask_holder, stream) : nullptr;
  bool const need_mask = smask != nullptr || nullify;
  rmm::device_buffer data{static_cast<std::size_t>(n) * size_of(src.type()), stream, mr};
  rmm::device_buffer mask = need_mask ? create_null_mask(n, mask_state::ALL_NULL, stream, mr) : rmm::device_buffer{};
  detail::gx_check(gx_gather(static_cast<int>(size_of(src.type())), detail::row0(src), smask, src.size(), map, n,
                             nullify ? 1 : 0, data.data(), static_cast<uint32_t*>(mask.data()), detail::gxs(stream)),
                   "gather");
  size_type nulls = 0;
  if (need_mask && n > 0) nulls = null_count(static_cast<bitmask_type const*>(mask.data()), 0, n, stream);
  return std::make_unique<column>(src.type(), n, std::move(data), std::move(mask), nulls);
}

std::unique_ptr<column> sorted_order_impl(table_view const& input, std::vector<order> const& column_order,
                                          std::vector<null_order> const& null_precedence,
                                          rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  detail::throw_pending_sort_faults();  // an earlier sort's device-side fault surfaces here (non-blocking: completed sorts only)
  check_order_args(input, column_order, null_precedence);
  auto const n = input.num_rows();
  if (n == 0 || input.num_columns() == 0) return make_numeric_column(data_type{type_id::INT32}, 0, mask_state::UNALLOCATED, stream, mr);
  auto ord   = [&](size_type i) { return column_order.empty() ? order::ASCENDING : column_order[i]; };
  auto nulls = [&](size_type i) { return null_precedence.empty() ? null_order::BEFORE : null_precedence[i]; };

  auto result = make_numeric_column(data_type{type_id::INT32}, n, mask_state::UNALLOCATED, stream, mr);
  auto* out   = result->mutable_view().data<int32_t>();
  if (input.num_columns() == 1) {  // the radix fast path of sort_column.cu / sorted_order_radix.cu
    column_sorted_order(input.column(0), ord(0), nulls(0), out, stream);
    return result;
  }
  // numeric key columns without nulls (<= 8 of them): one word sort on the whole tuple -- 2 x int64 at 1e9 rows in about a third of the
  // time of the loop below
  constexpr size_type TABLE_PATH_MIN_ROWS = 1 << 18;
  bool table_path = input.num_columns() <= 8 && n >= TABLE_PATH_MIN_ROWS;
  for (auto const& c : input) table_path = table_path && table_path_column(c);
  if (table_path) {
    table_sorted_order(input, column_order, out, stream);
    return result;
  }
  // Lexicographic order of several columns = LSD over the columns: stable sorts from the least to
  // the most significant column, each on the column gathered through the order so far.
  rmm::device_uvector<int32_t> perm(n, stream), order_so_far(n, stream);
  bool first = true;
  for (size_type c = input.num_columns() - 1; c >= 0; --c) {
    if (first) {
      column_sorted_order(input.column(c), ord(c), nulls(c), order_so_far.data(), stream);
      first = false;
      continue;
    }
    auto gathered = gather_column(input.column(c), order_so_far.data(), n, false, stream, cudf::get_current_device_resource_ref());
    column_sorted_order(gathered->view(), ord(c), nulls(c), perm.data(), stream);
    // order_so_far = order_so_far[perm]
    rmm::device_uvector<int32_t> next(n, stream);
    detail::gx_check(gx_gather(4, order_so_far.data(), nullptr, n, perm.data(), n, 0, next.data(), nullptr, detail::gxs(stream)),
                     "gather order");
    order_so_far = std::move(next);
  }
  CUDF_CUDA_TRY(hipMemcpyAsync(out, order_so_far.data(), static_cast<std::size_t>(n) * 4, hipMemcpyDeviceToDevice, stream.value()));
  return result;
}

}  // namespace

std::unique_ptr<table> gather(table_view const& source_table, column_view const& gather_map,
                              out_of_bounds_policy bounds_policy, rmm::cuda_stream_view stream,
                              rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(not gather_map.has_nulls(), "gather_map contains nulls", std::invalid_argument);
  CUDF_EXPECTS(gather_map.type().id() == type_id::INT32, "gather map must be an INT32 column", cudf::data_type_error);
  std::vector<std::unique_ptr<column>> cols;
  cols.reserve(source_table.num_columns());
  auto const* map = static_cast<int32_t const*>(detail::row0(gather_map));
  for (auto const& c : source_table)
    cols.emplace_back(gather_column(c, map, gather_map.size(), bounds_policy == out_of_bounds_policy::NULLIFY, stream, mr));
  return std::make_unique<table>(std::move(cols));
}

std::unique_ptr<column> sorted_order(table_view const& input, std::vector<order> const& column_order,
                                     std::vector<null_order> const& null_precedence, rmm::cuda_stream_view stream,
                                     rmm::device_async_resource_ref mr)
{
  return sorted_order_impl(input, column_order, null_precedence, stream, mr);
}

std::unique_ptr<column> stable_sorted_order(table_view const& input, std::vector<order> const& column_order,
                                            std::vector<null_order> const& null_precedence,
                                            rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  return sorted_order_impl(input, column_order, null_precedence, stream, mr);
}

// cudf::is_sorted (src/sort/is_sorted.cu:27-86).  One integer column without nulls: one streaming pass that counts the adjacent pairs
// out of order (gx_checksum's violation count: NaN the greatest value, as the sort's comparator has it).  Everything else -- several
// columns, nulls -- through the STABLE argsort: the rows are in order exactly when that argsort is the identity (equal rows keep
// their places in a stable sort, and any adjacent pair out of order moves), compared with an iota on the device.
bool is_sorted(table_view const& in, std::vector<order> const& column_order, std::vector<null_order> const& null_precedence,
               rmm::cuda_stream_view stream)
{
  if (in.num_columns() == 0 || in.num_rows() == 0) return true;
  if (!column_order.empty())
    CUDF_EXPECTS(static_cast<std::size_t>(in.num_columns()) == column_order.size(),
                 "Number of columns in the table doesn't match the vector column_order's size .\n");
  if (!null_precedence.empty())
    CUDF_EXPECTS(static_cast<std::size_t>(in.num_columns()) == null_precedence.size(),
                 "Number of columns in the table doesn't match the vector null_precedence's size .\n");
  auto const n = in.num_rows();
  // (integers only: the streaming pass orders floats by their bit image, where -0.0 < +0.0 and NaNs differ -- the comparator has them equal)
  if (in.num_columns() == 1 && !in.column(0).has_nulls() && table_path_column(in.column(0)) && !is_floating_point(in.column(0).type())) {
    auto const& c  = in.column(0);
    int const desc = (!column_order.empty() && column_order[0] == order::DESCENDING) ? 1 : 0;
    rmm::device_uvector<uint64_t> res(3, stream);
    detail::gx_check(gx_checksum(detail::gx_type(c.type()), detail::row0(c), n, desc, res.data(), detail::gxs(stream)), "is_sorted");
    uint64_t h[3] = {0, 0, 0};
    CUDF_CUDA_TRY(hipMemcpyAsync(h, res.data(), sizeof(h), hipMemcpyDeviceToHost, stream.value()));
    stream.synchronize();
    return h[2] == 0;
  }
  auto const ord = stable_sorted_order(in, column_order, null_precedence, stream, cudf::get_current_device_resource_ref());
  rmm::device_uvector<int32_t> iota(n, stream);
  detail::gx_check(gx_sequence_i32(iota.data(), n, 0, detail::gxs(stream)), "is_sorted");
  rmm::device_uvector<int64_t> bad(1, stream);
  void const* l[1] = {ord->view().head<int32_t>()};
  void const* r[1] = {iota.data()};
  int const dt[1]  = {GX_INT32};
  detail::gx_check(gx_rows_mismatch_count(1, l, r, dt, nullptr, nullptr, n, bad.data(), detail::gxs(stream)), "is_sorted");
  return detail::read_i64(bad.data(), stream) == 0;
}

std::unique_ptr<table> sort_by_key(table_view const& values, table_view const& keys,
                                   std::vector<order> const& column_order,
                                   std::vector<null_order> const& null_precedence, rmm::cuda_stream_view stream,
                                   rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(values.num_rows() == keys.num_rows(), "Mismatch in number of rows for values and keys",
               std::invalid_argument);
  auto order_col = sorted_order_impl(keys, column_order, null_precedence, stream, cudf::get_current_device_resource_ref());
  return gather(values, order_col->view(), out_of_bounds_policy::DONT_CHECK, stream, mr);
}

std::unique_ptr<table> stable_sort_by_key(table_view const& values, table_view const& keys,
                                          std::vector<order> const& column_order,
                                          std::vector<null_order> const& null_precedence,
                                          rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  return sort_by_key(values, keys, column_order, null_precedence, stream, mr);
}

std::unique_ptr<table> sort(table_view const& input, std::vector<order> const& column_order,
                            std::vector<null_order> const& null_precedence, rmm::cuda_stream_view stream,
                            rmm::device_async_resource_ref mr)
{
  detail::throw_pending_sort_faults();
  check_order_args(input, column_order, null_precedence);
  // fast path of sort.cu:57-64: one fixed-width column without nulls -> keys-only radix sort
  if (input.num_columns() == 1 && !input.column(0).has_nulls() && is_fixed_width(input.column(0).type()) &&
      input.num_rows() > 0) {
      auto const& col = input.column(0);
    auto out        = make_fixed_width_column(col.type(), col.size(), mask_state::UNALLOCATED, stream, mr);
    int const desc  = (!column_order.empty() && column_order[0] == order::DESCENDING) ? 1 : 0;
    detail::SortFaultMode soft;
    auto const tmp = detail::run_with_scratch(
      [&](void* t, std::size_t* b) {
        return gx_sort_keys(detail::gx_type(col.type()), detail::row0(col), out->mutable_view().head<void>(), col.size(),
                            desc, t, b, detail::gxs(stream));
      },
      "sort", stream);
    detail::post_sort_status(tmp, stream);
    std::vector<std::unique_ptr<column>> cols;
    cols.emplace_back(std::move(out));
    return std::make_unique<table>(std::move(cols));
  }
  return sort_by_key(input, input, column_order, null_precedence, stream, mr);
}

std::unique_ptr<table> stable_sort(table_view const& input, std::vector<order> const& column_order,
                                   std::vector<null_order> const& null_precedence, rmm::cuda_stream_view stream,
                                   rmm::device_async_resource_ref mr)
{
  return sort(input, column_order, null_precedence, stream, mr);
}

}  // namespace cudf
