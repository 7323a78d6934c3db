// cudf/groupby.hpp -- groupby aggregate / scan (reference: cpp/include/cudf/groupby.hpp:54-240;
// impl cpp/src/groupby/groupby.cu:40-259, hash path cpp/src/groupby/hash/*, scan path
// cpp/src/groupby/sort/{scan.cpp,sort_helper.cu,group_scan_util.cuh}).
#pragma once
#include <cudf/aggregation.hpp>
#include <cudf/column/column.hpp>
#include <cudf/replace.hpp>
#include <cudf/scalar/scalar.hpp>
#include <cudf/table/table.hpp>
#include <cudf/table/table_view.hpp>
#include <cudf/types.hpp>

#include <functional>
#include <memory>
#include <optional>
#include <span>
#include <utility>
#include <vector>

namespace cudf {
namespace groupby {

namespace sort_impl {
class sort_groupby_helper;  // the reference's detail::sort::sort_groupby_helper (detail/groupby/sort_helper.hpp)
}

struct aggregation_request {
  column_view values;                                              // the elements to aggregate
  std::vector<std::unique_ptr<groupby_aggregation>> aggregations;  // desired aggregations
};
struct scan_request {
  column_view values;
  std::vector<std::unique_ptr<groupby_scan_aggregation>> aggregations;
};
struct aggregation_result {
  std::vector<std::unique_ptr<column>> results{};  // one column per requested aggregation
};

class groupby {
 public:
  groupby() = delete;
  ~groupby();
  groupby(groupby const&)            = delete;
  groupby(groupby&&)                 = delete;
  groupby& operator=(groupby const&) = delete;
  groupby& operator=(groupby&&)      = delete;

  // The object views `keys`.  null_policy::EXCLUDE drops rows with a null key.
  explicit groupby(table_view const& keys, null_policy null_handling = null_policy::EXCLUDE,
                   sorted keys_are_sorted = sorted::NO, std::vector<order> const& column_order = {},
                   std::vector<null_order> const& null_precedence = {});

  // {unique keys (unspecified order), one aggregation_result per request}
  // throws cudf::logic_error "Size mismatch between request values and groupby keys."
  // Hash-based unless the keys are pre-sorted (sorted::YES), null keys are to be kept (null_policy::INCLUDE) or an
  // aggregation needs the sort path (PRODUCT): then sort-based, keys come out sorted (groupby.cu:54-71).
  std::pair<std::unique_ptr<table>, std::vector<aggregation_result>> aggregate(
    std::span<aggregation_request const> requests, rmm::cuda_stream_view stream = cudf::get_default_stream(),
    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

  // {keys sorted, per-request inclusive scans within each group in sorted-key order}
  std::pair<std::unique_ptr<table>, std::vector<aggregation_result>> scan(
    std::span<scan_request const> requests, rmm::cuda_stream_view stream = cudf::get_default_stream(),
    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

  // every column of `values` shifted by its offset INSIDE its group, filled with the scalar where no row exists;
  // {sorted keys, shifted values} (groupby.hpp:242-300)
  std::pair<std::unique_ptr<table>, std::unique_ptr<table>> shift(
    table_view const& values, std::span<size_type const> offsets,
    std::vector<std::reference_wrapper<scalar const>> const& fill_values,
    rmm::cuda_stream_view stream = cudf::get_default_stream(),
    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

  struct groups {
    std::unique_ptr<table> keys;     // grouped keys (every kept row, sorted)
    std::vector<size_type> offsets;  // group offsets (num_groups + 1)
    std::unique_ptr<table> values;   // grouped values (when values were given)
  };
  groups get_groups(cudf::table_view values = {}, rmm::cuda_stream_view stream = cudf::get_default_stream(),
                    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

  // nulls replaced by the preceding / following valid value of the same group; {sorted keys, replaced values}
  std::pair<std::unique_ptr<table>, std::unique_ptr<table>> replace_nulls(
    table_view const& values, std::span<cudf::replace_policy const> replace_policies,
    rmm::cuda_stream_view stream = cudf::get_default_stream(),
    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

 private:
  table_view _keys;
  null_policy _include_null_keys{null_policy::EXCLUDE};
  sorted _keys_are_sorted{sorted::NO};
  std::vector<order> _column_order{};
  std::vector<null_order> _null_precedence{};
  std::unique_ptr<sort_impl::sort_groupby_helper> _helper;  // built on first use by the sort-based paths
  sort_impl::sort_groupby_helper& helper();

  std::pair<std::unique_ptr<table>, std::vector<aggregation_result>> sort_aggregate(
    std::span<aggregation_request const> requests, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr);
  // several 8-byte integer key columns, one SUM / COUNT / MEAN request, no nulls: one partition pass, rows compared in the LDS
  // tables (gx_groupby_sum_count_wide); nullopt = not applicable or declined by the device
  std::optional<std::pair<std::unique_ptr<table>, std::vector<aggregation_result>>> wide_aggregate(
    std::span<aggregation_request const> requests, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr);
  // exact_keys: encode multi-column keys through dense ranks instead of 8-byte row keys (the retry after a hash collision)
  std::pair<std::unique_ptr<table>, std::vector<aggregation_result>> aggregate_impl(
    std::span<aggregation_request const> requests, bool exact_keys, rmm::cuda_stream_view stream,
    rmm::device_async_resource_ref mr);
};

}  // namespace groupby
}  // namespace cudf
