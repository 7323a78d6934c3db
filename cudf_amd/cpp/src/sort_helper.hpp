// sort_helper.hpp -- cudf::groupby::detail::sort::sort_groupby_helper over the C ABI: the sorted order of the
// keys, the group boundaries and the per-group views that the sort-based aggregations, groupby::scan,
// get_groups, shift and replace_nulls share.
// reference: cpp/include/cudf/detail/groupby/sort_helper.hpp, cpp/src/groupby/sort/sort_helper.cu:37-260.
#pragma once
#include "common.hpp"

#include <cudf/column/column.hpp>
#include <cudf/table/table.hpp>
#include <cudf/table/table_view.hpp>

#include <memory>
#include <vector>

namespace cudf {
namespace groupby {
namespace sort_impl {  // the reference's cudf::groupby::detail::sort (a nested `detail` here would shadow cudf::detail)

class sort_groupby_helper {
 public:
  // Pre-sorted keys are trusted only when no row has to be dropped: with nulls to exclude the rows are
  // re-sorted so that they gather at the end (sort_helper.cu:47-54).
  sort_groupby_helper(table_view const& keys, null_policy include_null_keys, sorted keys_pre_sorted,
                      std::vector<null_order> const& null_precedence);

  // rows that take part: all of them, or those without a null key (null_policy::EXCLUDE)
  size_type num_keys(rmm::cuda_stream_view stream);
  // INT32 map: sorted position -> row (the first num_keys() entries count); an iota for pre-sorted keys
  column_view key_sort_order(rmm::cuda_stream_view stream);
  [[nodiscard]] bool is_presorted() const { return _keys_pre_sorted == sorted::YES; }

  size_type num_groups(rmm::cuda_stream_view stream);
  int32_t const* group_offsets(rmm::cuda_stream_view stream);  // device, num_groups + 1 entries
  int32_t const* group_labels(rmm::cuda_stream_view stream);   // device, num_keys entries
  int32_t const* group_sizes(rmm::cuda_stream_view stream);    // device, num_groups entries
  uint8_t const* group_heads(rmm::cuda_stream_view stream);    // device, num_keys entries: 1 = first row of a group
  std::vector<size_type> group_offsets_host(rmm::cuda_stream_view stream);

  // one row per group (the group's first row), in sorted-key order
  std::unique_ptr<table> unique_keys(rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr);
  // every kept row, in sorted-key order
  std::unique_ptr<table> sorted_keys(rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr);
  // `values` in sorted-key order (num_keys rows)
  std::unique_ptr<column> grouped_values(column_view const& values, rmm::cuda_stream_view stream,
                                         rmm::device_async_resource_ref mr);

 private:
  void build_groups(rmm::cuda_stream_view stream);

  table_view _keys;
  size_type _num_keys{-1};
  sorted _keys_pre_sorted;
  null_policy _include_null_keys;
  std::vector<null_order> _null_precedence;
  std::unique_ptr<column> _order;        // INT32, all rows
  rmm::device_buffer _heads, _labels, _offsets, _sizes;
  size_type _num_groups{-1};
};

}  // namespace sort_impl
}  // namespace groupby
}  // namespace cudf
