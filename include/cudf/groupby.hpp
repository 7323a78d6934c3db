// cudf/groupby.hpp -- groupby aggregate / scan (reference: cpp/include/cudf/groupby.hpp:54-240;
// impl cpp/src/groupby/groupby.cu:40-259, hash path cpp/src/groupby/hash/*, scan path
// cpp/src/groupby/sort/{scan.cpp,sort_helper.cu,group_scan_util.cuh}).
#pragma once
#include <cudf/aggregation.hpp>
#include <cudf/column/column.hpp>
#include <cudf/table/table.hpp>
#include <cudf/table/table_view.hpp>
#include <cudf/types.hpp>

#include <memory>
#include <span>
#include <utility>
#include <vector>

namespace cudf {
namespace groupby {

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
  std::pair<std::unique_ptr<table>, std::vector<aggregation_result>> aggregate(
    std::span<aggregation_request const> requests, rmm::cuda_stream_view stream = cudf::get_default_stream(),
    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

  // {keys sorted, per-request inclusive scans within each group in sorted-key order}
  std::pair<std::unique_ptr<table>, std::vector<aggregation_result>> scan(
    std::span<scan_request const> requests, rmm::cuda_stream_view stream = cudf::get_default_stream(),
    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

 private:
  table_view _keys;
  null_policy _include_null_keys{null_policy::EXCLUDE};
  sorted _keys_are_sorted{sorted::NO};
  std::vector<order> _column_order{};
  std::vector<null_order> _null_precedence{};
};

}  // namespace groupby
}  // namespace cudf
