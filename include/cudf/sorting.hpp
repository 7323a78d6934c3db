// cudf/sorting.hpp -- the sort entry points of the hot path
// (reference: cpp/include/cudf/sorting.hpp:44-163; impl cpp/src/sort/{sort.cu,sort_impl.cuh,
// sort_radix.cu,sorted_order_radix.cu,stable_sort.cu}).
// Defaults: empty column_order = all ASCENDING; empty null_precedence = all null_order::BEFORE.
#pragma once
#include <cudf/column/column.hpp>
#include <cudf/table/table.hpp>
#include <cudf/table/table_view.hpp>
#include <cudf/types.hpp>

#include <memory>
#include <vector>

namespace cudf {

// row indices (INT32, non-nullable) that would sort `input` lexicographically.  Stable here in
// both variants (the radix path is stable in the reference too: sorted_order_radix.cu:81).
std::unique_ptr<column> sorted_order(table_view const& input, std::vector<order> const& column_order = {},
                                     std::vector<null_order> const& null_precedence = {},
                                     rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                                     rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());
std::unique_ptr<column> stable_sorted_order(table_view const& input, std::vector<order> const& column_order = {},
                                            std::vector<null_order> const& null_precedence = {},
                                            rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                                            rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

// whether the rows of `table` are in the given lexicographic order (cpp/include/cudf/sorting.hpp:83-86; src/sort/is_sorted.cu:27-86):
// empty `column_order` = all ascending, empty `null_precedence` = nulls before; no columns or no rows -> true
bool is_sorted(table_view const& table, std::vector<order> const& column_order, std::vector<null_order> const& null_precedence,
               rmm::cuda_stream_view stream = cudf::get_default_stream());

// new table with the rows of `input` in sorted order
std::unique_ptr<table> sort(table_view const& input, std::vector<order> const& column_order = {},
                            std::vector<null_order> const& null_precedence = {},
                            rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                            rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());
std::unique_ptr<table> stable_sort(table_view const& input, std::vector<order> const& column_order = {},
                                   std::vector<null_order> const& null_precedence = {},
                                   rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                                   rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

// gather(values, sorted_order(keys))
std::unique_ptr<table> sort_by_key(table_view const& values, table_view const& keys,
                                   std::vector<order> const& column_order         = {},
                                   std::vector<null_order> const& null_precedence = {},
                                   rmm::cuda_stream_view stream                   = cudf::get_default_stream(),
                                   rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());
std::unique_ptr<table> stable_sort_by_key(table_view const& values, table_view const& keys,
                                          std::vector<order> const& column_order         = {},
                                          std::vector<null_order> const& null_precedence = {},
                                          rmm::cuda_stream_view stream                   = cudf::get_default_stream(),
                                          rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

// ---- rank / top_k / segmented sort (sorting.hpp:166-509 of the reference; cpp/src/sort/{rank.cu,top_k.cu,
// segmented_sort.cu,segmented_sort_impl.cuh})
enum class rank_method : int32_t { FIRST, AVERAGE, MIN, MAX, DENSE };  // aggregation.hpp:37-43

// rank of every row of `input` in its sorted order: INT32, or FLOAT64 for AVERAGE / percentage.  null_policy::EXCLUDE
// leaves null rows null; percentage divides by the number of ranked rows (DENSE: by the number of distinct values).
std::unique_ptr<column> rank(column_view const& input, rank_method method, order column_order, null_policy null_handling,
                             null_order null_precedence, bool percentage,
                             rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                             rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

// the k largest (DESCENDING) / smallest (ASCENDING) values, and their row indices
std::unique_ptr<column> top_k(column_view const& col, size_type k, order topk_order = order::DESCENDING,
                              rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                              rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());
std::unique_ptr<column> top_k_order(column_view const& col, size_type k, order topk_order = order::DESCENDING,
                                    rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                                    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

// lexicographic order of `keys` INSIDE each segment [offsets[j], offsets[j+1]); rows outside every segment keep
// their place.  segment_offsets is an INT32 column.
std::unique_ptr<column> segmented_sorted_order(table_view const& keys, column_view const& segment_offsets,
                                               std::vector<order> const& column_order         = {},
                                               std::vector<null_order> const& null_precedence = {},
                                               rmm::cuda_stream_view stream                   = cudf::get_default_stream(),
                                               rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());
std::unique_ptr<column> stable_segmented_sorted_order(table_view const& keys, column_view const& segment_offsets,
                                                      std::vector<order> const& column_order         = {},
                                                      std::vector<null_order> const& null_precedence = {},
                                                      rmm::cuda_stream_view stream = cudf::get_default_stream(),
                                                      rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());
std::unique_ptr<table> segmented_sort_by_key(table_view const& values, table_view const& keys, column_view const& segment_offsets,
                                             std::vector<order> const& column_order         = {},
                                             std::vector<null_order> const& null_precedence = {},
                                             rmm::cuda_stream_view stream                   = cudf::get_default_stream(),
                                             rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());
std::unique_ptr<table> stable_segmented_sort_by_key(table_view const& values, table_view const& keys,
                                                    column_view const& segment_offsets,
                                                    std::vector<order> const& column_order         = {},
                                                    std::vector<null_order> const& null_precedence = {},
                                                    rmm::cuda_stream_view stream = cudf::get_default_stream(),
                                                    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

}  // namespace cudf
