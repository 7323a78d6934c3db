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

}  // namespace cudf
