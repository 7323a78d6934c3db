// cudf/join/join.hpp -- free-function equality joins returning gather maps
// (reference: cpp/include/cudf/join/join.hpp:45-240; impl cpp/src/join/join.cu:27-124).
#pragma once
#include <cudf/table/table_view.hpp>
#include <cudf/types.hpp>
#include <cudf/utilities/default_stream.hpp>
#include <cudf/utilities/memory_resource.hpp>
#include <rmm/device_uvector.hpp>

#include <limits>
#include <memory>
#include <utility>

namespace cudf {

// sentinel row index for "no match" in outer joins
constexpr size_type JoinNoMatch = std::numeric_limits<size_type>::min();

enum class join_kind : int32_t { INNER_JOIN = 0, LEFT_JOIN = 1, FULL_JOIN = 2, LEFT_SEMI_JOIN = 3, LEFT_ANTI_JOIN = 4 };

using join_result = std::pair<std::unique_ptr<rmm::device_uvector<size_type>>,
                              std::unique_ptr<rmm::device_uvector<size_type>>>;

// (left_indices, right_indices) of all row pairs with equal keys; order unspecified.  Builds
// the hash table on the smaller input and swaps the pair back (join.cu:49-59).
join_result inner_join(table_view const& left_keys, table_view const& right_keys,
                       null_equality compare_nulls       = null_equality::EQUAL,
                       rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                       rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

// every left row appears; unmatched rows pair with JoinNoMatch
join_result left_join(table_view const& left_keys, table_view const& right_keys,
                      null_equality compare_nulls       = null_equality::EQUAL,
                      rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                      rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

// left join plus the unmatched right rows as (JoinNoMatch, right_index)
join_result full_join(table_view const& left_keys, table_view const& right_keys,
                      null_equality compare_nulls       = null_equality::EQUAL,
                      rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                      rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

}  // namespace cudf
