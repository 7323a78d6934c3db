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

// number of matching right rows for every left row: what hash_join::*_join_match_context returns (join.hpp:75-112)
struct join_match_context {
  table_view _left_table;                                         // view of the left (probe) table
  std::unique_ptr<rmm::device_uvector<size_type>> _match_counts;  // matches per left row
  join_match_context(table_view const& left_table, std::unique_ptr<rmm::device_uvector<size_type>> match_counts)
    : _left_table{left_table}, _match_counts{std::move(match_counts)}
  {
  }
  join_match_context(join_match_context const&)            = delete;
  join_match_context& operator=(join_match_context const&) = delete;
  join_match_context(join_match_context&&)                 = default;
  join_match_context& operator=(join_match_context&&)      = default;
  virtual ~join_match_context()                            = default;
};

// one chunk [left_start_idx, left_end_idx) of the left table of a partitioned join (join.hpp:114-125)
struct join_partition_context {
  std::unique_ptr<join_match_context> left_table_context;
  size_type left_start_idx;
  size_type left_end_idx;
};

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
