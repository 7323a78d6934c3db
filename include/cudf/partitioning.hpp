// cudf/partitioning.hpp -- cudf::hash_partition / cudf::partition (reference: cpp/include/cudf/partitioning.hpp:31-145;
// impl cpp/src/partitioning/partitioning.cu:53-92, 568-745, 875-972).  What feeds the shuffle of config 5
// (cpp/libcudf_streaming/src/partition_utils.cpp:72-117).
#pragma once
#include <cudf/column/column_view.hpp>
#include <cudf/table/table.hpp>
#include <cudf/table/table_view.hpp>
#include <cudf/types.hpp>
#include <cudf/utilities/default_stream.hpp>
#include <cudf/utilities/memory_resource.hpp>

#include <memory>
#include <utility>
#include <vector>

namespace cudf {

static constexpr uint32_t DEFAULT_HASH_SEED = 0;  // partitioning.hpp:41 / hashing.hpp:35
enum class hash_id { HASH_IDENTITY = 0, HASH_MURMUR3 };  // partitioning.hpp:31-34

// Rows of `t` regrouped by partition_map[i] in [0, num_partitions): rows of one partition are consecutive and keep their
// input order.  Returns the table and num_partitions + 1 offsets: partition i = rows [offsets[i], offsets[i+1]), the last
// offset is the number of rows (partitioning.hpp:44-78; partitioning.cu:903-921).  partition_map: INT32 or UINT32, no nulls.
std::pair<std::unique_ptr<table>, std::vector<size_type>> partition(
  table_view const& t, column_view const& partition_map, size_type num_partitions,
  rmm::cuda_stream_view stream      = cudf::get_default_stream(),
  rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

// Rows of `input` regrouped into num_partitions partitions by murmur3(row of the key columns) & (P - 1) for a power of two,
// % P otherwise (partitioning.cu:53-92, 620-665).  ALWAYS num_partitions + 1 offsets, the last one = number of output rows --
// also for an empty input, no key columns or num_partitions == 0, which return empty_like(input) and zeros
// (partitioning.cu:883-886; tests/partitioning/hash_partition_test.cpp:73-141).  An invalid column index throws
// std::out_of_range (partitioning.hpp:91).  hash_function: HASH_MURMUR3, or HASH_IDENTITY over any numeric key table
// (the element cast to uint32, partitioning.cu:852-889; the reference's own use: a column of precomputed row hashes,
// hash_partition_test.cpp:411-415).
std::pair<std::unique_ptr<table>, std::vector<size_type>> hash_partition(
  table_view const& input, std::vector<size_type> const& columns_to_hash, int num_partitions,
  hash_id hash_function = hash_id::HASH_MURMUR3, uint32_t seed = DEFAULT_HASH_SEED,
  rmm::cuda_stream_view stream      = cudf::get_default_stream(),
  rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

// The same with the keys in a table of their own (partitioning.hpp:112-145): `keys` has the rows of `input` (or no columns:
// an empty result); a row-count mismatch throws std::invalid_argument (partitioning.cu:932-935).
std::pair<std::unique_ptr<table>, std::vector<size_type>> hash_partition(
  table_view const& input, table_view const& keys, int num_partitions, hash_id hash_function = hash_id::HASH_MURMUR3,
  uint32_t seed = DEFAULT_HASH_SEED, rmm::cuda_stream_view stream = cudf::get_default_stream(),
  rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

// Round-robin partitioning (partitioning.hpp:287-292; src/partitioning/round_robin.cu:150-275): row i goes to partition
// (start_partition + i) % num_partitions, the rows of a partition keep their order; num_partitions + 1 offsets.  num_partitions <= 0,
// start_partition outside [0, num_partitions) throw cudf::logic_error (round_robin.cu:160-166).
std::pair<std::unique_ptr<table>, std::vector<size_type>> round_robin_partition(
  table_view const& input, size_type num_partitions, size_type start_partition = 0,
  rmm::cuda_stream_view stream = cudf::get_default_stream(), rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

}  // namespace cudf
