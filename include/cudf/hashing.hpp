// cudf/hashing.hpp + cudf/partitioning.hpp subset (reference: cpp/include/cudf/hashing.hpp:30-70,
// cpp/include/cudf/partitioning.hpp:103-110).
#pragma once
#include <cudf/column/column.hpp>
#include <cudf/table/table.hpp>
#include <cudf/table/table_view.hpp>

#include <memory>
#include <utility>
#include <vector>

namespace cudf {

using hash_value_type = uint32_t;
static constexpr uint32_t DEFAULT_HASH_SEED = 0;
enum class hash_id { HASH_IDENTITY = 0, HASH_MURMUR3 };

namespace hashing {
// UINT32 column of MurmurHash3_x86_32 row hashes (null element -> UINT32_MAX before combining)
std::unique_ptr<column> murmurhash3_x86_32(table_view const& input, uint32_t seed = DEFAULT_HASH_SEED,
                                           rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                                           rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());
}  // namespace hashing

// rows of `input` regrouped into num_partitions partitions by murmur3(row of columns_to_hash) %
// num_partitions, plus the num_partitions start offsets
std::pair<std::unique_ptr<table>, std::vector<size_type>> hash_partition(
  table_view const& input, std::vector<size_type> const& columns_to_hash, int num_partitions,
  hash_id hash_function = hash_id::HASH_MURMUR3, uint32_t seed = DEFAULT_HASH_SEED,
  rmm::cuda_stream_view stream      = cudf::get_default_stream(),
  rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

}  // namespace cudf
