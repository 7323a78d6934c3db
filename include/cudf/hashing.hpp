// cudf/hashing.hpp (reference: cpp/include/cudf/hashing.hpp:30-70).  cudf::hash_partition lives in <cudf/partitioning.hpp>
// as in the reference; this header keeps including it for the callers of earlier rounds.
#pragma once
#include <cudf/column/column.hpp>
#include <cudf/partitioning.hpp>
#include <cudf/table/table.hpp>
#include <cudf/table/table_view.hpp>

#include <memory>
#include <utility>
#include <vector>

namespace cudf {

using hash_value_type = uint32_t;

namespace hashing {
// UINT32 column of MurmurHash3_x86_32 row hashes (null element -> UINT32_MAX before combining)
std::unique_ptr<column> murmurhash3_x86_32(table_view const& input, uint32_t seed = DEFAULT_HASH_SEED,
                                           rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                                           rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());
}  // namespace hashing

}  // namespace cudf
