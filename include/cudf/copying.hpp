// cudf/copying.hpp -- gather (reference: cpp/include/cudf/copying.hpp:48-95; kernels
// cpp/include/cudf/detail/gather.cuh:108-131,506-577).
#pragma once
#include <cudf/column/column.hpp>
#include <cudf/table/table.hpp>
#include <cudf/table/table_view.hpp>

#include <memory>

namespace cudf {

enum class out_of_bounds_policy : bool { NULLIFY, DONT_CHECK };

// out[i] = source_table[gather_map[i]] for every column.  gather_map must be a non-nullable
// INT32 column (the type of join / sorted_order outputs).  NULLIFY: rows whose index is outside
// [0, num_rows) -- e.g. JoinNoMatch -- become null.
std::unique_ptr<table> gather(table_view const& source_table, column_view const& gather_map,
                              out_of_bounds_policy bounds_policy = out_of_bounds_policy::DONT_CHECK,
                              rmm::cuda_stream_view stream       = cudf::get_default_stream(),
                              rmm::device_async_resource_ref mr  = cudf::get_current_device_resource_ref());

}  // namespace cudf
