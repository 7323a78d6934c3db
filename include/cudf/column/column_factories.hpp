// cudf/column/column_factories.hpp (reference: cpp/include/cudf/column/column_factories.hpp:40-140)
#pragma once
#include <cudf/column/column.hpp>

namespace cudf {

std::unique_ptr<column> make_empty_column(data_type type);
std::unique_ptr<column> make_empty_column(type_id id);

// uninitialised fixed-width column; mask allocated according to `state`
std::unique_ptr<column> make_numeric_column(data_type type, size_type size, mask_state state = mask_state::UNALLOCATED,
                                            rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                                            rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());
std::unique_ptr<column> make_fixed_width_column(data_type type, size_type size,
                                                mask_state state                  = mask_state::UNALLOCATED,
                                                rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                                                rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

}  // namespace cudf
