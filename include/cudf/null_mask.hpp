// cudf/null_mask.hpp -- validity bitmap helpers (reference: cpp/include/cudf/null_mask.hpp:55-190;
// cpp/src/bitmask/null_mask.cu).  Bits are LSB-first in uint32 words, 1 = valid, allocation padded
// to 64 bytes (Arrow layout).
#pragma once
#include <cudf/types.hpp>
#include <cudf/utilities/default_stream.hpp>
#include <cudf/utilities/memory_resource.hpp>
#include <rmm/device_buffer.hpp>

#include <utility>
#include <vector>

namespace cudf {

class column_view;
class table_view;

size_type state_null_count(mask_state state, size_type size);
std::size_t bitmask_allocation_size_bytes(size_type number_of_bits, std::size_t padding_boundary = 64);
size_type num_bitmask_words(size_type number_of_bits);

rmm::device_buffer create_null_mask(size_type size, mask_state state,
                                    rmm::cuda_stream_view stream          = cudf::get_default_stream(),
                                    rmm::device_async_resource_ref mr     = cudf::get_current_device_resource_ref());
void set_null_mask(bitmask_type* bitmask, size_type begin_bit, size_type end_bit, bool valid,
                   rmm::cuda_stream_view stream = cudf::get_default_stream());
size_type null_count(bitmask_type const* bitmask, size_type start, size_type stop,
                     rmm::cuda_stream_view stream = cudf::get_default_stream());
// AND of the null masks of all columns of `view`; returns {mask, null_count}
std::pair<rmm::device_buffer, size_type> bitmask_and(
  table_view const& view, rmm::cuda_stream_view stream = cudf::get_default_stream(),
  rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

}  // namespace cudf
