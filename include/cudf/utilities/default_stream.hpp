// reference: cpp/include/cudf/utilities/default_stream.hpp; cpp/src/utilities/default_stream.cpp:39
#pragma once
#include <rmm/cuda_stream_view.hpp>
namespace cudf {
inline rmm::cuda_stream_view get_default_stream() { return rmm::cuda_stream_view{}; }
inline bool is_ptds_enabled() { return false; }
}  // namespace cudf
