// reference: cpp/include/cudf/utilities/memory_resource.hpp (get_current_device_resource_ref)
#pragma once
#include <rmm/resource_ref.hpp>
namespace cudf {
inline rmm::device_async_resource_ref get_current_device_resource_ref()
{
  return rmm::device_async_resource_ref{rmm::mr::get_default_resource()};
}
}  // namespace cudf
