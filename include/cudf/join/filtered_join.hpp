// cudf/join/filtered_join.hpp -- left semi / left anti join object
// (reference: cpp/include/cudf/join/filtered_join.hpp:51-144; impl cpp/src/join/filtered_join/
// filtered_join.cu:124-222).  The right (filter) table is hashed once; semi_join / anti_join return the
// ASCENDING row indices of `left` that have / have no match -- the reference builds a contains map and
// runs thrust::copy_if over it (filtered_join.cu:137-156), which fixes the order; here gx_join_filter
// produces the same list (contains bits + chunked ordered compaction).
// The object views the right table: it must not outlive it.
#pragma once
#include <cudf/join/hash_join.hpp>

#include <memory>

namespace cudf {

class filtered_join {
 public:
  filtered_join() = delete;
  ~filtered_join();
  filtered_join(filtered_join const&)            = delete;
  filtered_join(filtered_join&&)                 = delete;
  filtered_join& operator=(filtered_join const&) = delete;
  filtered_join& operator=(filtered_join&&)      = delete;

  filtered_join(table_view const& right, null_equality compare_nulls, rmm::cuda_stream_view stream);
  // throws std::invalid_argument if load_factor is not in (0, 1]
  filtered_join(table_view const& right, null_equality compare_nulls, double load_factor, rmm::cuda_stream_view stream);

  // rows of `left` with at least one match in the right table (empty right table: none)
  [[nodiscard]] std::unique_ptr<rmm::device_uvector<size_type>> semi_join(
    table_view const& left, rmm::cuda_stream_view stream = cudf::get_default_stream(),
    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref()) const;
  // rows of `left` with no match in the right table (empty right table: all)
  [[nodiscard]] std::unique_ptr<rmm::device_uvector<size_type>> anti_join(
    table_view const& left, rmm::cuda_stream_view stream = cudf::get_default_stream(),
    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref()) const;

 private:
  std::unique_ptr<detail::hash_join_impl const> _impl;  // null when the right table is empty
  size_type _right_rows{0};
};

}  // namespace cudf
