// cudf/join/distinct_hash_join.hpp -- hash join against a build table whose key rows are DISTINCT
// (reference: cpp/include/cudf/join/distinct_hash_join.hpp:50-124; impl cpp/src/join/distinct_hash_join.cu).
// With distinct build keys a probe row has at most one partner, so left_join needs no output
// reservation: it returns, in probe order, the build row of every probe row or JoinNoMatch
// (gx_join_lookup).  Behaviour is undefined if the build table holds duplicate key rows, as in the
// reference (distinct_hash_join.hpp:43-45).
#pragma once
#include <cudf/join/hash_join.hpp>

#include <memory>

namespace cudf {

class distinct_hash_join {
 public:
  distinct_hash_join() = delete;
  ~distinct_hash_join();
  distinct_hash_join(distinct_hash_join const&)            = delete;
  distinct_hash_join(distinct_hash_join&&)                 = delete;
  distinct_hash_join& operator=(distinct_hash_join const&) = delete;
  distinct_hash_join& operator=(distinct_hash_join&&)      = delete;

  // throws std::invalid_argument if `right` has no columns or load_factor is not in (0, 1]
  distinct_hash_join(table_view const& right, null_equality compare_nulls = null_equality::EQUAL, double load_factor = 0.5,
                     rmm::cuda_stream_view stream = cudf::get_default_stream());

  // (left_indices, right_indices) of the matching rows, order unspecified
  [[nodiscard]] join_result inner_join(table_view const& left, rmm::cuda_stream_view stream = cudf::get_default_stream(),
                                       rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref()) const;
  // right_indices[i] = the right row matching left row i, or JoinNoMatch: a gather map for the right
  // table aligned with the left table (distinct_hash_join.hpp:96-116)
  [[nodiscard]] std::unique_ptr<rmm::device_uvector<size_type>> left_join(
    table_view const& left, rmm::cuda_stream_view stream = cudf::get_default_stream(),
    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref()) const;

 private:
  std::unique_ptr<detail::hash_join_impl const> _impl;
};

}  // namespace cudf
