// cudf/join/hash_join.hpp -- build-once / probe-many hash join object
// (reference: cpp/include/cudf/join/hash_join.hpp:61-444; impl cpp/src/join/hash_join/hash_join.cu).
// The object views the build table: it must not outlive it (hash_join.hpp:83-84).  Probe methods
// are const and may be called concurrently from several host threads on distinct streams.
#pragma once
#include <cudf/join/join.hpp>
#include <cudf/utilities/span.hpp>

#include <cstddef>
#include <memory>
#include <optional>

namespace cudf {

enum class nullable_join : bool { YES, NO };

namespace detail {
class hash_join_impl;
}

class hash_join {
 public:
  hash_join() = delete;
  ~hash_join();
  hash_join(hash_join const&)            = delete;
  hash_join(hash_join&&)                 = delete;
  hash_join& operator=(hash_join const&) = delete;
  hash_join& operator=(hash_join&&)      = delete;

  // throws std::invalid_argument if `build` has no columns
  hash_join(table_view const& build, null_equality compare_nulls,
            rmm::cuda_stream_view stream = cudf::get_default_stream());
  // has_nulls: whether build or any later probe table may contain nulls; load_factor in (0, 1]
  hash_join(table_view const& build, nullable_join has_nulls, null_equality compare_nulls, double load_factor,
            rmm::cuda_stream_view stream = cudf::get_default_stream());

  [[nodiscard]] join_result inner_join(table_view const& probe, std::optional<std::size_t> output_size = {},
                                       rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                                       rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref()) const;
  [[nodiscard]] join_result left_join(table_view const& probe, std::optional<std::size_t> output_size = {},
                                      rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                                      rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref()) const;
  [[nodiscard]] join_result full_join(table_view const& probe, std::optional<std::size_t> output_size = {},
                                      rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                                      rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref()) const;

  [[nodiscard]] std::size_t inner_join_size(table_view const& probe,
                                            rmm::cuda_stream_view stream = cudf::get_default_stream()) const;
  [[nodiscard]] std::size_t left_join_size(table_view const& probe,
                                           rmm::cuda_stream_view stream = cudf::get_default_stream()) const;
  [[nodiscard]] std::size_t full_join_size(table_view const& probe,
                                           rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                                           rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref()) const;

  // Chunked joins for outputs beyond size_type (hash_join.hpp:259-440): the match context holds the number of
  // matches of every left row (left / full: at least 1), from which the caller cuts the left table into chunks whose
  // output fits; partitioned_*_join joins one chunk and returns indices into the WHOLE left table.  A full join is
  // its left-join chunks plus the unmatched right rows, appended by finalize_partitioned_full_join.
  [[nodiscard]] join_match_context inner_join_match_context(
    table_view const& left, rmm::cuda_stream_view stream = cudf::get_default_stream(),
    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref()) const;
  [[nodiscard]] join_match_context left_join_match_context(
    table_view const& left, rmm::cuda_stream_view stream = cudf::get_default_stream(),
    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref()) const;
  [[nodiscard]] join_match_context full_join_match_context(
    table_view const& left, rmm::cuda_stream_view stream = cudf::get_default_stream(),
    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref()) const;
  [[nodiscard]] join_result partitioned_inner_join(
    join_partition_context const& context, rmm::cuda_stream_view stream = cudf::get_default_stream(),
    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref()) const;
  [[nodiscard]] join_result partitioned_left_join(
    join_partition_context const& context, rmm::cuda_stream_view stream = cudf::get_default_stream(),
    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref()) const;
  [[nodiscard]] join_result partitioned_full_join(
    join_partition_context const& context, rmm::cuda_stream_view stream = cudf::get_default_stream(),
    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref()) const;
  [[nodiscard]] static join_result finalize_partitioned_full_join(
    host_span<device_span<size_type const> const> left_partials, host_span<device_span<size_type const> const> right_partials,
    size_type left_table_num_rows, size_type right_table_num_rows, rmm::cuda_stream_view stream = cudf::get_default_stream(),
    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

 private:
  std::unique_ptr<detail::hash_join_impl const> _impl;
};

}  // namespace cudf
