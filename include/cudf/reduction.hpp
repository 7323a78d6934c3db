// cudf/reduction.hpp -- column reduce / scan (reference: cpp/include/cudf/reduction.hpp:60-65,
// 124-130,229-235; impl cpp/src/reductions/reductions.cpp:484-507, scan/scan.cpp:13-54).
#pragma once
#include <cudf/aggregation.hpp>
#include <cudf/column/column.hpp>
#include <cudf/scalar/scalar.hpp>
#include <cudf/types.hpp>

#include <functional>
#include <memory>
#include <optional>
#include <utility>

namespace cudf {

// column -> scalar of `output_type`; nulls are skipped; the result is invalid when there is no
// valid element.  SUM/PRODUCT compute in output_type (INT64, UINT64 or FLOAT64 here), MIN/MAX
// require output_type == col.type().  MEAN (reductions/mean.cu): floating output, sum / valid count; COUNT_VALID / COUNT_ALL
// (reductions/count.cpp): size - null_count / size in any numeric non-bool type, always valid; ANY / ALL (reductions/any.cu,
// all.cu): BOOL8 output, elements cast to bool, false / true (valid) when there is no valid element.
std::unique_ptr<scalar> reduce(column_view const& col, reduce_aggregation const& agg, data_type output_type,
                               rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                               rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

// The same with an initial value (reduction.hpp:124-130): result = op(init, reduce(col)), the initial value cast to
// output_type first (simple.cuh:56-66).  `init` must have the column's type (cudf::data_type_error otherwise); an invalid
// (null) initial value or a column without a valid row gives an invalid result (simple.cuh:80-83); SUM, PRODUCT, MIN, MAX, ANY
// and ALL take one here (std::invalid_argument otherwise, reductions.cpp:492-499).
std::unique_ptr<scalar> reduce(column_view const& col, reduce_aggregation const& agg, data_type output_type,
                               std::optional<std::reference_wrapper<scalar const>> init,
                               rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                               rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

// minimum and maximum of a column in one call (reduction.hpp:247-250; src/reductions/minmax.cu): scalars of the column's type, nulls
// skipped, both invalid when there is no valid element
std::pair<std::unique_ptr<scalar>, std::unique_ptr<scalar>> minmax(
  column_view const& col, rmm::cuda_stream_view stream = cudf::get_default_stream(),
  rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

// prefix SUM / MIN / MAX / PRODUCT; output type == input type (integers wrap).
// null_policy::EXCLUDE: nulls are skipped and stay null; INCLUDE: the first null poisons the rest.
std::unique_ptr<column> scan(column_view const& input, scan_aggregation const& agg, scan_type inclusive,
                             null_policy null_handling         = null_policy::EXCLUDE,
                             rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                             rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

}  // namespace cudf
