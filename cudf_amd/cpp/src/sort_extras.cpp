// cudf::rank, cudf::top_k / top_k_order and the segmented sorts over the C ABI: all of them are a sorted_order plus one
// streaming kernel (group boundaries in sorted order, rank scatter, segment ids).
// reference: cpp/src/sort/rank.cu:59-369, cpp/src/sort/top_k.cu:118-165, cpp/src/sort/segmented_sort_impl.cuh:152-324.
#include "common.hpp"

#include <cudf/column/column_factories.hpp>
#include <cudf/copying.hpp>
#include <cudf/null_mask.hpp>
#include <cudf/sorting.hpp>

namespace cudf {

std::unique_ptr<column> rank(column_view const& input, rank_method method, order column_order, null_policy null_handling,
                             null_order null_precedence, bool percentage, rmm::cuda_stream_view stream,
                             rmm::device_async_resource_ref mr)
{
  auto const n        = input.size();
  bool const as_f64   = percentage || method == rank_method::AVERAGE;
  auto const out_type = data_type{as_f64 ? type_id::FLOAT64 : type_id::INT32};
  auto out            = make_fixed_width_column(out_type, n, mask_state::UNALLOCATED, stream, mr);
  if (n == 0) return out;
  // na_option = keep: null rows stay null (rank.cu:262-272)
  if (null_handling == null_policy::EXCLUDE && input.nullable()) {
    rmm::device_buffer holder;
    auto const* m = detail::rebased_mask(input, holder, stream);
    rmm::device_buffer mask{m, bitmask_allocation_size_bytes(n), stream, mr};
    out->set_null_mask(std::move(mask), input.null_count());
  }
  auto order_col   = cudf::stable_sorted_order(table_view{{input}}, {column_order}, {null_precedence}, stream);
  auto const* ord  = order_col->view().head<int32_t>();
  auto const count = null_handling == null_policy::EXCLUDE ? n - input.null_count() : n;  // rows that are ranked
  rmm::device_buffer heads, labels, offsets;
  double scale = percentage ? static_cast<double>(count) : 0.0;
  if (method != rank_method::FIRST) {  // equal values share a rank: runs of equal rows in sorted order
    heads   = rmm::device_buffer{static_cast<std::size_t>(n), stream};
    labels  = rmm::device_buffer{static_cast<std::size_t>(n) * 4, stream};
    offsets = rmm::device_buffer{(static_cast<std::size_t>(n) + 1) * 4, stream};
    rmm::device_buffer holder;
    auto const* mask = input.has_nulls() ? detail::rebased_mask(input, holder, stream) : nullptr;
    detail::gx_check(gx_group_heads(detail::gx_type(input.type()), detail::row0(input), mask, ord, n, 0,
                                    static_cast<uint8_t*>(heads.data()), detail::gxs(stream)),
                     "rank: runs of equal values");
    rmm::device_buffer ng{sizeof(int64_t), stream};
    detail::run_with_scratch(
      [&](void* t, std::size_t* b) {
        return gx_group_offsets(static_cast<uint8_t const*>(heads.data()), n, static_cast<int32_t*>(labels.data()),
                                static_cast<int32_t*>(offsets.data()), nullptr, static_cast<int64_t*>(ng.data()), t, b,
                                detail::gxs(stream));
      },
      "rank: group offsets", stream);
    if (percentage && method == rank_method::DENSE) {  // r / dense rank of the last ranked row (rank.cu:330-345)
      int32_t last = 0;
      if (count > 0) {
        CUDF_CUDA_TRY(hipMemcpyAsync(&last, static_cast<int32_t const*>(labels.data()) + (count - 1), sizeof(int32_t),
                                     hipMemcpyDeviceToHost, stream.value()));
        stream.synchronize();
      }
      scale = static_cast<double>(last + 1);
    }
  }
  detail::gx_check(gx_rank_from_groups(ord, static_cast<int32_t const*>(labels.data()), static_cast<int32_t const*>(offsets.data()), n,
                                       static_cast<int>(method), scale, 0, as_f64 ? nullptr : out->mutable_view().head<int32_t>(),
                                       as_f64 ? out->mutable_view().head<double>() : nullptr, detail::gxs(stream)),
                   "rank");
  return out;
}

std::unique_ptr<column> top_k_order(column_view const& col, size_type k, order topk_order, rmm::cuda_stream_view stream,
                                    rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(k >= 0, "k must be non-negative", std::invalid_argument);
  if (k == 0 || col.size() == 0) return make_empty_column(data_type{type_id::INT32});
  if (k >= col.size()) {  // top_k.cu:139-145: every row is in the top -- the identity, matching top_k's copy of the column
    auto all = make_fixed_width_column(data_type{type_id::INT32}, col.size(), mask_state::UNALLOCATED, stream, mr);
    detail::gx_check(gx_sequence_i32(all->mutable_view().head<int32_t>(), col.size(), 0, detail::gxs(stream)), "top_k_order");
    return all;
  }
  // nulls never make the top (top_k.cu:136-137)
  auto const nulls = topk_order == order::ASCENDING ? null_order::AFTER : null_order::BEFORE;
  auto indices     = cudf::stable_sorted_order(table_view{{col}}, {topk_order}, {nulls}, stream);
  auto const kk    = std::min(k, col.size());
  column_view first{data_type{type_id::INT32}, kk, indices->view().head<void>(), nullptr, 0};
  auto out = std::make_unique<column>(first, stream, mr);
  return out;
}

std::unique_ptr<column> top_k(column_view const& col, size_type k, order topk_order, rmm::cuda_stream_view stream,
                              rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(k >= 0, "k must be non-negative", std::invalid_argument);
  if (k == 0 || col.size() == 0) return make_empty_column(col.type());
  if (k >= col.size()) return std::make_unique<column>(col, stream, mr);
  auto idx = top_k_order(col, k, topk_order, stream, cudf::get_current_device_resource_ref());
  auto t   = cudf::gather(table_view{{col}}, idx->view(), out_of_bounds_policy::DONT_CHECK, stream, mr);
  return std::move(t->release().front());
}

namespace {
std::unique_ptr<column> segmented_order(table_view const& keys, column_view const& segment_offsets,
                                        std::vector<order> const& column_order, std::vector<null_order> const& null_precedence,
                                        rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  if (keys.num_rows() == 0 || keys.num_columns() == 0) return make_empty_column(data_type{type_id::INT32});
  CUDF_EXPECTS(segment_offsets.type().id() == type_id::INT32, "segment offsets should be size_type");
  if (!column_order.empty())
    CUDF_EXPECTS(static_cast<std::size_t>(keys.num_columns()) == column_order.size(),
                 "Mismatch between number of columns and column order.");
  if (!null_precedence.empty())
    CUDF_EXPECTS(static_cast<std::size_t>(keys.num_columns()) == null_precedence.size(),
                 "Mismatch between number of columns and null_precedence size.");
  // the segment id goes in front of the key columns: one lexicographic (stable) sort (segmented_sort_impl.cuh:265-293)
  auto ids = make_fixed_width_column(data_type{type_id::INT32}, keys.num_rows(), mask_state::UNALLOCATED, stream);
  detail::gx_check(gx_segment_ids(static_cast<int32_t const*>(detail::row0(segment_offsets)), segment_offsets.size(), keys.num_rows(),
                                  ids->mutable_view().head<int32_t>(), detail::gxs(stream)),
                   "segment ids");
  std::vector<column_view> cols{ids->view()};
  for (auto const& c : keys) cols.push_back(c);
  std::vector<order> ord;
  std::vector<null_order> prec;
  if (!column_order.empty()) {
    ord.push_back(order::ASCENDING);
    ord.insert(ord.end(), column_order.begin(), column_order.end());
  }
  if (!null_precedence.empty()) {
    prec.push_back(null_order::AFTER);
    prec.insert(prec.end(), null_precedence.begin(), null_precedence.end());
  }
  auto out = cudf::stable_sorted_order(table_view{cols}, ord, prec, stream, mr);
  return out;
}
}  // namespace

std::unique_ptr<column> segmented_sorted_order(table_view const& keys, column_view const& segment_offsets,
                                               std::vector<order> const& column_order, std::vector<null_order> const& null_precedence,
                                               rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  return segmented_order(keys, segment_offsets, column_order, null_precedence, stream, mr);
}
std::unique_ptr<column> stable_segmented_sorted_order(table_view const& keys, column_view const& segment_offsets,
                                                      std::vector<order> const& column_order,
                                                      std::vector<null_order> const& null_precedence, rmm::cuda_stream_view stream,
                                                      rmm::device_async_resource_ref mr)
{
  return segmented_order(keys, segment_offsets, column_order, null_precedence, stream, mr);
}
std::unique_ptr<table> segmented_sort_by_key(table_view const& values, table_view const& keys, column_view const& segment_offsets,
                                             std::vector<order> const& column_order, std::vector<null_order> const& null_precedence,
                                             rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(values.num_rows() == keys.num_rows(), "Mismatch in number of rows for values and keys");
  auto ord = segmented_order(keys, segment_offsets, column_order, null_precedence, stream, cudf::get_current_device_resource_ref());
  return cudf::gather(values, ord->view(), out_of_bounds_policy::DONT_CHECK, stream, mr);
}
std::unique_ptr<table> stable_segmented_sort_by_key(table_view const& values, table_view const& keys,
                                                    column_view const& segment_offsets, std::vector<order> const& column_order,
                                                    std::vector<null_order> const& null_precedence, rmm::cuda_stream_view stream,
                                                    rmm::device_async_resource_ref mr)
{
  return segmented_sort_by_key(values, keys, segment_offsets, column_order, null_precedence, stream, mr);
}

}  // namespace cudf
