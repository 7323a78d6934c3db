// Sort-based groupby over the C ABI: the sort helper, sort_aggregate (the path cudf::groupby::aggregate takes for
// pre-sorted keys, null keys that are to be kept, and aggregations the hash kernels do not implement), get_groups,
// shift and replace_nulls.
// reference: cpp/src/groupby/sort/{sort_helper.cu:37-260, aggregate.cpp:94-142,276-301,879-903,
// group_single_pass_reduction_util.cuh:133-200, group_count.cu:25-89, group_replace_nulls.cu},
// cpp/src/groupby/groupby.cu:54-71,261-362.
#include "common.hpp"
#include "sort_helper.hpp"

#include <cudf/column/column_factories.hpp>
#include <cudf/copying.hpp>
#include <cudf/groupby.hpp>
#include <cudf/null_mask.hpp>
#include <cudf/sorting.hpp>

#include <algorithm>

namespace cudf {
namespace groupby {
namespace sort_impl {

sort_groupby_helper::sort_groupby_helper(table_view const& keys, null_policy include_null_keys, sorted keys_pre_sorted,
                                         std::vector<null_order> const& null_precedence)
  : _keys{keys}, _keys_pre_sorted{keys_pre_sorted}, _include_null_keys{include_null_keys}, _null_precedence{null_precedence}
{
  if (keys_pre_sorted == sorted::YES && include_null_keys == null_policy::EXCLUDE && cudf::has_nulls(keys))
    _keys_pre_sorted = sorted::NO;
}

size_type sort_groupby_helper::num_keys(rmm::cuda_stream_view stream)
{
  if (_num_keys > -1) return _num_keys;
  if (_include_null_keys == null_policy::EXCLUDE && cudf::has_nulls(_keys)) {
    auto [mask, nulls] = cudf::bitmask_and(_keys, stream);
    _num_keys          = _keys.num_rows() - nulls;
  } else {
    _num_keys = _keys.num_rows();
  }
  return _num_keys;
}

column_view sort_groupby_helper::key_sort_order(rmm::cuda_stream_view stream)
{
  auto sliced = [&] {
    return column_view{data_type{type_id::INT32}, num_keys(stream), _order->view().head<void>(), nullptr, 0};
  };
  if (_order) return sliced();
  auto const n = _keys.num_rows();
  if (_keys_pre_sorted == sorted::YES) {
    _order = make_fixed_width_column(data_type{type_id::INT32}, n, mask_state::UNALLOCATED, stream);
    if (n) cudf::detail::gx_check(gx_sequence_i32(_order->mutable_view().head<int32_t>(), n, 0, cudf::detail::gxs(stream)), "sequence");
    return sliced();
  }
  auto precedence = _null_precedence.empty() ? std::vector<null_order>(_keys.num_columns(), null_order::AFTER) : _null_precedence;
  if (_include_null_keys == null_policy::INCLUDE || !cudf::has_nulls(_keys)) {  // SQL style
    _order = cudf::stable_sorted_order(_keys, {}, precedence, stream);
  } else {  // Pandas style: a leading all-zero column that is null where any key is null sends those rows to the end
    auto [mask, nulls] = cudf::bitmask_and(_keys, stream);
    auto flag          = make_fixed_width_column(data_type{type_id::INT8}, n, mask_state::UNALLOCATED, stream);
    if (n) CUDF_CUDA_TRY(hipMemsetAsync(flag->mutable_view().head<void>(), 0, static_cast<std::size_t>(n), stream.value()));
    flag->set_null_mask(std::move(mask), nulls);
    std::vector<column_view> cols{flag->view()};
    for (auto const& c : _keys) cols.push_back(c);
    precedence.insert(precedence.begin(), null_order::AFTER);
    _order = cudf::stable_sorted_order(table_view{cols}, {}, precedence, stream);
  }
  return sliced();
}

void sort_groupby_helper::build_groups(rmm::cuda_stream_view stream)
{
  if (_num_groups > -1) return;
  auto const n = num_keys(stream);
  _offsets     = rmm::device_buffer{(static_cast<std::size_t>(n) + 1) * sizeof(int32_t), stream};
  _labels      = rmm::device_buffer{std::max<std::size_t>(1, n) * sizeof(int32_t), stream};
  _sizes       = rmm::device_buffer{std::max<std::size_t>(1, n) * sizeof(int32_t), stream};
  _heads       = rmm::device_buffer{std::max<std::size_t>(1, n), stream};
  if (n == 0) {
    CUDF_CUDA_TRY(hipMemsetAsync(_offsets.data(), 0, sizeof(int32_t), stream.value()));
    _num_groups = 0;
    return;
  }
  auto const order = key_sort_order(stream);
  // rows differ when ANY key column differs: one pass per column, ORed into the head flags
  std::vector<rmm::device_buffer> holders(static_cast<std::size_t>(_keys.num_columns()));
  for (size_type i = 0; i < _keys.num_columns(); ++i) {
    auto const& c    = _keys.column(i);
    auto const* mask = c.has_nulls() ? cudf::detail::rebased_mask(c, holders[i], stream) : nullptr;
    cudf::detail::gx_check(gx_group_heads(cudf::detail::gx_type(c.type()), cudf::detail::row0(c), mask,
                                          is_presorted() ? nullptr : order.head<int32_t>(), n, i > 0 ? 1 : 0,
                                          static_cast<uint8_t*>(_heads.data()), cudf::detail::gxs(stream)),
                           "groupby group heads");
  }
  rmm::device_buffer ng{sizeof(int64_t), stream};
  cudf::detail::run_with_scratch(
    [&](void* t, std::size_t* b) {
      return gx_group_offsets(static_cast<uint8_t const*>(_heads.data()), n, static_cast<int32_t*>(_labels.data()),
                              static_cast<int32_t*>(_offsets.data()), static_cast<int32_t*>(_sizes.data()),
                              static_cast<int64_t*>(ng.data()), t, b, cudf::detail::gxs(stream));
    },
    "groupby group offsets", stream);
  _num_groups = static_cast<size_type>(cudf::detail::read_i64(static_cast<int64_t const*>(ng.data()), stream));
}

size_type sort_groupby_helper::num_groups(rmm::cuda_stream_view stream)
{
  build_groups(stream);
  return _num_groups;
}
int32_t const* sort_groupby_helper::group_offsets(rmm::cuda_stream_view stream)
{
  build_groups(stream);
  return static_cast<int32_t const*>(_offsets.data());
}
int32_t const* sort_groupby_helper::group_labels(rmm::cuda_stream_view stream)
{
  build_groups(stream);
  return static_cast<int32_t const*>(_labels.data());
}
int32_t const* sort_groupby_helper::group_sizes(rmm::cuda_stream_view stream)
{
  build_groups(stream);
  return static_cast<int32_t const*>(_sizes.data());
}
uint8_t const* sort_groupby_helper::group_heads(rmm::cuda_stream_view stream)
{
  build_groups(stream);
  return static_cast<uint8_t const*>(_heads.data());
}
std::vector<size_type> sort_groupby_helper::group_offsets_host(rmm::cuda_stream_view stream)
{
  std::vector<size_type> h(static_cast<std::size_t>(num_groups(stream)) + 1);
  CUDF_CUDA_TRY(hipMemcpyAsync(h.data(), _offsets.data(), h.size() * sizeof(size_type), hipMemcpyDeviceToHost, stream.value()));
  stream.synchronize();
  return h;
}

std::unique_ptr<table> sort_groupby_helper::unique_keys(rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  auto const g = num_groups(stream);
  column_view starts{data_type{type_id::INT32}, g, _offsets.data(), nullptr, 0};  // sorted position of each group's first row
  if (is_presorted()) return cudf::gather(_keys, starts, out_of_bounds_policy::DONT_CHECK, stream, mr);
  auto rows = cudf::gather(table_view{{key_sort_order(stream)}}, starts, out_of_bounds_policy::DONT_CHECK, stream);
  return cudf::gather(_keys, rows->get_column(0).view(), out_of_bounds_policy::DONT_CHECK, stream, mr);
}

std::unique_ptr<table> sort_groupby_helper::sorted_keys(rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  return cudf::gather(_keys, key_sort_order(stream), out_of_bounds_policy::DONT_CHECK, stream, mr);
}

std::unique_ptr<column> sort_groupby_helper::grouped_values(column_view const& values, rmm::cuda_stream_view stream,
                                                            rmm::device_async_resource_ref mr)
{
  auto t = cudf::gather(table_view{{values}}, key_sort_order(stream), out_of_bounds_policy::DONT_CHECK, stream, mr);
  return std::move(t->release().front());
}

}  // namespace sort_impl

namespace {

data_type sum_type_of(data_type v) { return is_floating_point(v) ? v : data_type{type_id::INT64}; }

// validity of a per-group result from the number of valid values that went into it
void mask_from_counts(column& c, int32_t const* counts, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  auto const g = c.size();
  if (g == 0) return;
  rmm::device_buffer mask = create_null_mask(g, mask_state::ALL_VALID, stream, mr);
  rmm::device_buffer cnt{sizeof(int64_t), stream};
  detail::gx_check(gx_valid_from_counts(counts, g, static_cast<uint32_t*>(mask.data()), static_cast<int64_t*>(cnt.data()),
                                        detail::gxs(stream)),
                   "groupby validity");
  auto const nulls = static_cast<size_type>(detail::read_i64(static_cast<int64_t const*>(cnt.data()), stream));
  if (nulls > 0) c.set_null_mask(std::move(mask), nulls);
}

}  // namespace

sort_impl::sort_groupby_helper& groupby::helper()
{
  if (!_helper) _helper = std::make_unique<sort_impl::sort_groupby_helper>(_keys, _include_null_keys, _keys_are_sorted, _null_precedence);
  return *_helper;
}

// aggregate.cpp:94-142 (dispatch per kind), 879-903 (sort_aggregate): results in sorted-key order
std::pair<std::unique_ptr<table>, std::vector<aggregation_result>> groupby::sort_aggregate(
  std::span<aggregation_request const> requests, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  auto& h      = helper();
  auto const n = h.num_keys(stream);
  auto const g = h.num_groups(stream);
  std::vector<aggregation_result> results(requests.size());
  auto const* heads  = h.group_heads(stream);
  auto const* labels = h.group_labels(stream);
  for (std::size_t i = 0; i < requests.size(); ++i) {
    auto const& req = requests[i];
    // the values in key order (a pre-sorted table is used as it is)
    std::unique_ptr<column> owner;
    column_view sv = req.values;
    if (!h.is_presorted()) {
      owner = h.grouped_values(req.values, stream, cudf::get_current_device_resource_ref());
      sv    = owner->view();
    }
    rmm::device_buffer vh;
    auto const* vmask = sv.has_nulls() ? detail::rebased_mask(sv, vh, stream) : nullptr;
    auto counts       = make_fixed_width_column(data_type{type_id::INT32}, g, mask_state::UNALLOCATED, stream);
    bool have_counts  = false;
    auto seg_reduce   = [&](column_view const& vals, bitmask_type const* mask, int op, data_type out_type, bool want_counts) {
      auto out = make_fixed_width_column(out_type, g, mask_state::UNALLOCATED, stream, mr);
      if (n > 0)
        detail::run_with_scratch(
          [&](void* t, std::size_t* b) {
            return gx_segmented_reduce(detail::gx_type(vals.type()), detail::row0(vals), mask, heads, labels, n, op,
                                       op == GX_OP_COUNT_VALID ? nullptr : out->mutable_view().head<void>(),
                                       want_counts ? counts->mutable_view().head<int32_t>() : nullptr, t, b, detail::gxs(stream));
          },
          "groupby segmented reduce", stream);
      return out;
    };
    auto ensure_counts = [&] {
      if (have_counts) return;
      (void)seg_reduce(sv, vmask, GX_OP_COUNT_VALID, data_type{type_id::INT32}, true);
      have_counts = true;
    };
    for (auto const& agg : req.aggregations) {
      std::unique_ptr<column> out;
      switch (agg->kind) {
        case aggregation::SUM:
        case aggregation::PRODUCT:
        case aggregation::MIN:
        case aggregation::MAX: {
          int const op = agg->kind == aggregation::SUM ? GX_OP_SUM
                         : agg->kind == aggregation::PRODUCT ? GX_OP_PRODUCT
                         : agg->kind == aggregation::MIN ? GX_OP_MIN : GX_OP_MAX;
          bool const widen = agg->kind == aggregation::SUM || agg->kind == aggregation::PRODUCT;
          out         = seg_reduce(sv, vmask, op, widen ? sum_type_of(sv.type()) : sv.type(), true);
          have_counts = true;
          mask_from_counts(*out, counts->view().head<int32_t>(), stream, mr);
          break;
        }
        case aggregation::COUNT_VALID: {
          ensure_counts();
          out = std::make_unique<column>(counts->view(), stream, mr);
          break;
        }
        case aggregation::COUNT_ALL: {
          column_view sizes{data_type{type_id::INT32}, g, h.group_sizes(stream), nullptr, 0};
          out = std::make_unique<column>(sizes, stream, mr);
          break;
        }
        case aggregation::MEAN: {
          auto sum = seg_reduce(sv, vmask, GX_OP_SUM, sum_type_of(sv.type()), true);
          have_counts = true;
          out = make_fixed_width_column(data_type{type_id::FLOAT64}, g, mask_state::UNALLOCATED, stream, mr);
          if (g > 0)
            detail::gx_check(gx_mean_from_sum(detail::gx_type(sum->type()), sum->view().head<void>(), counts->view().head<int32_t>(), g,
                                              out->mutable_view().head<double>(), detail::gxs(stream)),
                             "groupby mean");
          mask_from_counts(*out, counts->view().head<int32_t>(), stream, mr);
          break;
        }
        case aggregation::SUM_OF_SQUARES:
        case aggregation::M2:
        case aggregation::VARIANCE:
        case aggregation::STD: {
          auto sq = make_fixed_width_column(sum_type_of(sv.type()), n, mask_state::UNALLOCATED, stream);
          if (n > 0)
            detail::gx_check(gx_square(detail::gx_type(sv.type()), detail::row0(sv), n, sq->mutable_view().head<void>(), detail::gxs(stream)),
                             "groupby sum of squares");
          column_view sqv{sq->type(), n, sq->view().head<void>(), vmask, vmask ? sv.null_count() : 0};
          auto ssq    = seg_reduce(sqv, vmask, GX_OP_SUM, sq->type(), true);
          have_counts = true;
          if (agg->kind == aggregation::SUM_OF_SQUARES) {
            out = std::move(ssq);
            mask_from_counts(*out, counts->view().head<int32_t>(), stream, mr);
            break;
          }
          auto sum = seg_reduce(sv, vmask, GX_OP_SUM, sum_type_of(sv.type()), false);
          out      = make_fixed_width_column(data_type{type_id::FLOAT64}, g, mask_state::UNALLOCATED, stream, mr);
          if (g > 0) {
            rmm::device_buffer mask = create_null_mask(g, mask_state::ALL_VALID, stream, mr);
            rmm::device_buffer cnt{sizeof(int64_t), stream};
            int ddof = 1;
            if (auto const* sva = dynamic_cast<cudf::detail::std_var_aggregation const*>(agg.get())) ddof = sva->_ddof;
            int const mode = agg->kind == aggregation::M2 ? 0 : (agg->kind == aggregation::VARIANCE ? 1 : 2);
            detail::gx_check(gx_var_from_sums(detail::gx_type(sum->type()), ssq->view().head<void>(), sum->view().head<void>(),
                                              counts->view().head<int32_t>(), g, ddof, mode, out->mutable_view().head<double>(),
                                              static_cast<uint32_t*>(mask.data()), static_cast<int64_t*>(cnt.data()),
                                              detail::gxs(stream)),
                             "groupby variance");
            auto const nulls = static_cast<size_type>(detail::read_i64(static_cast<int64_t const*>(cnt.data()), stream));
            if (nulls > 0) out->set_null_mask(std::move(mask), nulls);
          }
          break;
        }
        case aggregation::ARGMIN:
        case aggregation::ARGMAX: {
          // the group's extreme, then the first sorted row holding it, mapped back to a row of the input
          auto ext    = seg_reduce(sv, vmask, agg->kind == aggregation::ARGMIN ? GX_OP_MIN : GX_OP_MAX, sv.type(), true);
          have_counts = true;
          auto pos    = make_fixed_width_column(data_type{type_id::INT32}, g, mask_state::UNALLOCATED, stream);
          if (g > 0)
            detail::gx_check(gx_groupby_arg_select(detail::gx_type(sv.type()), detail::row0(sv), vmask, labels, n, ext->view().head<void>(),
                                                   g, pos->mutable_view().head<int32_t>(), detail::gxs(stream)),
                             "groupby argmin/argmax");
          if (h.is_presorted() || g == 0) {
            out = std::make_unique<column>(pos->view(), stream, mr);
          } else {
            auto t = cudf::gather(table_view{{h.key_sort_order(stream)}}, pos->view(), out_of_bounds_policy::NULLIFY, stream, mr);
            out    = std::move(t->release().front());
            out->set_null_mask(rmm::device_buffer{}, 0);
          }
          mask_from_counts(*out, counts->view().head<int32_t>(), stream, mr);
          break;
        }
        case aggregation::NTH_ELEMENT: {
          // sort/group_nth_element.cu: value at (group start + n), or (group end + n) for negative n; null when the
          // group is shorter.  Index arithmetic on the host (G + 1 offsets), then one gather.
          auto const* nth = dynamic_cast<cudf::detail::nth_element_aggregation const*>(agg.get());
          CUDF_EXPECTS(nth != nullptr && nth->_null_handling == null_policy::INCLUDE,
                       "groupby NTH_ELEMENT: only null_policy::INCLUDE is implemented");
          auto const off = h.group_offsets_host(stream);
          std::vector<size_type> idx(static_cast<std::size_t>(g));
          for (size_type k = 0; k < g; ++k) {
            auto const size = off[k + 1] - off[k];
            auto const j    = nth->_n >= 0 ? nth->_n : size + nth->_n;
            idx[k]          = (j >= 0 && j < size) ? off[k] + j : -1;  // -1: out of bounds -> null
          }
          rmm::device_buffer didx{idx.data(), idx.size() * sizeof(size_type), stream};
          column_view map{data_type{type_id::INT32}, g, didx.data(), nullptr, 0};
          auto t = cudf::gather(table_view{{sv}}, map, out_of_bounds_policy::NULLIFY, stream, mr);
          stream.synchronize();  // idx / didx
          out = std::move(t->release().front());
          break;
        }
        default:
          CUDF_FAIL("groupby aggregation kind not implemented (SUM, PRODUCT, MIN, MAX, COUNT, MEAN, SUM_OF_SQUARES, M2, VARIANCE, "
                    "STD, ARGMIN, ARGMAX, NTH_ELEMENT are)");
      }
      results[i].results.emplace_back(std::move(out));
    }
  }
  return {h.unique_keys(stream, mr), std::move(results)};
}

// groupby.cu:261-283
groupby::groups groupby::get_groups(table_view values, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  auto& h = helper();
  groups out;
  out.keys    = h.sorted_keys(stream, mr);
  out.offsets = h.group_offsets_host(stream);
  if (values.num_columns() > 0) {
    CUDF_EXPECTS(values.num_rows() == _keys.num_rows(), "Size mismatch between group values and keys.", std::invalid_argument);
    out.values = cudf::gather(values, h.key_sort_order(stream), out_of_bounds_policy::DONT_CHECK, stream, mr);
  }
  return out;
}

// groupby.cu:306-346: every column shifted inside its groups, keys returned in sorted order
std::pair<std::unique_ptr<table>, std::unique_ptr<table>> groupby::shift(
  table_view const& values, std::span<size_type const> offsets, std::vector<std::reference_wrapper<scalar const>> const& fill_values,
  rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(values.num_columns() == static_cast<size_type>(fill_values.size()), "Mismatch number of fill_values and columns.");
  CUDF_EXPECTS(values.num_columns() == static_cast<size_type>(offsets.size()), "Mismatch number of offsets and columns.");
  for (size_type i = 0; i < values.num_columns(); ++i)
    CUDF_EXPECTS(values.column(i).type() == fill_values[i].get().type(), "values and fill_value should have the same type.",
                 cudf::data_type_error);
  CUDF_EXPECTS(values.num_columns() == 0 || values.num_rows() == _keys.num_rows(), "Size mismatch between group values and keys.",
               std::invalid_argument);
  auto& h            = helper();
  auto const n       = h.num_keys(stream);
  auto const* labels = h.group_labels(stream);
  std::vector<std::unique_ptr<column>> results;
  for (size_type i = 0; i < values.num_columns(); ++i) {
    auto gv        = h.grouped_values(values.column(i), stream, cudf::get_current_device_resource_ref());
    auto const& sc = fill_values[i].get();
    bool const fv  = sc.is_valid(stream);
    uint64_t bits  = 0;
    auto const w   = size_of(gv->type());
    if (fv) {  // the scalar's value bytes (device -> host)
      auto const* src = sc.device_value_ptr();
      CUDF_EXPECTS(src != nullptr, "groupby::shift: fill values must be fixed-width scalars", cudf::data_type_error);
      CUDF_CUDA_TRY(hipMemcpyAsync(&bits, src, w, hipMemcpyDeviceToHost, stream.value()));
      stream.synchronize();
    }
    auto out = make_fixed_width_column(gv->type(), n, mask_state::ALL_VALID, stream, mr);
    rmm::device_buffer vh;
    auto const view   = gv->view();
    auto const* vmask = view.has_nulls() ? detail::rebased_mask(view, vh, stream) : nullptr;
    if (n > 0)
      detail::gx_check(gx_segmented_shift(static_cast<int>(w), view.head<void>(), vmask, labels, n, offsets[i], bits, fv ? 1 : 0,
                                          out->mutable_view().head<void>(), out->mutable_view().null_mask(), detail::gxs(stream)),
                       "groupby shift");
    out->set_null_count(n > 0 ? cudf::null_count(out->view().null_mask(), 0, n, stream) : 0);
    results.emplace_back(std::move(out));
  }
  return {h.sorted_keys(stream, mr), std::make_unique<table>(std::move(results))};
}

// groupby.cu:285-321, sort/group_replace_nulls.cu
std::pair<std::unique_ptr<table>, std::unique_ptr<table>> groupby::replace_nulls(table_view const& values,
                                                                                 std::span<cudf::replace_policy const> replace_policies,
                                                                                 rmm::cuda_stream_view stream,
                                                                                 rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(_keys.num_rows() == values.num_rows(), "Size mismatch between group labels and value.");
  CUDF_EXPECTS(static_cast<size_type>(replace_policies.size()) == values.num_columns(),
               "Size mismatch between num_columns and replace_policies.");
  auto& h           = helper();
  auto const n      = h.num_keys(stream);
  auto const* heads = h.group_heads(stream);
  std::vector<std::unique_ptr<column>> results;
  for (size_type i = 0; i < values.num_columns(); ++i) {
    auto gv = h.grouped_values(values.column(i), stream, mr);
    if (!values.column(i).nullable() || n == 0) {
      results.emplace_back(std::move(gv));
      continue;
    }
    auto const view = gv->view();
    auto out        = make_fixed_width_column(gv->type(), n, mask_state::ALL_VALID, stream, mr);
    detail::run_with_scratch(
      [&](void* t, std::size_t* b) {
        return gx_segmented_fill_nulls(static_cast<int>(size_of(view.type())), view.head<void>(), view.null_mask(), heads, n,
                                       replace_policies[i] == replace_policy::FOLLOWING ? 1 : 0, out->mutable_view().head<void>(),
                                       out->mutable_view().null_mask(), t, b, detail::gxs(stream));
      },
      "groupby replace_nulls", stream);
    out->set_null_count(cudf::null_count(out->view().null_mask(), 0, n, stream));
    results.emplace_back(std::move(out));
  }
  return {h.sorted_keys(stream, mr), std::make_unique<table>(std::move(results))};
}

}  // namespace groupby
}  // namespace cudf
