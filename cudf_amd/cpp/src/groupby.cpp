// cudf::groupby::groupby::aggregate / scan over the C ABI.
// reference: cpp/src/groupby/groupby.cu:40-71,220-259; hash path cpp/src/groupby/hash/*;
// scan path cpp/src/groupby/sort/{scan.cpp:214-238, sort_helper.cu:73-162, group_scan_util.cuh:77-133}.
#include "common.hpp"
#include "row_encoding.hpp"
#include "sort_helper.hpp"

#include <optional>
#include <cudf/column/column_factories.hpp>
#include <cudf/copying.hpp>
#include <cudf/groupby.hpp>
#include <cudf/sorting.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace cudf {
namespace groupby {
namespace {

struct hash_agg_out {
  std::unique_ptr<column> keys, sum, count_valid, count_all;
};

data_type sum_type(data_type v)
{
  // SUM target types (cpp/include/cudf/detail/aggregation/aggregation.hpp:949-970): integers -> INT64,
  // floats keep their type
  return is_floating_point(v) ? v : data_type{type_id::INT64};
}

// one gx_groupby_sum_count call, retried with a larger table when the group estimate was too small
hash_agg_out hash_aggregate(column_view const& keys, column_view const& vals, rmm::cuda_stream_view stream,
                            rmm::device_async_resource_ref mr)
{
  auto const n = keys.size();
  rmm::device_buffer kh, vh;
  auto const* kmask = keys.has_nulls() ? detail::rebased_mask(keys, kh, stream) : nullptr;
  auto const* vmask = vals.has_nulls() ? detail::rebased_mask(vals, vh, stream) : nullptr;
  int64_t max_groups = std::max<int64_t>(1, std::min<int64_t>(n, int64_t{1} << 20));
  rmm::device_buffer ng{sizeof(int64_t), stream};
  for (;;) {
    hash_agg_out o;
    auto const g = static_cast<size_type>(max_groups);
    o.keys        = make_fixed_width_column(keys.type(), g, mask_state::UNALLOCATED, stream, mr);
    o.sum         = make_fixed_width_column(sum_type(vals.type()), g, mask_state::UNALLOCATED, stream, mr);
    o.count_valid = make_fixed_width_column(data_type{type_id::INT32}, g, mask_state::UNALLOCATED, stream, mr);
    o.count_all   = make_fixed_width_column(data_type{type_id::INT32}, g, mask_state::UNALLOCATED, stream, mr);
    detail::run_with_scratch(
      [&](void* t, std::size_t* b) {
        return gx_groupby_sum_count(detail::gx_type(keys.type()), detail::row0(keys), kmask, detail::gx_type(vals.type()),
                                    detail::row0(vals), vmask, n, max_groups, o.keys->mutable_view().head<void>(),
                                    o.sum->mutable_view().head<void>(), o.count_valid->mutable_view().head<int32_t>(),
                                    o.count_all->mutable_view().head<int32_t>(), static_cast<int64_t*>(ng.data()), t, b,
                                    detail::gxs(stream));
      },
      "groupby aggregate", stream);
    auto const groups = detail::read_i64(static_cast<int64_t const*>(ng.data()), stream);
    if (groups >= 0 && groups <= max_groups) {
      // trim the columns to the group count (buffers keep their capacity)
      auto trim = [&](std::unique_ptr<column>& c) {
        auto type     = c->type();
        auto contents = c->release();
        c = std::make_unique<column>(type, static_cast<size_type>(groups), std::move(*contents.data), rmm::device_buffer{}, 0);
      };
      trim(o.keys);
      trim(o.sum);
      trim(o.count_valid);
      trim(o.count_all);
      return o;
    }
    CUDF_EXPECTS(max_groups < n, "groupby: group table overflow");
    max_groups = std::min<int64_t>(n, max_groups * 8);
  }
}

struct hash_minmax_out {
  std::unique_ptr<column> keys, mn, mx, count_valid;
};

hash_minmax_out hash_minmax(column_view const& keys, column_view const& vals, rmm::cuda_stream_view stream,
                            rmm::device_async_resource_ref mr)
{
  auto const n = keys.size();
  rmm::device_buffer kh, vh;
  auto const* kmask = keys.has_nulls() ? detail::rebased_mask(keys, kh, stream) : nullptr;
  auto const* vmask = vals.has_nulls() ? detail::rebased_mask(vals, vh, stream) : nullptr;
  int64_t max_groups = std::max<int64_t>(1, std::min<int64_t>(n, int64_t{1} << 20));
  rmm::device_buffer ng{sizeof(int64_t), stream};
  for (;;) {
    hash_minmax_out o;
    auto const g  = static_cast<size_type>(max_groups);
    o.keys        = make_fixed_width_column(keys.type(), g, mask_state::UNALLOCATED, stream, mr);
    o.mn          = make_fixed_width_column(vals.type(), g, mask_state::UNALLOCATED, stream, mr);
    o.mx          = make_fixed_width_column(vals.type(), g, mask_state::UNALLOCATED, stream, mr);
    o.count_valid = make_fixed_width_column(data_type{type_id::INT32}, g, mask_state::UNALLOCATED, stream, mr);
    detail::run_with_scratch(
      [&](void* t, std::size_t* b) {
        return gx_groupby_min_max(detail::gx_type(keys.type()), detail::row0(keys), kmask, detail::gx_type(vals.type()),
                                  detail::row0(vals), vmask, n, max_groups, o.keys->mutable_view().head<void>(),
                                  o.mn->mutable_view().head<void>(), o.mx->mutable_view().head<void>(),
                                  o.count_valid->mutable_view().head<int32_t>(), static_cast<int64_t*>(ng.data()), t, b,
                                  detail::gxs(stream));
      },
      "groupby min/max", stream);
    auto const groups = detail::read_i64(static_cast<int64_t const*>(ng.data()), stream);
    if (groups >= 0 && groups <= max_groups) {
      auto trim = [&](std::unique_ptr<column>& c) {
        auto type     = c->type();
        auto contents = c->release();
        c = std::make_unique<column>(type, static_cast<size_type>(groups), std::move(*contents.data), rmm::device_buffer{}, 0);
      };
      trim(o.keys);
      trim(o.mn);
      trim(o.mx);
      trim(o.count_valid);
      return o;
    }
    CUDF_EXPECTS(max_groups < n, "groupby: group table overflow");
    max_groups = std::min<int64_t>(n, max_groups * 8);
  }
}

// permute `c` by an INT32 map (same length)
std::unique_ptr<column> permute(column_view const& c, column_view const& map, rmm::cuda_stream_view stream,
                                rmm::device_async_resource_ref mr)
{
  auto t = cudf::gather(table_view{{c}}, map, out_of_bounds_policy::DONT_CHECK, stream, mr);
  return std::move(t->release().front());
}

}  // namespace

groupby::~groupby() = default;

groupby::groupby(table_view const& keys, null_policy null_handling, sorted keys_are_sorted,
                 std::vector<order> const& column_order, std::vector<null_order> const& null_precedence)
  : _keys{keys},
    _include_null_keys{null_handling},
    _keys_are_sorted{keys_are_sorted},
    _column_order{column_order},
    _null_precedence{null_precedence}
{
}

namespace {
struct row_key_collision {};  // two different key rows shared a 64-bit hash: thrown by decode_keys, caught by aggregate
}  // namespace

std::pair<std::unique_ptr<table>, std::vector<aggregation_result>> groupby::aggregate(
  std::span<aggregation_request const> requests, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  try {
    return aggregate_impl(requests, false, stream, mr);
  } catch (row_key_collision const&) {
    return aggregate_impl(requests, true, stream, mr);
  }
}

std::optional<std::pair<std::unique_ptr<table>, std::vector<aggregation_result>>> groupby::wide_aggregate(
  std::span<aggregation_request const> requests, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  auto const n  = _keys.num_rows();
  auto const nk = _keys.num_columns();
  if (nk < 2 || nk > 4 || n < (size_type{1} << 18) || requests.size() != 1) return std::nullopt;
  for (auto const& c : _keys)
    if ((c.type().id() != type_id::INT64 && c.type().id() != type_id::UINT64) || c.has_nulls()) return std::nullopt;
  auto const& req = requests[0];
  auto const vt   = req.values.type().id();
  if (req.values.has_nulls() || (vt != type_id::INT32 && vt != type_id::INT64 && vt != type_id::FLOAT32 && vt != type_id::FLOAT64))
    return std::nullopt;
  for (auto const& agg : req.aggregations)
    if (agg->kind != aggregation::SUM && agg->kind != aggregation::COUNT_VALID && agg->kind != aggregation::COUNT_ALL &&
        agg->kind != aggregation::MEAN)
      return std::nullopt;
  std::vector<void const*> kin(nk);
  for (size_type i = 0; i < nk; ++i) kin[i] = detail::row0(_keys.column(i));
  int64_t max_groups = std::max<int64_t>(1, std::min<int64_t>(n, int64_t{1} << 20));
  rmm::device_buffer ng{sizeof(int64_t), stream};
  for (;;) {
    auto const g = static_cast<size_type>(max_groups);
    std::vector<std::unique_ptr<column>> kout;
    std::vector<void*> kptr(nk);
    for (size_type i = 0; i < nk; ++i) {
      kout.emplace_back(make_fixed_width_column(_keys.column(i).type(), g, mask_state::UNALLOCATED, stream, mr));
      kptr[i] = kout.back()->mutable_view().head<void>();
    }
    auto sum = make_fixed_width_column(sum_type(req.values.type()), g, mask_state::UNALLOCATED, stream, mr);
    auto cnt = make_fixed_width_column(data_type{type_id::INT32}, g, mask_state::UNALLOCATED, stream, mr);
    detail::run_with_scratch(
      [&](void* t, std::size_t* b) {
        return gx_groupby_sum_count_wide(nk, kin.data(), detail::gx_type(req.values.type()), detail::row0(req.values), n, max_groups,
                                         kptr.data(), sum->mutable_view().head<void>(), cnt->mutable_view().head<int32_t>(),
                                         static_cast<int64_t*>(ng.data()), t, b, detail::gxs(stream));
      },
      "groupby aggregate (several key columns)", stream);
    auto const groups = detail::read_i64(static_cast<int64_t const*>(ng.data()), stream);
    if (groups == -2) return std::nullopt;  // skew / more groups per partition than an LDS table holds: the encoded path
    if (groups >= 0) {
      auto trim = [&](std::unique_ptr<column>& c) {
        auto type     = c->type();
        auto contents = c->release();
        c = std::make_unique<column>(type, static_cast<size_type>(groups), std::move(*contents.data), rmm::device_buffer{}, 0);
      };
      for (auto& c : kout) trim(c);
      trim(sum);
      trim(cnt);
      std::vector<aggregation_result> results(1);
      for (auto const& agg : req.aggregations) {
        switch (agg->kind) {
          case aggregation::SUM: results[0].results.emplace_back(std::make_unique<column>(sum->view(), stream, mr)); break;
          case aggregation::MEAN: {
            auto c = make_fixed_width_column(data_type{type_id::FLOAT64}, static_cast<size_type>(groups), mask_state::UNALLOCATED, stream, mr);
            detail::gx_check(gx_mean_from_sum(detail::gx_type(sum->type()), sum->view().head<void>(), cnt->view().head<int32_t>(), groups,
                                              c->mutable_view().head<double>(), detail::gxs(stream)),
                             "groupby mean");
            results[0].results.emplace_back(std::move(c));
            break;
          }
          default: results[0].results.emplace_back(std::make_unique<column>(cnt->view(), stream, mr)); break;  // COUNT_VALID == COUNT_ALL: no nulls
        }
      }
      return std::make_pair(std::make_unique<table>(std::move(kout)), std::move(results));
    }
    CUDF_EXPECTS(max_groups < n, "groupby: group table overflow");
    max_groups = std::min<int64_t>(n, max_groups * 8);
  }
}

std::pair<std::unique_ptr<table>, std::vector<aggregation_result>> groupby::aggregate_impl(
  std::span<aggregation_request const> requests, bool exact_keys, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(std::all_of(requests.begin(), requests.end(),
                           [this](auto const& r) { return r.values.size() == _keys.num_rows(); }),
               "Size mismatch between request values and groupby keys.", std::invalid_argument);
  CUDF_EXPECTS(_keys.num_columns() >= 1, "groupby needs at least one key column");
  // groupby.cu:54-71 dispatch_aggregation: hash-based only when the keys are not pre-sorted and every aggregation is one
  // the hash kernels implement; here also when no null key has to be KEPT (the hash tables drop null keys).
  bool sort_path = _keys_are_sorted == sorted::YES || (_include_null_keys == null_policy::INCLUDE && cudf::has_nulls(_keys));
  for (auto const& r : requests)
    for (auto const& agg : r.aggregations)
      sort_path = sort_path || agg->kind == aggregation::PRODUCT || agg->kind == aggregation::NTH_ELEMENT;
  if (sort_path && _keys.num_rows() > 0 && !requests.empty()) return sort_aggregate(requests, stream, mr);
  // Several 8-byte integer key columns, one request of SUM / COUNT / MEAN, no nulls: ONE partition pass with the rows compared
  // inside the LDS tables (gx_groupby_sum_count_wide) -- no row encoding, no certificate.  The device may decline (-2).
  if (!exact_keys) {
    if (auto r = wide_aggregate(requests, stream, mr)) return std::move(*r);
  }
  // One 32/64-bit integer key column goes to the hash kernels as it is; anything else (several columns, floats,
  // narrow types) becomes ONE 8-byte row key first (detail::row_keys: the packed values, or a 64-bit row hash whose
  // result is certified against the key columns), and the key columns come back from the distinct row keys.
  // exact_keys (the retry after a hash collision, or more than 8 key columns): dense INT32 row ids
  // (gx_dense_rank, one radix sort per column), aggregated by id, keys gathered through the first row of every id.
  auto const k0      = _keys.column(0).type().id();
  bool const encoded = !(_keys.num_columns() == 1 && (k0 == type_id::INT32 || k0 == type_id::INT64 ||
                                                      k0 == type_id::UINT32 || k0 == type_id::UINT64));
  detail::dense_rank_result enc;
  std::unique_ptr<detail::row_keys> rk;
  column_view keys = _keys.column(0);
  if (encoded && _keys.num_rows() > 0) {
    if (!exact_keys && _keys.num_columns() <= 8) {
      rk   = std::make_unique<detail::row_keys>(_keys, stream);
      keys = rk->view();
    } else {
      enc  = detail::dense_row_ids(_keys, stream);
      keys = enc.ids->view();
    }
  }
  // unique ids (in whatever order the hash pass produced) -> the key columns
  auto decode_keys = [&](std::unique_ptr<column> ids) {
    if (!encoded) {
      std::vector<std::unique_ptr<column>> kc;
      kc.emplace_back(std::move(ids));
      return std::make_unique<table>(std::move(kc));
    }
    if (_keys.num_rows() == 0 || ids->size() == 0) {
      std::vector<std::unique_ptr<column>> kc;
      for (auto const& c : _keys) kc.emplace_back(make_empty_column(c.type()));
      return std::make_unique<table>(std::move(kc));
    }
    if (rk) {
      auto kt = rk->key_columns(ids->view(), stream, mr);
      if (!kt) throw row_key_collision{};
      return kt;
    }
    auto rows = cudf::gather(table_view{{enc.rep->view()}}, ids->view(), out_of_bounds_policy::DONT_CHECK, stream);
    return cudf::gather(_keys, rows->get_column(0).view(), out_of_bounds_policy::DONT_CHECK, stream, mr);
  };

  std::vector<aggregation_result> results(requests.size());
  std::unique_ptr<column> out_keys;
  auto needs_sum = [](aggregation::Kind k) {
    return k == aggregation::SUM || k == aggregation::COUNT_VALID || k == aggregation::COUNT_ALL || k == aggregation::MEAN;
  };
  auto needs_mm  = [](aggregation::Kind k) {
    return k == aggregation::MIN || k == aggregation::MAX || k == aggregation::ARGMIN || k == aggregation::ARGMAX;
  };
  // SUM_OF_SQUARES / M2 / VARIANCE / STD: the SUM pass plus a second SUM pass over the squared values
  auto needs_sq = [](aggregation::Kind k) {
    return k == aggregation::SUM_OF_SQUARES || k == aggregation::M2 || k == aggregation::VARIANCE || k == aggregation::STD;
  };
  std::size_t passes = 0;  // hash passes over the keys: more than one -> bring every result into key order
  for (auto const& r : requests) {
    bool a = false, b = false, c = false;
    for (auto const& agg : r.aggregations) {
      a = a || needs_sum(agg->kind) || needs_sq(agg->kind);
      b = b || needs_mm(agg->kind);
      c = c || needs_sq(agg->kind);
    }
    passes += (a ? 1 : 0) + (b ? 1 : 0) + (c ? 1 : 0);
  }
  bool const canonical = passes > 1;

  if (_keys.num_rows() == 0 || requests.empty()) {  // empty input -> empty keys + typed empty results (groupby.cu:234)
    for (std::size_t i = 0; i < requests.size(); ++i)
      for (auto const& agg : requests[i].aggregations) {
        data_type t = requests[i].values.type();
        if (agg->kind == aggregation::SUM) t = sum_type(t);
        if (agg->kind == aggregation::COUNT_VALID || agg->kind == aggregation::COUNT_ALL) t = data_type{type_id::INT32};
        if (agg->kind == aggregation::MEAN || agg->kind == aggregation::M2 || agg->kind == aggregation::VARIANCE ||
            agg->kind == aggregation::STD)
          t = data_type{type_id::FLOAT64};
        if (agg->kind == aggregation::SUM_OF_SQUARES) t = sum_type(t);
        if (agg->kind == aggregation::ARGMIN || agg->kind == aggregation::ARGMAX) t = data_type{type_id::INT32};
        // MIN / MAX keep the values' type
        results[i].results.emplace_back(make_empty_column(t));
      }
    if (requests.empty() && _keys.num_rows() > 0) {
      // keys only: aggregate a dummy count to obtain the distinct keys
      auto o   = hash_aggregate(keys, keys, stream, mr);
      out_keys = std::move(o.keys);
    } else {
      out_keys = make_empty_column(keys.type());
    }
    return {decode_keys(std::move(out_keys)), std::move(results)};
  }

  for (std::size_t i = 0; i < requests.size(); ++i) {
    auto const& req = requests[i];
    bool want_sum = false, want_mm = false, want_sq = false, want_arg = false;
    for (auto const& agg : req.aggregations) {
      CUDF_EXPECTS(needs_sum(agg->kind) || needs_mm(agg->kind) || needs_sq(agg->kind),
                   "groupby aggregation kind not implemented on this path (SUM, COUNT, MEAN, MIN, MAX, ARGMIN, ARGMAX, "
                   "SUM_OF_SQUARES, M2, VARIANCE, STD are)");
      want_sum = want_sum || needs_sum(agg->kind) || needs_sq(agg->kind);
      want_mm  = want_mm || needs_mm(agg->kind);
      want_sq  = want_sq || needs_sq(agg->kind);
      want_arg = want_arg || agg->kind == aggregation::ARGMIN || agg->kind == aggregation::ARGMAX;
    }
    hash_agg_out o, q;
    hash_minmax_out mm;
    std::unique_ptr<column> order, mm_order, sq_order, group_of_row;
    if (want_sum || !want_mm) {
      o = hash_aggregate(keys, req.values, stream, mr);
      if (canonical) order = cudf::sorted_order(table_view{{o.keys->view()}}, {}, {}, stream);
    }
    if (want_sq) {  // second SUM pass over the squared values (same validity): SUM_OF_SQUARES
      auto const& v = req.values;
      auto sq       = make_fixed_width_column(sum_type(v.type()), v.size(), mask_state::UNALLOCATED, stream);
      detail::gx_check(gx_square(detail::gx_type(v.type()), detail::row0(v), v.size(), sq->mutable_view().head<void>(),
                                 detail::gxs(stream)),
                       "groupby sum of squares");
      // the squares start at row 0; the validity is the values' (re-based when the view is sliced)
      rmm::device_buffer holder;
      auto const* m = v.has_nulls() ? detail::rebased_mask(v, holder, stream) : nullptr;
      column_view sqv{sq->type(), v.size(), sq->view().head<void>(), m, m ? v.null_count() : 0};
      q        = hash_aggregate(keys, sqv, stream, mr);
      sq_order = cudf::sorted_order(table_view{{q.keys->view()}}, {}, {}, stream);  // passes >= 2: canonical is set
    }
    if (want_mm) {
      mm = hash_minmax(keys, req.values, stream, mr);
      if (canonical) mm_order = cudf::sorted_order(table_view{{mm.keys->view()}}, {}, {}, stream);
      if (!o.keys) {  // only MIN / MAX requested: this pass provides keys and validity
        o.keys        = std::make_unique<column>(mm.keys->view(), stream, mr);
        o.count_valid = std::make_unique<column>(mm.count_valid->view(), stream, mr);
        if (canonical) order = std::make_unique<column>(mm_order->view(), stream, mr);
      }
      if (want_arg && mm.keys->size() > 0) {
        // row -> position of its key among this pass's groups: a lookup table over the distinct keys
        auto const ksz    = static_cast<int>(size_of(keys.type()));
        auto const tbytes = gx_join_table_bytes(ksz, mm.keys->size(), 0.5);
        rmm::device_buffer table{tbytes, stream};
        detail::gx_check(gx_join_build(ksz, mm.keys->view().head<void>(), nullptr, mm.keys->size(), table.data(), tbytes, 0.5,
                                       detail::gxs(stream)),
                         "groupby argmin/argmax dictionary");
        group_of_row = make_fixed_width_column(data_type{type_id::INT32}, keys.size(), mask_state::UNALLOCATED, stream);
        rmm::device_buffer kh;
        auto const* kmask = keys.has_nulls() ? detail::rebased_mask(keys, kh, stream) : nullptr;
        detail::gx_check(gx_join_lookup(ksz, detail::row0(keys), kmask, keys.size(), table.data(), tbytes,
                                        group_of_row->mutable_view().head<int32_t>(), detail::gxs(stream)),
                         "groupby argmin/argmax lookup");
      }
    }
    auto fin = [&](std::unique_ptr<column> c) {
      return canonical ? permute(c->view(), order->view(), stream, mr) : std::move(c);
    };
    auto fin_mm = [&](std::unique_ptr<column> c) {
      return canonical ? permute(c->view(), mm_order->view(), stream, mr) : std::move(c);
    };
    auto const g = o.keys->size();
    // validity of SUM / MEAN: groups without a valid value are null
    auto validity = [&](size_type& nulls) {
      rmm::device_buffer mask = create_null_mask(g, mask_state::ALL_VALID, stream, mr);
      rmm::device_buffer cnt{sizeof(int64_t), stream};
      detail::gx_check(gx_valid_from_counts(o.count_valid->view().head<int32_t>(), g, static_cast<uint32_t*>(mask.data()),
                                            static_cast<int64_t*>(cnt.data()), detail::gxs(stream)),
                       "groupby validity");
      nulls = static_cast<size_type>(detail::read_i64(static_cast<int64_t const*>(cnt.data()), stream));
      return mask;
    };
    for (auto const& agg : req.aggregations) {
      switch (agg->kind) {
        case aggregation::SUM: {
          auto c = std::make_unique<column>(o.sum->view(), stream, mr);
          size_type nulls = 0;
          auto mask       = validity(nulls);
          if (nulls > 0) c->set_null_mask(std::move(mask), nulls);
          results[i].results.emplace_back(fin(std::move(c)));
          break;
        }
        case aggregation::COUNT_VALID: results[i].results.emplace_back(fin(std::make_unique<column>(o.count_valid->view(), stream, mr))); break;
        case aggregation::COUNT_ALL: results[i].results.emplace_back(fin(std::make_unique<column>(o.count_all->view(), stream, mr))); break;
        case aggregation::MEAN: {
          auto c = make_fixed_width_column(data_type{type_id::FLOAT64}, g, mask_state::UNALLOCATED, stream, mr);
          detail::gx_check(gx_mean_from_sum(detail::gx_type(o.sum->type()), o.sum->view().head<void>(),
                                            o.count_valid->view().head<int32_t>(), g, c->mutable_view().head<double>(),
                                            detail::gxs(stream)),
                           "groupby mean");
          size_type nulls = 0;
          auto mask       = validity(nulls);
          if (nulls > 0) c->set_null_mask(std::move(mask), nulls);
          results[i].results.emplace_back(fin(std::move(c)));
          break;
        }
        case aggregation::SUM_OF_SQUARES: {
          auto c = permute(q.sum->view(), sq_order->view(), stream, mr);  // already in key order
          rmm::device_buffer mask = create_null_mask(c->size(), mask_state::ALL_VALID, stream, mr);
          rmm::device_buffer cnt{sizeof(int64_t), stream};
          auto cv = permute(q.count_valid->view(), sq_order->view(), stream, mr);
          detail::gx_check(gx_valid_from_counts(cv->view().head<int32_t>(), c->size(), static_cast<uint32_t*>(mask.data()),
                                                static_cast<int64_t*>(cnt.data()), detail::gxs(stream)),
                           "groupby validity");
          auto const nulls = static_cast<size_type>(detail::read_i64(static_cast<int64_t const*>(cnt.data()), stream));
          if (nulls > 0) c->set_null_mask(std::move(mask), nulls);
          results[i].results.emplace_back(std::move(c));
          break;
        }
        case aggregation::M2:
        case aggregation::VARIANCE:
        case aggregation::STD: {
          // both passes brought into key order, then M2 = sum_sqr - sum^2/count etc. (m2_var_std.cu:44-61,153-190)
          auto ssum = permute(q.sum->view(), sq_order->view(), stream, mr);
          auto sum  = permute(o.sum->view(), order->view(), stream, mr);
          auto cv   = permute(o.count_valid->view(), order->view(), stream, mr);
          auto c    = make_fixed_width_column(data_type{type_id::FLOAT64}, g, mask_state::UNALLOCATED, stream, mr);
          rmm::device_buffer mask = create_null_mask(g, mask_state::ALL_VALID, stream, mr);
          rmm::device_buffer cnt{sizeof(int64_t), stream};
          int ddof = 1;
          if (auto const* sv = dynamic_cast<detail::std_var_aggregation const*>(agg.get())) ddof = sv->_ddof;
          int const mode = agg->kind == aggregation::M2 ? 0 : (agg->kind == aggregation::VARIANCE ? 1 : 2);
          detail::gx_check(gx_var_from_sums(detail::gx_type(sum->type()), ssum->view().head<void>(), sum->view().head<void>(),
                                            cv->view().head<int32_t>(), g, ddof, mode, c->mutable_view().head<double>(),
                                            static_cast<uint32_t*>(mask.data()), static_cast<int64_t*>(cnt.data()),
                                            detail::gxs(stream)),
                           "groupby variance");
          auto const nulls = static_cast<size_type>(detail::read_i64(static_cast<int64_t const*>(cnt.data()), stream));
          if (nulls > 0) c->set_null_mask(std::move(mask), nulls);
          results[i].results.emplace_back(std::move(c));  // already in key order
          break;
        }
        case aggregation::ARGMIN:
        case aggregation::ARGMAX: {
          auto const& target = agg->kind == aggregation::ARGMIN ? mm.mn : mm.mx;
          auto const gm      = mm.keys->size();
          auto c             = make_fixed_width_column(data_type{type_id::INT32}, gm, mask_state::UNALLOCATED, stream, mr);
          if (gm > 0) {
            rmm::device_buffer vh;
            auto const* vmask = req.values.has_nulls() ? detail::rebased_mask(req.values, vh, stream) : nullptr;
            detail::gx_check(gx_groupby_arg_select(detail::gx_type(req.values.type()), detail::row0(req.values), vmask,
                                                   group_of_row->view().head<int32_t>(), keys.size(),
                                                   target->view().head<void>(), gm, c->mutable_view().head<int32_t>(),
                                                   detail::gxs(stream)),
                             "groupby argmin/argmax");
          }
          rmm::device_buffer mask = create_null_mask(gm, mask_state::ALL_VALID, stream, mr);
          rmm::device_buffer cnt{sizeof(int64_t), stream};
          detail::gx_check(gx_valid_from_counts(mm.count_valid->view().head<int32_t>(), gm, static_cast<uint32_t*>(mask.data()),
                                                static_cast<int64_t*>(cnt.data()), detail::gxs(stream)),
                           "groupby validity");
          auto const nulls = static_cast<size_type>(detail::read_i64(static_cast<int64_t const*>(cnt.data()), stream));
          if (nulls > 0) c->set_null_mask(std::move(mask), nulls);
          results[i].results.emplace_back(fin_mm(std::move(c)));
          break;
        }
        case aggregation::MIN:
        case aggregation::MAX: {
          auto const& src = agg->kind == aggregation::MIN ? mm.mn : mm.mx;
          auto c          = std::make_unique<column>(src->view(), stream, mr);
          // validity from THIS pass's counts (its group order)
          rmm::device_buffer mask = create_null_mask(c->size(), mask_state::ALL_VALID, stream, mr);
          rmm::device_buffer cnt{sizeof(int64_t), stream};
          detail::gx_check(gx_valid_from_counts(mm.count_valid->view().head<int32_t>(), c->size(),
                                                static_cast<uint32_t*>(mask.data()), static_cast<int64_t*>(cnt.data()),
                                                detail::gxs(stream)),
                           "groupby validity");
          auto const nulls = static_cast<size_type>(detail::read_i64(static_cast<int64_t const*>(cnt.data()), stream));
          if (nulls > 0) c->set_null_mask(std::move(mask), nulls);
          results[i].results.emplace_back(fin_mm(std::move(c)));
          break;
        }
        default: break;
      }
    }
    if (!out_keys) out_keys = fin(std::move(o.keys));
  }
  return {decode_keys(std::move(out_keys)), std::move(results)};
}

std::pair<std::unique_ptr<table>, std::vector<aggregation_result>> groupby::scan(
  std::span<scan_request const> requests, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(std::all_of(requests.begin(), requests.end(),
                           [this](auto const& r) { return r.values.size() == _keys.num_rows(); }),
               "Size mismatch between request values and groupby keys.", std::invalid_argument);
  // sort helper: stable order of the keys (skipped when they are pre-sorted), nulls AFTER (sort_helper.cu:92-94);
  // null keys are dropped under null_policy::EXCLUDE.  Any number of key columns.
  auto& h           = helper();
  auto const kept   = h.num_keys(stream);
  auto const* labels = h.group_labels(stream);
  std::vector<aggregation_result> results(requests.size());
  for (std::size_t i = 0; i < requests.size(); ++i) {
    std::unique_ptr<column> owner;
    column_view sv = requests[i].values;
    if (!h.is_presorted()) {
      owner = h.grouped_values(requests[i].values, stream, mr);
      sv    = owner->view();
    }
    rmm::device_buffer vh;
    auto const* vmask = sv.has_nulls() ? detail::rebased_mask(sv, vh, stream) : nullptr;
    for (auto const& agg : requests[i].aggregations) {
      int op = -1;
      if (agg->kind == aggregation::SUM) op = GX_OP_SUM;
      if (agg->kind == aggregation::MIN) op = GX_OP_MIN;
      if (agg->kind == aggregation::MAX) op = GX_OP_MAX;
      if (agg->kind == aggregation::COUNT_VALID) op = GX_OP_COUNT_VALID;
      if (agg->kind == aggregation::COUNT_ALL) op = GX_OP_COUNT_ALL;
      CUDF_EXPECTS(op >= 0, "groupby scan kind not implemented on this path (SUM, MIN, MAX, COUNT are)");
      bool const counting = op == GX_OP_COUNT_VALID || op == GX_OP_COUNT_ALL;  // sort/scan.cpp:113-136: INT32, never null
      data_type const ot  = counting ? data_type{type_id::INT32} : (op == GX_OP_SUM) ? sum_type(sv.type()) : sv.type();
      auto out = make_fixed_width_column(ot, kept, mask_state::UNALLOCATED, stream, mr);
      if (kept > 0)
        detail::run_with_scratch(
          [&](void* t, std::size_t* b) {
            // the group labels stand in for the keys: an INT32 "key" column whose runs are the groups
            return gx_segmented_scan(GX_INT32, labels, detail::gx_type(sv.type()), detail::row0(sv), vmask, kept, op,
                                     out->mutable_view().head<void>(), t, b, detail::gxs(stream));
          },
          "groupby scan", stream);
      if (!counting && sv.nullable() && kept > 0) {  // null rows stay null
        rmm::device_buffer m{bitmask_allocation_size_bytes(kept), stream, mr};
        CUDF_CUDA_TRY(hipMemsetAsync(m.data(), 0, m.size(), stream.value()));
        detail::gx_check(gx_bitmask_copy(static_cast<uint32_t*>(m.data()), 0, sv.null_mask(), sv.offset(), kept, detail::gxs(stream)),
                         "groupby scan validity");
        out->set_null_mask(std::move(m), sv.null_count());
      }
      results[i].results.emplace_back(std::move(out));
    }
  }
  return {h.sorted_keys(stream, mr), std::move(results)};
}

}  // namespace groupby
}  // namespace cudf
