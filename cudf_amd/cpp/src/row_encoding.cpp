// row_encoding.cpp -- see row_encoding.hpp.
#include "row_encoding.hpp"

#include <cudf/column/column_factories.hpp>
#include <cudf/copying.hpp>
#include <cudf/null_mask.hpp>
#include <cudf/sorting.hpp>

namespace cudf {
namespace detail {

namespace {
std::unique_ptr<column> make_i32(size_type n, rmm::cuda_stream_view stream)
{
  return make_numeric_column(data_type{type_id::INT32}, n, mask_state::UNALLOCATED, stream);
}

// view of `c` without its validity (the packed / id columns carry validity separately)
column_view without_mask(column_view const& c)
{
  return column_view{c.type(), c.size(), c.head<void>(), nullptr, 0, c.offset()};
}

// AND of the validity of the columns of `t` as an owned (mask, null_count); empty mask when no column has nulls
std::pair<rmm::device_buffer, size_type> and_of_masks(table_view const& t, rmm::cuda_stream_view stream)
{
  if (!cudf::has_nulls(t)) return {rmm::device_buffer{0, stream}, 0};
  return cudf::bitmask_and(t, stream);
}
}  // namespace

std::unique_ptr<column> pack_columns(std::vector<column_view> const& cols, rmm::cuda_stream_view stream)
{
  CUDF_EXPECTS(!cols.empty() && cols.size() <= 8, "pack_columns: 1 to 8 key columns");
  auto const n = cols.front().size();
  auto out     = make_numeric_column(data_type{type_id::UINT64}, n, mask_state::UNALLOCATED, stream);
  std::vector<void const*> ptrs;
  std::vector<int> dts;
  for (auto const& c : cols) {
    ptrs.push_back(row0(c));
    dts.push_back(gx_type(c.type()));
  }
  gx_check(gx_pack_keys(static_cast<int>(cols.size()), ptrs.data(), dts.data(), n,
                        out->mutable_view().data<uint64_t>(), gxs(stream)),
           "pack_keys");
  return out;
}

std::unique_ptr<column> hash_columns(std::vector<column_view> const& cols, rmm::cuda_stream_view stream)
{
  CUDF_EXPECTS(!cols.empty() && cols.size() <= 8, "hash_columns: 1 to 8 key columns");
  auto const n = cols.front().size();
  auto out     = make_numeric_column(data_type{type_id::UINT64}, n, mask_state::UNALLOCATED, stream);
  std::vector<void const*> ptrs;
  std::vector<int> dts;
  for (auto const& c : cols) {
    ptrs.push_back(row0(c));
    dts.push_back(gx_type(c.type()));
  }
  gx_check(gx_hash_rows64(static_cast<int>(cols.size()), ptrs.data(), dts.data(), n, 0, out->mutable_view().data<uint64_t>(),
                          gxs(stream)),
           "hash_rows64");
  return out;
}

int64_t count_row_mismatches(table_view const& left, table_view const& right, size_type const* lidx, size_type const* ridx,
                             std::size_t npairs, rmm::cuda_stream_view stream)
{
  std::vector<void const*> lp, rp;
  std::vector<int> dts;
  for (size_type k = 0; k < left.num_columns(); ++k) {
    lp.push_back(row0(left.column(k)));
    rp.push_back(row0(right.column(k)));
    dts.push_back(gx_type(left.column(k).type()));
  }
  rmm::device_buffer cnt{sizeof(int64_t), stream};
  gx_check(gx_rows_mismatch_count(left.num_columns(), lp.data(), rp.data(), dts.data(), lidx, ridx, static_cast<int64_t>(npairs),
                                  static_cast<int64_t*>(cnt.data()), gxs(stream)),
           "rows_mismatch_count");
  return read_i64(static_cast<int64_t const*>(cnt.data()), stream);
}

dense_rank_result dense_rank(column_view const& col, rmm::cuda_stream_view stream)
{
  auto const n = col.size();
  dense_rank_result r;
  r.ids = make_i32(n, stream);
  auto rep = make_i32(n, stream);
  rmm::device_buffer ng{sizeof(int64_t), stream};
  rmm::device_buffer holder;
  auto const* mask = col.has_nulls() ? rebased_mask(col, holder, stream) : nullptr;
  run_with_scratch(
    [&](void* t, std::size_t* b) {
      return gx_dense_rank(gx_type(col.type()), row0(col), mask, n, mask ? col.null_count() : 0,
                           r.ids->mutable_view().data<int32_t>(), rep->mutable_view().data<int32_t>(),
                           static_cast<int64_t*>(ng.data()), t, b, gxs(stream));
    },
    "dense_rank", stream);
  r.num_ids = static_cast<size_type>(read_i64(static_cast<int64_t const*>(ng.data()), stream));
  // shrink rep to one entry per id
  auto contents = rep->release();
  r.rep = std::make_unique<column>(data_type{type_id::INT32}, r.num_ids, std::move(*contents.data), rmm::device_buffer{}, 0);
  return r;
}

dense_rank_result dense_row_ids(table_view const& keys, rmm::cuda_stream_view stream)
{
  CUDF_EXPECTS(keys.num_columns() >= 1, "groupby needs at least one key column");
  std::size_t width = 0;
  for (auto const& c : keys) width += size_of(c.type());
  bool const nulls = cudf::has_nulls(keys);
  dense_rank_result r;
  if (width <= 8 && !nulls) {
    std::vector<column_view> cols(keys.begin(), keys.end());
    auto packed = pack_columns(cols, stream);
    r           = dense_rank(packed->view(), stream);
  } else {
    std::unique_ptr<column> cur;
    for (size_type k = 0; k < keys.num_columns(); ++k) {
      // nulls are excluded below through the AND of the masks: rank the values only
      auto one = dense_rank(without_mask(keys.column(k)), stream);
      if (k == 0) {
        if (keys.num_columns() == 1) r = std::move(one);
        else cur = std::move(one.ids);
        continue;
      }
      auto pair = pack_columns({cur->view(), one.ids->view()}, stream);
      auto next = dense_rank(pair->view(), stream);
      if (k == keys.num_columns() - 1) r = std::move(next);
      else cur = std::move(next.ids);
    }
  }
  if (nulls) {
    auto [mask, nc] = and_of_masks(keys, stream);
    r.ids->set_null_mask(std::move(mask), nc);
  }
  return r;
}

// ------------------------------------------------------------------------------------------ row keys
row_keys::row_keys(table_view const& keys, rmm::cuda_stream_view stream) : _keys{keys}
{
  CUDF_EXPECTS(keys.num_columns() >= 1 && keys.num_columns() <= 8, "row_keys: 1 to 8 key columns");
  std::size_t width = 0;
  for (auto const& c : keys) {
    width += size_of(c.type());
    _bare.push_back(without_mask(c));
  }
  _exact = width <= 8;
  _col   = _exact ? pack_columns(_bare, stream) : hash_columns(_bare, stream);
  if (cudf::has_nulls(keys)) {
    auto [mask, nc] = and_of_masks(keys, stream);
    _col->set_null_mask(std::move(mask), nc);
  }
}

std::unique_ptr<table> row_keys::key_columns(column_view const& distinct, rmm::cuda_stream_view stream,
                                             rmm::device_async_resource_ref mr) const
{
  auto const g = distinct.size();
  std::vector<std::unique_ptr<column>> out;
  if (_exact || g == 0) {
    std::vector<void*> ptrs;
    std::vector<int> dts;
    for (auto const& c : _keys) {
      out.emplace_back(make_fixed_width_column(c.type(), g, mask_state::UNALLOCATED, stream, mr));
      ptrs.push_back(out.back()->mutable_view().head<void>());
      dts.push_back(gx_type(c.type()));
    }
    if (g > 0)
      gx_check(gx_unpack_keys(_keys.num_columns(), ptrs.data(), dts.data(), g, distinct.head<uint64_t>() + distinct.offset(),
                              gxs(stream)),
               "unpack_keys");
    return std::make_unique<table>(std::move(out));
  }
  // hashed keys: per key column one streaming groupby MIN + MAX by hash.  MIN == MAX in every group and column <=> all
  // rows of a group carry the same key values (returned as the group's keys) <=> no two different rows shared a hash.
  auto const* kmask = _col->view().null_mask();
  auto const n      = _keys.num_rows();
  std::vector<std::unique_ptr<column>> mns, mxs;
  rmm::device_buffer ng{sizeof(int64_t), stream};
  for (auto const& c : _bare) {
    auto hk = make_numeric_column(data_type{type_id::UINT64}, g, mask_state::UNALLOCATED, stream);
    auto mn = make_fixed_width_column(c.type(), g, mask_state::UNALLOCATED, stream);
    auto mx = make_fixed_width_column(c.type(), g, mask_state::UNALLOCATED, stream);
    run_with_scratch(
      [&](void* t, std::size_t* b) {
        return gx_groupby_min_max(GX_UINT64, _col->view().head<void>(), kmask, gx_type(c.type()), row0(c), nullptr, n, g,
                                  hk->mutable_view().head<void>(), mn->mutable_view().head<void>(), mx->mutable_view().head<void>(),
                                  nullptr, static_cast<int64_t*>(ng.data()), t, b, gxs(stream));
      },
      "row keys: key column", stream);
    if (read_i64(static_cast<int64_t const*>(ng.data()), stream) != g) return nullptr;
    auto order = cudf::sorted_order(table_view{{hk->view()}}, {}, {}, stream);  // ascending-hash order
    auto t     = cudf::gather(table_view{{mn->view(), mx->view()}}, order->view(), out_of_bounds_policy::DONT_CHECK, stream);
    auto cols  = t->release();
    mns.emplace_back(std::move(cols[0]));
    mxs.emplace_back(std::move(cols[1]));
  }
  std::vector<column_view> mnv, mxv;
  for (auto const& c : mns) mnv.push_back(c->view());
  for (auto const& c : mxs) mxv.push_back(c->view());
  if (count_row_mismatches(table_view{mnv}, table_view{mxv}, nullptr, nullptr, static_cast<std::size_t>(g), stream) != 0) return nullptr;
  // ascending-hash position of every entry of `distinct`: the inverse of its sorted order
  auto order = cudf::sorted_order(table_view{{distinct}}, {}, {}, stream);
  auto back  = cudf::sorted_order(table_view{{order->view()}}, {}, {}, stream);
  return cudf::gather(table_view{mnv}, back->view(), out_of_bounds_policy::DONT_CHECK, stream, mr);
}

// ------------------------------------------------------------------------------------------ encoder
row_encoder::~row_encoder() = default;

row_encoder::dictionary row_encoder::make_dictionary(column_view const& packed, dense_rank_result const& r, size_type nulls,
                                                     rmm::cuda_stream_view stream) const
{
  dictionary d;
  d.num_ids = r.num_ids;
  d.null_id = nulls > 0 ? r.num_ids - 1 : -1;  // nulls rank last
  auto const nvalues = nulls > 0 ? r.num_ids - 1 : r.num_ids;
  // distinct values in id order: the table's "build row" of a value IS its id
  column_view rep_valid{data_type{type_id::INT32}, nvalues, r.rep->view().head<void>(), nullptr, 0};
  auto distinct = cudf::gather(table_view{{without_mask(packed)}}, rep_valid, out_of_bounds_policy::DONT_CHECK, stream);
  d.table_bytes = gx_join_table_bytes(8, nvalues, 0.5);
  d.table       = rmm::device_buffer{d.table_bytes, stream};
  gx_check(gx_join_build(8, row0(distinct->get_column(0).view()), nullptr, nvalues, d.table.data(), d.table_bytes, 0.5,
                         gxs(stream)),
           "row_encoder dictionary build");
  return d;
}

std::unique_ptr<column> row_encoder::lookup(dictionary const& d, column_view const& packed, bitmask_type const* valid,
                                            rmm::cuda_stream_view stream) const
{
  auto ids = make_i32(packed.size(), stream);
  gx_check(gx_join_lookup(8, row0(packed), valid, packed.size(), d.table.data(), d.table_bytes,
                          ids->mutable_view().data<int32_t>(), gxs(stream)),
           "row_encoder lookup");
  // null probe values: the build side's null id when nulls compare equal and it has one; otherwise they
  // keep JoinNoMatch, a value no build id equals
  if (valid && _nulls_equal && d.null_id >= 0)
    gx_check(gx_fill_nulls(4, ids->mutable_view().data<int32_t>(), valid, packed.size(),
                           static_cast<uint64_t>(static_cast<uint32_t>(d.null_id)), gxs(stream)),
             "row_encoder null ids");
  return ids;
}

row_encoder::row_encoder(table_view const& build, bool nulls_equal, rmm::cuda_stream_view stream, bool allow_hash)
  : _nulls_equal{nulls_equal}
{
  std::size_t width = 0;
  for (auto const& c : build) {
    _types.push_back(c.type());
    width += size_of(c.type());
  }
  auto const ncols = build.num_columns();
  bool const nulls = cudf::has_nulls(build);
  std::vector<column_view> cols(build.begin(), build.end());
  if (ncols == 1 || (width <= 8 && !nulls)) {
    // one packed key; a single column keeps its validity (the join's own null handling applies)
    _pack_only  = true;
    std::vector<column_view> bare;
    for (auto const& c : cols) bare.push_back(without_mask(c));
    _build_keys = pack_columns(bare, stream);
    if (ncols == 1 && cols[0].has_nulls()) {
      rmm::device_buffer holder;
      auto const* m = rebased_mask(cols[0], holder, stream);
      _build_keys->set_null_mask(rmm::device_buffer{m, bitmask_allocation_size_bytes(cols[0].size()), stream},
                                 cols[0].null_count());
    }
    return;
  }
  if (allow_hash && ncols <= 8 && (!nulls || !_nulls_equal)) {
    // wider rows: the 64-bit row hash is the key (one pass over the columns); rows holding a null match nothing under
    // null != null and are not inserted.  With null == null and nulls on the build side the exact encoding below runs.
    _hashed = true;
    std::vector<column_view> bare;
    for (auto const& c : cols) bare.push_back(without_mask(c));
    _build_keys = hash_columns(bare, stream);
    if (nulls) {
      auto [mask, nc] = and_of_masks(build, stream);
      _build_keys->set_null_mask(std::move(mask), nc);
    }
    return;
  }
  // dictionary per column, then per (ids so far, next column's ids) pair
  std::unique_ptr<column> cur;
  for (size_type k = 0; k < ncols; ++k) {
    auto packed = pack_columns({without_mask(cols[k])}, stream);
    if (cols[k].has_nulls()) {
      rmm::device_buffer holder;
      auto const* m = rebased_mask(cols[k], holder, stream);
      packed->set_null_mask(rmm::device_buffer{m, bitmask_allocation_size_bytes(cols[k].size()), stream}, cols[k].null_count());
    }
    auto r = dense_rank(packed->view(), stream);
    _col_dict.push_back(make_dictionary(packed->view(), r, cols[k].null_count(), stream));
    if (k == 0) {
      cur = std::move(r.ids);
      continue;
    }
    auto pair = pack_columns({cur->view(), r.ids->view()}, stream);
    if (k == ncols - 1) {
      _build_keys = std::move(pair);
    } else {
      auto pr = dense_rank(pair->view(), stream);
      _pair_dict.push_back(make_dictionary(pair->view(), pr, 0, stream));
      cur = std::move(pr.ids);
    }
  }
  if (!_nulls_equal && nulls) {  // rows holding a null match nothing: they are not inserted
    auto [mask, nc] = and_of_masks(build, stream);
    _build_keys->set_null_mask(std::move(mask), nc);
  }
}

std::unique_ptr<column> row_encoder::encode(table_view const& probe, rmm::cuda_stream_view stream) const
{
  auto const ncols = probe.num_columns();
  std::vector<column_view> cols(probe.begin(), probe.end());
  bool const nulls = cudf::has_nulls(probe);
  std::unique_ptr<column> out;
  if (_pack_only || _hashed) {
    std::vector<column_view> bare;
    for (auto const& c : cols) bare.push_back(without_mask(c));
    out = _hashed ? hash_columns(bare, stream) : pack_columns(bare, stream);
  } else {
    std::unique_ptr<column> cur;
    for (size_type k = 0; k < ncols; ++k) {
      auto packed = pack_columns({without_mask(cols[k])}, stream);
      rmm::device_buffer holder;
      auto const* m = cols[k].has_nulls() ? rebased_mask(cols[k], holder, stream) : nullptr;
      auto ids      = lookup(_col_dict[k], packed->view(), m, stream);
      if (k == 0) {
        cur = std::move(ids);
        continue;
      }
      auto pair = pack_columns({cur->view(), ids->view()}, stream);
      if (k == ncols - 1) out = std::move(pair);
      else cur = lookup(_pair_dict[k - 1], pair->view(), nullptr, stream);
    }
  }
  // validity of the encoded key: a single column keeps its own; several columns: rows holding a null can
  // match nothing when the build side has no nulls (pack-only) or when nulls compare unequal
  if (nulls && (_pack_only || _hashed || !_nulls_equal)) {  // hashed + null == null: the build side has no nulls
    auto [mask, nc] = and_of_masks(probe, stream);
    out->set_null_mask(std::move(mask), nc);
  }
  return out;
}

}  // namespace detail
}  // namespace cudf
