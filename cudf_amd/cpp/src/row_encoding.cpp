// row_encoding.cpp -- see row_encoding.hpp.
#include "row_encoding.hpp"

#include <cudf/column/column_factories.hpp>
#include <cudf/copying.hpp>
#include <cudf/null_mask.hpp>

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

row_encoder::row_encoder(table_view const& build, bool nulls_equal, rmm::cuda_stream_view stream) : _nulls_equal{nulls_equal}
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
  if (_pack_only) {
    std::vector<column_view> bare;
    for (auto const& c : cols) bare.push_back(without_mask(c));
    out = pack_columns(bare, stream);
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
  if (nulls && (_pack_only || !_nulls_equal)) {
    auto [mask, nc] = and_of_masks(probe, stream);
    out->set_null_mask(std::move(mask), nc);
  }
  return out;
}

}  // namespace detail
}  // namespace cudf
