// column / column_view / table / table_view / null mask / scalar: the data model of the API
// (reference: cpp/src/column/column_view.cpp:101-132, column.cpp, cpp/src/table/*.cpp,
// cpp/src/bitmask/null_mask.cu:48-56,152,339-409, cpp/src/scalar/scalar.cpp).
#include "common.hpp"

#include <cudf/column/column_factories.hpp>
#include <cudf/null_mask.hpp>
#include <cudf/scalar/scalar.hpp>
#include <cudf/table/table.hpp>
#include <cudf/table/table_view.hpp>

#include <algorithm>
#include <numeric>

namespace cudf {
namespace detail {

column_view_base::column_view_base(data_type type, size_type size, void const* data, bitmask_type const* null_mask,
                                   size_type null_count, size_type offset)
  : _type{type}, _size{size}, _data{data}, _null_mask{null_mask}, _null_count{null_count}, _offset{offset}
{
  CUDF_EXPECTS(size >= 0, "Column size cannot be negative.");
  if (type.id() == type_id::EMPTY) {
    _null_count = size;
    CUDF_EXPECTS(nullptr == data, "EMPTY column should have no data.");
    CUDF_EXPECTS(nullptr == null_mask, "EMPTY column should have no null mask.");
  } else if (is_fixed_width(type)) {
    if (size > 0) CUDF_EXPECTS(nullptr != data, "Null data pointer.");
  }
  CUDF_EXPECTS(offset >= 0, "Invalid offset.");
  if ((null_count > 0) and (type.id() != type_id::EMPTY)) {
    CUDF_EXPECTS(nullptr != null_mask, "Invalid null mask for non-zero null count.");
  }
}

size_type column_view_base::null_count(size_type begin, size_type end, rmm::cuda_stream_view stream) const
{
  CUDF_EXPECTS((begin >= 0) && (end <= size()) && (begin <= end), "Range is out of bounds.");
  if (!nullable()) return 0;
  return cudf::null_count(null_mask(), offset() + begin, offset() + end, stream);
}

bitmask_type const* rebased_mask(column_view const& c, rmm::device_buffer& holder, rmm::cuda_stream_view stream)
{
  if (!c.nullable()) return nullptr;
  if (c.offset() == 0) return c.null_mask();
  if (c.offset() % 32 == 0) return c.null_mask() + c.offset() / 32;  // word aligned: plain pointer shift
  // unaligned slice: the bits are re-based on the device (cudf::copy_bitmask, null_mask.cu:357-390)
  holder = rmm::device_buffer{bitmask_allocation_size_bytes(c.size()), stream};
  CUDF_CUDA_TRY(hipMemsetAsync(holder.data(), 0, holder.size(), stream.value()));
  gx_check(gx_bitmask_copy(static_cast<uint32_t*>(holder.data()), 0, c.null_mask(), c.offset(), c.size(), gxs(stream)),
           "copy_bitmask");
  return static_cast<bitmask_type const*>(holder.data());
}

template <typename ColumnView>
table_view_base<ColumnView>::table_view_base(std::vector<ColumnView> const& cols) : _columns{cols}
{
  if (num_columns() > 0) {
    std::for_each(_columns.begin(), _columns.end(), [this](ColumnView const& col) {
      CUDF_EXPECTS(col.size() == _columns.front().size(), "Column size mismatch.");
    });
    _num_rows = _columns.front().size();
  } else {
    _num_rows = 0;
  }
}
template class table_view_base<column_view>;
template class table_view_base<mutable_column_view>;

}  // namespace detail

column_view::column_view(data_type type, size_type size, void const* data, bitmask_type const* null_mask,
                         size_type null_count, size_type offset, std::vector<column_view> const& children)
  : detail::column_view_base{type, size, data, null_mask, null_count, offset}, _children{children}
{
  if (type.id() == type_id::EMPTY) { CUDF_EXPECTS(num_children() == 0, "EMPTY column cannot have children."); }
}

mutable_column_view::mutable_column_view(data_type type, size_type size, void* data, bitmask_type* null_mask,
                                         size_type null_count, size_type offset,
                                         std::vector<mutable_column_view> const& children)
  : detail::column_view_base{type, size, data, null_mask, null_count, offset}, mutable_children{children}
{
  if (type.id() == type_id::EMPTY) { CUDF_EXPECTS(num_children() == 0, "EMPTY column cannot have children."); }
}

void mutable_column_view::set_null_count(size_type new_null_count)
{
  if (new_null_count > 0) { CUDF_EXPECTS(nullable(), "Invalid null count."); }
  _null_count = new_null_count;
}

mutable_column_view::operator column_view() const
{
  std::vector<column_view> child_views(mutable_children.begin(), mutable_children.end());
  return column_view{_type, _size, _data, _null_mask, _null_count, _offset, std::move(child_views)};
}

column_view bit_cast(column_view const& input, data_type type)
{
  CUDF_EXPECTS(is_fixed_width(input.type()) && size_of(type) == size_of(input.type()),
               "bit_cast requires fixed-width types of equal size");
  return column_view{type, input.size(), input.head<void>(), input.null_mask(), input.null_count(), input.offset()};
}

// ------------------------------------------------------------------------------------ column
column::column(column const& other, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
  : _type{other._type},
    _size{other._size},
    _data{other._data, stream, mr},
    _null_mask{other._null_mask, stream, mr},
    _null_count{other._null_count}
{
  _children.reserve(other._children.size());
  for (auto const& c : other._children) _children.emplace_back(std::make_unique<column>(*c, stream, mr));
}

column::column(column&& other) noexcept
  : _type{other._type},
    _size{other._size},
    _data{std::move(other._data)},
    _null_mask{std::move(other._null_mask)},
    _null_count{other._null_count},
    _children{std::move(other._children)}
{
  other._size       = 0;
  other._null_count = 0;
  other._type       = data_type{type_id::EMPTY};
}

column::column(column_view view, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
  : _type{view.type()}, _size{view.size()}, _null_count{view.null_count()}
{
  CUDF_EXPECTS(view.num_children() == 0 && (is_fixed_width(view.type()) || view.type().id() == type_id::EMPTY),
               "Only fixed-width columns can be copied on this path", cudf::data_type_error);
  _data = rmm::device_buffer{detail::row0(view), static_cast<std::size_t>(view.size()) * size_of(view.type()), stream, mr};
  if (view.nullable()) {
    rmm::device_buffer holder;
    auto const* m = detail::rebased_mask(view, holder, stream);
    _null_mask    = rmm::device_buffer{m, bitmask_allocation_size_bytes(view.size()), stream, mr};
  }
}

column_view column::view() const
{
  std::vector<column_view> child_views;
  child_views.reserve(_children.size());
  for (auto const& c : _children) child_views.emplace_back(*c);
  return column_view{type(), size(), _data.data(), static_cast<bitmask_type const*>(_null_mask.data()), null_count(),
                     0, child_views};
}

mutable_column_view column::mutable_view()
{
  std::vector<mutable_column_view> child_views;
  child_views.reserve(_children.size());
  for (auto const& c : _children) child_views.emplace_back(*c);
  return mutable_column_view{type(), size(), _data.data(), static_cast<bitmask_type*>(_null_mask.data()), _null_count,
                             0, child_views};
}

void column::set_null_mask(rmm::device_buffer&& new_null_mask, size_type new_null_count)
{
  if (new_null_count > 0) {
    CUDF_EXPECTS(new_null_mask.size() >= bitmask_allocation_size_bytes(this->size()),
                 "Column with null values must be nullable and the null mask buffer size should match the size of the column.");
  }
  _null_mask  = std::move(new_null_mask);
  _null_count = new_null_count;
}

void column::set_null_count(size_type new_null_count)
{
  if (new_null_count > 0) { CUDF_EXPECTS(nullable(), "Invalid null count."); }
  _null_count = new_null_count;
}

column::contents column::release() noexcept
{
  _size       = 0;
  _null_count = 0;
  _type       = data_type{type_id::EMPTY};
  return column::contents{std::make_unique<rmm::device_buffer>(std::move(_data)),
                          std::make_unique<rmm::device_buffer>(std::move(_null_mask)), std::move(_children)};
}

std::unique_ptr<column> make_empty_column(data_type type)
{
  return std::make_unique<column>(type, 0, rmm::device_buffer{}, rmm::device_buffer{}, 0);
}
std::unique_ptr<column> make_empty_column(type_id id) { return make_empty_column(data_type{id}); }

std::unique_ptr<column> make_fixed_width_column(data_type type, size_type size, mask_state state,
                                                rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(is_fixed_width(type), "Invalid, non-fixed-width type.", cudf::data_type_error);
  CUDF_EXPECTS(size >= 0, "Column size cannot be negative.");
  return std::make_unique<column>(type, size, rmm::device_buffer{static_cast<std::size_t>(size) * size_of(type), stream, mr},
                                  create_null_mask(size, state, stream, mr), state_null_count(state, size));
}
std::unique_ptr<column> make_numeric_column(data_type type, size_type size, mask_state state,
                                            rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  return make_fixed_width_column(type, size, state, stream, mr);
}

// ------------------------------------------------------------------------------------ tables
table_view::table_view(std::vector<table_view> const& views)
{
  if (!views.empty()) {
    _num_rows = views.front().num_rows();
    for (auto const& v : views) {
      CUDF_EXPECTS(v.num_columns() == 0 || v.num_rows() == _num_rows || _columns.empty(), "All tables must have the same number of rows");
      if (_columns.empty() && v.num_columns() > 0) _num_rows = v.num_rows();
      _columns.insert(_columns.end(), v.begin(), v.end());
    }
  }
}

table_view table_view::select(std::vector<size_type> const& column_indices) const
{
  std::vector<column_view> cols;
  cols.reserve(column_indices.size());
  for (auto i : column_indices) cols.push_back(column(i));
  return table_view{cols};
}

mutable_table_view::operator table_view()
{
  std::vector<column_view> cols{begin(), end()};
  return table_view{cols};
}

bool has_nulls(table_view const& view)
{
  return std::any_of(view.begin(), view.end(), [](column_view const& c) { return c.has_nulls(); });
}
bool nullable(table_view const& view)
{
  return std::any_of(view.begin(), view.end(), [](column_view const& c) { return c.nullable(); });
}

table::table(table const& other, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
  : _num_rows{other.num_rows()}
{
  _columns.reserve(other._columns.size());
  for (auto const& c : other._columns) _columns.emplace_back(std::make_unique<column>(*c, stream, mr));
}

table::table(std::vector<std::unique_ptr<column>>&& columns) : _columns{std::move(columns)}
{
  if (num_columns() > 0) {
    for (auto const& c : _columns) {
      CUDF_EXPECTS(c, "Unexpected null column");
      CUDF_EXPECTS(c->size() == _columns.front()->size(), "Column size mismatch: " + std::to_string(c->size()) +
                                                            " != " + std::to_string(_columns.front()->size()));
    }
    _num_rows = _columns.front()->size();
  }
}

table::table(table_view view, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
  : _num_rows{view.num_rows()}
{
  _columns.reserve(view.num_columns());
  for (auto const& c : view) _columns.emplace_back(std::make_unique<column>(c, stream, mr));
}

table_view table::view() const
{
  std::vector<column_view> views;
  views.reserve(_columns.size());
  for (auto const& c : _columns) views.push_back(c->view());
  return table_view{views};
}

mutable_table_view table::mutable_view()
{
  std::vector<mutable_column_view> views;
  views.reserve(_columns.size());
  for (auto const& c : _columns) views.push_back(c->mutable_view());
  return mutable_table_view{views};
}

std::vector<std::unique_ptr<column>> table::release()
{
  _num_rows = 0;
  return std::move(_columns);
}

// ------------------------------------------------------------------------------------ null masks
size_type state_null_count(mask_state state, size_type size)
{
  switch (state) {
    case mask_state::UNALLOCATED: return 0;
    case mask_state::ALL_NULL: return size;
    case mask_state::ALL_VALID: return 0;
    default: CUDF_FAIL("Invalid null mask state.");
  }
}

std::size_t bitmask_allocation_size_bytes(size_type number_of_bits, std::size_t padding_boundary)
{
  CUDF_EXPECTS(padding_boundary > 0, "Invalid padding boundary");
  auto const necessary_bytes = (static_cast<std::size_t>(number_of_bits) + 7) / 8;
  return (necessary_bytes + padding_boundary - 1) / padding_boundary * padding_boundary;
}

size_type num_bitmask_words(size_type number_of_bits) { return (number_of_bits + 31) / 32; }

rmm::device_buffer create_null_mask(size_type size, mask_state state, rmm::cuda_stream_view stream,
                                    rmm::device_async_resource_ref mr)
{
  if (state == mask_state::UNALLOCATED) return rmm::device_buffer{0, stream, mr};
  rmm::device_buffer mask{bitmask_allocation_size_bytes(size), stream, mr};
  if (state != mask_state::UNINITIALIZED && mask.size() > 0) {
    CUDF_CUDA_TRY(hipMemsetAsync(mask.data(), state == mask_state::ALL_VALID ? 0xFF : 0x00, mask.size(), stream.value()));
  }
  return mask;
}

void set_null_mask(bitmask_type* bitmask, size_type begin_bit, size_type end_bit, bool valid, rmm::cuda_stream_view stream)
{
  CUDF_EXPECTS(begin_bit >= 0, "Invalid range.");
  CUDF_EXPECTS(begin_bit <= end_bit, "Invalid bit range.");
  if (begin_bit == end_bit || bitmask == nullptr) return;
  detail::gx_check(gx_bitmask_set(bitmask, begin_bit, end_bit, valid ? 1 : 0, detail::gxs(stream)), "set_null_mask");
}

size_type null_count(bitmask_type const* bitmask, size_type start, size_type stop, rmm::cuda_stream_view stream)
{
  if (bitmask == nullptr) return 0;
  CUDF_EXPECTS(start >= 0, "Invalid range.");
  CUDF_EXPECTS(start <= stop, "Invalid bit range.");
  if (start == stop) return 0;
  rmm::device_buffer cnt{sizeof(int64_t), stream};
  detail::gx_check(gx_bitmask_count(bitmask, start, stop, static_cast<int64_t*>(cnt.data()), detail::gxs(stream)), "null_count");
  auto const set_bits = detail::read_i64(static_cast<int64_t const*>(cnt.data()), stream);
  return static_cast<size_type>((stop - start) - set_bits);
}

std::pair<rmm::device_buffer, size_type> bitmask_and(table_view const& view, rmm::cuda_stream_view stream,
                                                     rmm::device_async_resource_ref mr)
{
  std::vector<uint32_t const*> masks;
  std::vector<rmm::device_buffer> holders(view.num_columns());
  size_type k = 0;
  for (auto const& c : view) {
    if (c.nullable()) masks.push_back(detail::rebased_mask(c, holders[k], stream));
    ++k;
  }
  if (masks.empty() || view.num_rows() == 0) return {rmm::device_buffer{0, stream, mr}, 0};
  rmm::device_buffer out{bitmask_allocation_size_bytes(view.num_rows()), stream, mr};
  CUDF_CUDA_TRY(hipMemsetAsync(out.data(), 0, out.size(), stream.value()));
  rmm::device_buffer cnt{sizeof(int64_t), stream};
  detail::gx_check(gx_bitmask_and(masks.data(), static_cast<int>(masks.size()), view.num_rows(),
                                  static_cast<uint32_t*>(out.data()), static_cast<int64_t*>(cnt.data()), detail::gxs(stream)),
                   "bitmask_and");
  auto const valid = detail::read_i64(static_cast<int64_t const*>(cnt.data()), stream);
  return {std::move(out), static_cast<size_type>(view.num_rows() - valid)};
}

// ------------------------------------------------------------------------------------ scalar
scalar::scalar(data_type type, bool is_valid, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
  : _type{type}, _is_valid{1, stream, mr}
{
  set_valid_async(is_valid, stream);  // a memset: no host source, nothing to wait for
}

void scalar::set_valid_async(bool is_valid, rmm::cuda_stream_view stream)
{
  CUDF_CUDA_TRY(hipMemsetAsync(_is_valid.data(), is_valid ? 1 : 0, 1, stream.value()));
}

bool scalar::is_valid(rmm::cuda_stream_view stream) const
{
  bool v = false;
  CUDF_CUDA_TRY(hipMemcpyAsync(&v, _is_valid.data(), 1, hipMemcpyDeviceToHost, stream.value()));
  stream.synchronize();
  return v;
}

}  // namespace cudf
