// cudf/table/table_view.hpp -- ordered set of equally sized column views
// (reference: cpp/include/cudf/table/table_view.hpp:41-264; size check cpp/src/table/table_view.cpp).
#pragma once
#include <cudf/column/column_view.hpp>
#include <cudf/types.hpp>

#include <vector>

namespace cudf {
namespace detail {

template <typename ColumnView>
class table_view_base {
 public:
  using iterator       = decltype(std::begin(std::declval<std::vector<ColumnView>&>()));
  using const_iterator = decltype(std::cbegin(std::declval<std::vector<ColumnView> const&>()));

  table_view_base() = default;
  explicit table_view_base(std::vector<ColumnView> const& cols);

  iterator begin() noexcept { return std::begin(_columns); }
  [[nodiscard]] const_iterator begin() const noexcept { return std::begin(_columns); }
  iterator end() noexcept { return std::end(_columns); }
  [[nodiscard]] const_iterator end() const noexcept { return std::end(_columns); }
  [[nodiscard]] ColumnView const& column(size_type column_index) const { return _columns.at(column_index); }
  [[nodiscard]] size_type num_columns() const noexcept { return static_cast<size_type>(_columns.size()); }
  [[nodiscard]] size_type num_rows() const noexcept { return _num_rows; }
  [[nodiscard]] bool is_empty() const noexcept { return num_columns() == 0; }

 protected:
  std::vector<ColumnView> _columns{};
  size_type _num_rows{};
};

}  // namespace detail

class table_view : public detail::table_view_base<column_view> {
  using detail::table_view_base<column_view>::table_view_base;

 public:
  using ColumnView = column_view;
  table_view()     = default;
  // concatenate the columns of several views
  table_view(std::vector<table_view> const& views);
  // the columns with the given indices
  [[nodiscard]] table_view select(std::vector<size_type> const& column_indices) const;
};

class mutable_table_view : public detail::table_view_base<mutable_column_view> {
  using detail::table_view_base<mutable_column_view>::table_view_base;

 public:
  using ColumnView     = mutable_column_view;
  mutable_table_view() = default;
  operator table_view();
};

bool has_nulls(table_view const& view);
bool nullable(table_view const& view);

extern template class detail::table_view_base<column_view>;
extern template class detail::table_view_base<mutable_column_view>;

}  // namespace cudf
