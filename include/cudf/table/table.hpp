// cudf/table/table.hpp -- owning set of equally sized columns
// (reference: cpp/include/cudf/table/table.hpp:31-213).
#pragma once
#include <cudf/column/column.hpp>
#include <cudf/table/table_view.hpp>

#include <memory>
#include <vector>

namespace cudf {

class table {
 public:
  table()                        = default;
  ~table()                       = default;
  table(table&&)                 = default;
  table& operator=(table const&) = delete;
  table& operator=(table&&)      = delete;

  explicit table(table const& other, rmm::cuda_stream_view stream = cudf::get_default_stream(),
                 rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());
  table(std::vector<std::unique_ptr<column>>&& columns);
  table(table_view view, rmm::cuda_stream_view stream = cudf::get_default_stream(),
        rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

  [[nodiscard]] size_type num_columns() const noexcept { return static_cast<size_type>(_columns.size()); }
  [[nodiscard]] size_type num_rows() const noexcept { return _num_rows; }
  [[nodiscard]] table_view view() const;
  operator table_view() const { return this->view(); }
  mutable_table_view mutable_view();
  operator mutable_table_view() { return this->mutable_view(); }
  std::vector<std::unique_ptr<column>> release();
  [[nodiscard]] table_view select(std::vector<size_type> const& column_indices) const { return view().select(column_indices); }
  column& get_column(size_type i) { return *(_columns.at(i)); }
  [[nodiscard]] column const& get_column(size_type i) const { return *(_columns.at(i)); }

 private:
  std::vector<std::unique_ptr<column>> _columns{};
  size_type _num_rows{};
};

}  // namespace cudf
