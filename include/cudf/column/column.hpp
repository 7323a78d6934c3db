// cudf/column/column.hpp -- owning device column (reference: cpp/include/cudf/column/column.hpp:36-331).
// Owns an rmm::device_buffer of data and an optional validity bitmap; release() hands both out.
#pragma once
#include <cudf/column/column_view.hpp>
#include <cudf/null_mask.hpp>
#include <cudf/types.hpp>
#include <cudf/utilities/default_stream.hpp>
#include <cudf/utilities/memory_resource.hpp>
#include <rmm/device_buffer.hpp>
#include <rmm/device_uvector.hpp>

#include <limits>
#include <memory>
#include <stdexcept>
#include <utility>
#include <vector>

namespace cudf {

class column {
 public:
  column()                         = default;
  ~column()                        = default;
  column& operator=(column const&) = delete;
  column& operator=(column&&)      = delete;

  // deep copy on `stream` with memory from `mr`
  column(column const& other, rmm::cuda_stream_view stream = cudf::get_default_stream(),
         rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());
  column(column&& other) noexcept;

  // adopt a typed device vector (column.hpp:64-81): size > INT32_MAX -> std::overflow_error
  template <typename T>
  column(rmm::device_uvector<T>&& other, rmm::device_buffer&& null_mask, size_type null_count)
    : _type{data_type{type_to_id<T>()}},
      _size{checked_size(other.size())},
      _data{other.release()},
      _null_mask{std::move(null_mask)},
      _null_count{null_count}
  {
  }

  column(data_type dtype, size_type size, rmm::device_buffer&& data, rmm::device_buffer&& null_mask,
         size_type null_count, std::vector<std::unique_ptr<column>>&& children = {})
    : _type{dtype},
      _size{size},
      _data{std::move(data)},
      _null_mask{std::move(null_mask)},
      _null_count{null_count},
      _children{std::move(children)}
  {
    CUDF_EXPECTS(size >= 0, "Column size cannot be negative.");
  }

  // deep copy of a view
  explicit column(column_view view, rmm::cuda_stream_view stream = cudf::get_default_stream(),
                  rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

  [[nodiscard]] data_type type() const noexcept { return _type; }
  [[nodiscard]] size_type size() const noexcept { return _size; }
  [[nodiscard]] size_type null_count() const { return _null_count; }
  void set_null_mask(rmm::device_buffer&& new_null_mask, size_type new_null_count);
  void set_null_count(size_type new_null_count);
  [[nodiscard]] bool nullable() const noexcept { return _null_mask.size() > 0; }
  [[nodiscard]] bool has_nulls() const noexcept { return null_count() > 0; }
  [[nodiscard]] size_type num_children() const noexcept { return static_cast<size_type>(_children.size()); }
  column& child(size_type i) noexcept { return *_children[i]; }
  [[nodiscard]] column const& child(size_type i) const noexcept { return *_children[i]; }

  struct contents {
    std::unique_ptr<rmm::device_buffer> data;
    std::unique_ptr<rmm::device_buffer> null_mask;
    std::vector<std::unique_ptr<column>> children;
  };
  // After release() the column is empty: size() == 0, null_count() == 0, type() == EMPTY
  contents release() noexcept;

  [[nodiscard]] column_view view() const;
  operator column_view() const { return this->view(); }
  mutable_column_view mutable_view();
  operator mutable_column_view() { return this->mutable_view(); }

 private:
  static size_type checked_size(std::size_t n)
  {
    CUDF_EXPECTS(n <= static_cast<std::size_t>(std::numeric_limits<size_type>::max()),
                 "The device_uvector size exceeds the column size limit", std::overflow_error);
    return static_cast<size_type>(n);
  }
  data_type _type{type_id::EMPTY};
  size_type _size{};
  rmm::device_buffer _data{};
  rmm::device_buffer _null_mask{};
  mutable size_type _null_count{};
  std::vector<std::unique_ptr<column>> _children{};
};

}  // namespace cudf
