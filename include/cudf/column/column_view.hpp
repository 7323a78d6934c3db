// cudf/column/column_view.hpp -- non-owning, immutable / mutable views of device column buffers
// (reference: cpp/include/cudf/column/column_view.hpp:36-700; checks in
// cpp/src/column/column_view.cpp:101-132).  Arrow layout: data pointer, optional validity bitmap
// (LSB-first uint32 words, 1 = valid), element offset for zero-copy slices, children.
// Only fixed-width leaf columns are functional on this hot path; children are carried for ABI shape.
#pragma once
#include <cudf/types.hpp>
#include <cudf/utilities/default_stream.hpp>
#include <cudf/utilities/error.hpp>

#include <vector>

namespace cudf {
namespace detail {

class column_view_base {
 public:
  template <typename T = void>
  [[nodiscard]] T const* head() const noexcept
  {
    return static_cast<T const*>(get_data());
  }
  template <typename T>
  [[nodiscard]] T const* data() const noexcept
  {
    return head<T>() + _offset;
  }
  template <typename T>
  [[nodiscard]] T const* begin() const noexcept
  {
    return data<T>();
  }
  template <typename T>
  [[nodiscard]] T const* end() const noexcept
  {
    return begin<T>() + size();
  }
  [[nodiscard]] size_type size() const noexcept { return _size; }
  [[nodiscard]] bool is_empty() const noexcept { return size() == 0; }
  [[nodiscard]] data_type type() const noexcept { return _type; }
  [[nodiscard]] bool nullable() const noexcept { return nullptr != _null_mask; }
  [[nodiscard]] size_type null_count() const { return _null_count; }
  // nulls in rows [begin, end): counts bits on the device (synchronises `stream`)
  [[nodiscard]] size_type null_count(size_type begin, size_type end,
                                     rmm::cuda_stream_view stream = cudf::get_default_stream()) const;
  [[nodiscard]] bool has_nulls() const { return null_count() > 0; }
  [[nodiscard]] bool has_nulls(size_type begin, size_type end,
                               rmm::cuda_stream_view stream = cudf::get_default_stream()) const
  {
    return null_count(begin, end, stream) > 0;
  }
  [[nodiscard]] bitmask_type const* null_mask() const noexcept { return _null_mask; }
  [[nodiscard]] size_type offset() const noexcept { return _offset; }

 protected:
  [[nodiscard]] virtual void const* get_data() const noexcept { return _data; }

  data_type _type{type_id::EMPTY};
  size_type _size{};
  void const* _data{};
  bitmask_type const* _null_mask{};
  mutable size_type _null_count{};
  size_type _offset{};

  column_view_base()                                   = default;
  virtual ~column_view_base()                          = default;
  column_view_base(column_view_base const&)            = default;
  column_view_base(column_view_base&&)                 = default;
  column_view_base& operator=(column_view_base const&) = default;
  column_view_base& operator=(column_view_base&&)      = default;

  column_view_base(data_type type, size_type size, void const* data, bitmask_type const* null_mask,
                   size_type null_count, size_type offset = 0);
};

}  // namespace detail

class column_view : public detail::column_view_base {
 public:
  column_view()                              = default;
  ~column_view() override                    = default;
  column_view(column_view const&)            = default;
  column_view(column_view&&)                 = default;
  column_view& operator=(column_view const&) = default;
  column_view& operator=(column_view&&)      = default;

  column_view(data_type type, size_type size, void const* data, bitmask_type const* null_mask,
              size_type null_count, size_type offset = 0, std::vector<column_view> const& children = {});

  [[nodiscard]] column_view child(size_type child_index) const noexcept { return _children[child_index]; }
  [[nodiscard]] size_type num_children() const noexcept { return static_cast<size_type>(_children.size()); }
  auto child_begin() const noexcept { return _children.cbegin(); }
  auto child_end() const noexcept { return _children.cend(); }

 private:
  friend column_view bit_cast(column_view const& input, data_type type);
  std::vector<column_view> _children{};
};

class mutable_column_view : public detail::column_view_base {
 public:
  mutable_column_view()                                      = default;
  ~mutable_column_view() override                            = default;
  mutable_column_view(mutable_column_view const&)            = default;
  mutable_column_view(mutable_column_view&&)                 = default;
  mutable_column_view& operator=(mutable_column_view const&) = default;
  mutable_column_view& operator=(mutable_column_view&&)      = default;

  mutable_column_view(data_type type, size_type size, void* data, bitmask_type* null_mask, size_type null_count,
                      size_type offset = 0, std::vector<mutable_column_view> const& children = {});

  template <typename T = void>
  [[nodiscard]] T* head() const noexcept
  {
    return const_cast<T*>(detail::column_view_base::head<T>());
  }
  template <typename T>
  [[nodiscard]] T* data() const noexcept
  {
    return const_cast<T*>(detail::column_view_base::data<T>());
  }
  template <typename T>
  [[nodiscard]] T* begin() const noexcept
  {
    return data<T>();
  }
  template <typename T>
  [[nodiscard]] T* end() const noexcept
  {
    return begin<T>() + size();
  }
  [[nodiscard]] bitmask_type* null_mask() const noexcept
  {
    return const_cast<bitmask_type*>(detail::column_view_base::null_mask());
  }
  void set_null_count(size_type new_null_count);
  [[nodiscard]] mutable_column_view child(size_type child_index) const noexcept { return mutable_children[child_index]; }
  [[nodiscard]] size_type num_children() const noexcept { return static_cast<size_type>(mutable_children.size()); }
  operator column_view() const;

 private:
  std::vector<mutable_column_view> mutable_children;
};

// zero-copy reinterpretation between same-width fixed-width types
column_view bit_cast(column_view const& input, data_type type);

}  // namespace cudf
