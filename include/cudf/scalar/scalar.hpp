// cudf/scalar/scalar.hpp -- device-resident scalar values returned by cudf::reduce
// (reference: cpp/include/cudf/scalar/scalar.hpp:41-330).  Value and validity flag both live in
// device memory so a reduction can finish asynchronously; value()/is_valid() synchronise.
#pragma once
#include <cudf/types.hpp>
#include <cudf/utilities/default_stream.hpp>
#include <cudf/utilities/error.hpp>
#include <cudf/utilities/memory_resource.hpp>
#include <rmm/device_buffer.hpp>

namespace cudf {

class scalar {
 public:
  scalar()                         = delete;
  virtual ~scalar()                = default;
  scalar& operator=(scalar const&) = delete;
  scalar& operator=(scalar&&)      = delete;

  [[nodiscard]] data_type type() const noexcept { return _type; }
  void set_valid_async(bool is_valid, rmm::cuda_stream_view stream = cudf::get_default_stream());
  [[nodiscard]] bool is_valid(rmm::cuda_stream_view stream = cudf::get_default_stream()) const;
  // device address of the value bytes of a fixed-width scalar (NULL for other scalars): what type-erased callers
  // such as groupby::shift's fill values read
  [[nodiscard]] virtual void const* device_value_ptr() const noexcept { return nullptr; }
  bool* validity_data() { return static_cast<bool*>(_is_valid.data()); }
  [[nodiscard]] bool const* validity_data() const { return static_cast<bool const*>(_is_valid.data()); }

 protected:
  data_type _type{type_id::EMPTY};
  rmm::device_buffer _is_valid;  // one byte

  scalar(scalar&& other) = default;
  scalar(data_type type, bool is_valid = false, rmm::cuda_stream_view stream = cudf::get_default_stream(),
         rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());
};

namespace detail {

struct uninitialized_value_t {};

template <typename T>
class fixed_width_scalar : public scalar {
 public:
  using value_type = T;
  ~fixed_width_scalar() override             = default;
  fixed_width_scalar(fixed_width_scalar&&)   = default;

  void set_value(T value, rmm::cuda_stream_view stream = cudf::get_default_stream())
  {
    CUDF_CUDA_TRY(hipMemcpyAsync(_data.data(), &value, sizeof(T), hipMemcpyHostToDevice, stream.value()));
    stream.synchronize();  // `value` is a stack temporary
    set_valid_async(true, stream);
  }
  [[nodiscard]] T value(rmm::cuda_stream_view stream = cudf::get_default_stream()) const
  {
    T v{};
    CUDF_CUDA_TRY(hipMemcpyAsync(&v, _data.data(), sizeof(T), hipMemcpyDeviceToHost, stream.value()));
    stream.synchronize();
    return v;
  }
  T* data() { return static_cast<T*>(_data.data()); }
  [[nodiscard]] T const* data() const { return static_cast<T const*>(_data.data()); }
  [[nodiscard]] void const* device_value_ptr() const noexcept override { return _data.data(); }

 protected:
  rmm::device_buffer _data;

  fixed_width_scalar(T value, bool is_valid, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
    : scalar(data_type{type_to_id<T>()}, is_valid, stream, mr), _data{&value, sizeof(T), stream, mr}
  {
    stream.synchronize();  // `value` is a parameter: the copy must finish before returning
  }
  // value left for the caller to fill on `stream` (reduction results: device-to-device, nothing to wait for)
  fixed_width_scalar(uninitialized_value_t, bool is_valid, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
    : scalar(data_type{type_to_id<T>()}, is_valid, stream, mr), _data{sizeof(T), stream, mr}
  {
  }
};

}  // namespace detail

template <typename T>
class numeric_scalar : public detail::fixed_width_scalar<T> {
 public:
  ~numeric_scalar() override        = default;
  numeric_scalar(numeric_scalar&&)  = default;
  numeric_scalar(T value, bool is_valid = true, rmm::cuda_stream_view stream = cudf::get_default_stream(),
                 rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref())
    : detail::fixed_width_scalar<T>(value, is_valid, stream, mr)
  {
  }
  numeric_scalar(detail::uninitialized_value_t u, bool is_valid, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
    : detail::fixed_width_scalar<T>(u, is_valid, stream, mr)
  {
  }
};

}  // namespace cudf
