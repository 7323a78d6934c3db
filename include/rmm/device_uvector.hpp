// rmm::device_uvector<T> shim: typed uninitialised device vector (join gather maps:
// cpp/include/cudf/join/join.hpp:160-166 return unique_ptr<device_uvector<size_type>> pairs).
#pragma once
#include <rmm/device_buffer.hpp>

namespace rmm {

template <typename T>
class device_uvector {
 public:
  using value_type = T;
  using size_type  = std::size_t;
  device_uvector(std::size_t n, cuda_stream_view stream, device_async_resource_ref mr = mr::get_default_resource())
    : buf_{n * sizeof(T), stream, mr}
  {
  }
  device_uvector(device_uvector&&) noexcept            = default;
  device_uvector& operator=(device_uvector&&) noexcept = default;
  [[nodiscard]] T* data() noexcept { return static_cast<T*>(buf_.data()); }
  [[nodiscard]] T const* data() const noexcept { return static_cast<T const*>(buf_.data()); }
  [[nodiscard]] T* begin() noexcept { return data(); }
  [[nodiscard]] T const* begin() const noexcept { return data(); }
  [[nodiscard]] T* end() noexcept { return data() + size(); }
  [[nodiscard]] T const* end() const noexcept { return data() + size(); }
  [[nodiscard]] std::size_t size() const noexcept { return buf_.size() / sizeof(T); }
  [[nodiscard]] bool is_empty() const noexcept { return size() == 0; }
  [[nodiscard]] cuda_stream_view stream() const noexcept { return buf_.stream(); }
  void shrink(std::size_t n) { buf_.resize_down(n * sizeof(T)); }
  device_buffer release() noexcept { return std::move(buf_); }

 private:
  device_buffer buf_;
};

}  // namespace rmm
