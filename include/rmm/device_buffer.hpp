// rmm::device_buffer shim: untyped owning device allocation (column data / null masks).
#pragma once
#include <rmm/cuda_stream_view.hpp>
#include <rmm/resource_ref.hpp>

#include <cstddef>
#include <stdexcept>
#include <utility>

namespace rmm {

class device_buffer {
 public:
  device_buffer() : mr_{mr::get_default_resource()} {}
  device_buffer(std::size_t size, cuda_stream_view stream,
                device_async_resource_ref mr = mr::get_default_resource())
    : stream_{stream}, mr_{mr}
  {
    allocate(size);
  }
  // copy `size` bytes from host or device memory
  device_buffer(void const* src, std::size_t size, cuda_stream_view stream,
                device_async_resource_ref mr = mr::get_default_resource())
    : stream_{stream}, mr_{mr}
  {
    allocate(size);
    if (size) {
      if (hipMemcpyAsync(data_, src, size, hipMemcpyDefault, stream.value()) != hipSuccess)
        throw std::runtime_error("device_buffer: hipMemcpyAsync failed");
    }
  }
  device_buffer(device_buffer const& o, cuda_stream_view stream,
                device_async_resource_ref mr = mr::get_default_resource())
    : device_buffer(o.data_, o.size_, stream, mr)
  {
  }
  device_buffer(device_buffer&& o) noexcept
    : data_{o.data_}, size_{o.size_}, capacity_{o.capacity_}, stream_{o.stream_}, mr_{o.mr_}
  {
    o.data_ = nullptr;
    o.size_ = o.capacity_ = 0;
  }
  device_buffer& operator=(device_buffer&& o) noexcept
  {
    if (this != &o) {
      release();
      data_ = o.data_; size_ = o.size_; capacity_ = o.capacity_; stream_ = o.stream_; mr_ = o.mr_;
      o.data_ = nullptr; o.size_ = o.capacity_ = 0;
    }
    return *this;
  }
  device_buffer(device_buffer const&)            = delete;
  device_buffer& operator=(device_buffer const&) = delete;
  ~device_buffer() { release(); }

  [[nodiscard]] void* data() noexcept { return data_; }
  [[nodiscard]] void const* data() const noexcept { return data_; }
  [[nodiscard]] std::size_t size() const noexcept { return size_; }
  [[nodiscard]] std::size_t capacity() const noexcept { return capacity_; }
  [[nodiscard]] bool is_empty() const noexcept { return size_ == 0; }
  [[nodiscard]] cuda_stream_view stream() const noexcept { return stream_; }
  void set_stream(cuda_stream_view s) noexcept { stream_ = s; }
  // shrink the logical size (no reallocation); used by join outputs sized after the probe
  void resize_down(std::size_t new_size) { if (new_size <= capacity_) size_ = new_size; else throw std::length_error("resize_down"); }

 private:
  void allocate(std::size_t size)
  {
    size_ = capacity_ = size;
    data_ = size ? mr_.allocate_async(size, stream_) : nullptr;
  }
  void release() noexcept
  {
    if (data_) mr_.deallocate_async(data_, capacity_, stream_);
    data_ = nullptr;
    size_ = capacity_ = 0;
  }
  void* data_{nullptr};
  std::size_t size_{0};
  std::size_t capacity_{0};
  cuda_stream_view stream_{};
  device_async_resource_ref mr_;
};

}  // namespace rmm
