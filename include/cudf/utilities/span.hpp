// cudf/utilities/span.hpp -- non-owning (pointer, size) views of host and device arrays
// (reference: cpp/include/cudf/utilities/span.hpp: host_span / device_span; the subset the join API needs).
#pragma once
#include <cstddef>
#include <vector>

namespace cudf {

template <typename T>
class device_span {
 public:
  constexpr device_span() noexcept = default;
  constexpr device_span(T* data, std::size_t size) noexcept : _data{data}, _size{size} {}
  template <typename C>
  device_span(C& c) : _data{c.data()}, _size{c.size()}  // rmm::device_uvector and friends
  {
  }
  template <typename C>
  device_span(C const& c) : _data{c.data()}, _size{c.size()}
  {
  }
  [[nodiscard]] constexpr T* data() const noexcept { return _data; }
  [[nodiscard]] constexpr std::size_t size() const noexcept { return _size; }
  [[nodiscard]] constexpr bool empty() const noexcept { return _size == 0; }

 private:
  T* _data{nullptr};
  std::size_t _size{0};
};

template <typename T>
class host_span {
 public:
  constexpr host_span() noexcept = default;
  constexpr host_span(T* data, std::size_t size) noexcept : _data{data}, _size{size} {}
  template <typename U>
  host_span(std::vector<U> const& v) : _data{v.data()}, _size{v.size()}
  {
  }
  template <typename U>
  host_span(std::vector<U>& v) : _data{v.data()}, _size{v.size()}
  {
  }
  [[nodiscard]] constexpr T* data() const noexcept { return _data; }
  [[nodiscard]] constexpr std::size_t size() const noexcept { return _size; }
  [[nodiscard]] constexpr T& operator[](std::size_t i) const { return _data[i]; }
  [[nodiscard]] constexpr T* begin() const noexcept { return _data; }
  [[nodiscard]] constexpr T* end() const noexcept { return _data + _size; }

 private:
  T* _data{nullptr};
  std::size_t _size{0};
};

}  // namespace cudf
