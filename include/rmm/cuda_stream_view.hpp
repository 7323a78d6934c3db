// rmm::cuda_stream_view shim: the stream type that appears in every libcudf signature
// (reference: rapidsai/rmm, not vendored; usage e.g. cpp/include/cudf/sorting.hpp:44-49).
// On MI355X it wraps a hipStream_t; the CUDA-flavoured name is kept so call sites compile unchanged.
#pragma once
#include <hip/hip_runtime_api.h>

#include <stdexcept>
#include <string>

namespace rmm {

class cuda_stream_view {
 public:
  constexpr cuda_stream_view() = default;
  constexpr cuda_stream_view(hipStream_t s) noexcept : stream_{s} {}
  [[nodiscard]] constexpr hipStream_t value() const noexcept { return stream_; }
  constexpr operator hipStream_t() const noexcept { return stream_; }
  [[nodiscard]] bool is_default() const noexcept { return stream_ == nullptr; }
  void synchronize() const
  {
    auto const e = hipStreamSynchronize(stream_);
    if (e != hipSuccess) throw std::runtime_error(std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
  }
  void synchronize_no_throw() const noexcept { (void)hipStreamSynchronize(stream_); }

 private:
  hipStream_t stream_{nullptr};
};

static constexpr cuda_stream_view cuda_stream_default{};
inline bool operator==(cuda_stream_view a, cuda_stream_view b) { return a.value() == b.value(); }
inline bool operator!=(cuda_stream_view a, cuda_stream_view b) { return !(a == b); }

}  // namespace rmm
