// rmm::device_async_resource_ref shim: stream-ordered device allocation, the allocator type of
// every libcudf signature (reference usage: cpp/include/cudf/sorting.hpp:48).
// Default resource = hipMallocAsync / hipFreeAsync (the HIP stream-ordered pool), sized for
// 288 GB of HBM3E: nothing here caps the pool.
#pragma once
#include <rmm/cuda_stream_view.hpp>

#include <cstddef>
#include <cstdlib>
#include <new>

namespace rmm {
namespace mr {

class device_memory_resource {
 public:
  virtual ~device_memory_resource() = default;
  void* allocate(std::size_t bytes, cuda_stream_view stream) { return bytes ? do_allocate(bytes, stream) : nullptr; }
  void deallocate(void* p, std::size_t bytes, cuda_stream_view stream) noexcept
  {
    if (p) do_deallocate(p, bytes, stream);
  }

 private:
  virtual void* do_allocate(std::size_t bytes, cuda_stream_view stream)            = 0;
  virtual void do_deallocate(void* p, std::size_t bytes, cuda_stream_view stream) noexcept = 0;
};

// stream-ordered HIP pool
class hip_async_memory_resource final : public device_memory_resource {
  void* do_allocate(std::size_t bytes, cuda_stream_view stream) override
  {
    void* p = nullptr;
    if (hipMallocAsync(&p, bytes, stream.value()) != hipSuccess) {
      (void)hipGetLastError();
      if (hipMalloc(&p, bytes) != hipSuccess) throw std::bad_alloc();  // -> MemoryError in Python
    }
    return p;
  }
  void do_deallocate(void* p, std::size_t, cuda_stream_view stream) noexcept override
  {
    if (hipFreeAsync(p, stream.value()) != hipSuccess) (void)hipFree(p);
  }
};

// plain hipMalloc / hipFree; deallocation waits for the stream so the block cannot be reused under
// work that is still in flight
class hip_memory_resource final : public device_memory_resource {
  void* do_allocate(std::size_t bytes, cuda_stream_view) override
  {
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) throw std::bad_alloc();
    return p;
  }
  void do_deallocate(void* p, std::size_t, cuda_stream_view stream) noexcept override
  {
    (void)hipStreamSynchronize(stream.value());
    (void)hipFree(p);
  }
};

// CUDF_AMD_ALLOC=async selects the stream-ordered pool (hipMallocAsync); the default is plain
// hipMalloc: on ROCm 7.2 blocks recycled by the stream-ordered pool were observed to lose
// host-to-device copies issued right after re-allocation (tests/cpp reproduces it with the pool).
inline device_memory_resource* get_default_resource()
{
  static hip_async_memory_resource pool;
  static hip_memory_resource plain;
  static bool const use_pool = [] {
    char const* e = std::getenv("CUDF_AMD_ALLOC");
    return e != nullptr && e[0] == 'a';
  }();
  return use_pool ? static_cast<device_memory_resource*>(&pool) : static_cast<device_memory_resource*>(&plain);
}

}  // namespace mr

class device_async_resource_ref {
 public:
  device_async_resource_ref(mr::device_memory_resource* r) : r_{r} {}
  device_async_resource_ref(mr::device_memory_resource& r) : r_{&r} {}
  void* allocate_async(std::size_t bytes, cuda_stream_view s) { return r_->allocate(bytes, s); }
  void deallocate_async(void* p, std::size_t bytes, cuda_stream_view s) noexcept { r_->deallocate(p, bytes, s); }
  [[nodiscard]] mr::device_memory_resource* resource() const noexcept { return r_; }

 private:
  mr::device_memory_resource* r_;
};

}  // namespace rmm
