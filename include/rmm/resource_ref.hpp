// rmm::device_async_resource_ref shim: stream-ordered device allocation, the allocator type of
// every libcudf signature (reference usage: cpp/include/cudf/sorting.hpp:48).
// Default resource = mr::pool_memory_resource, a stream-ordered caching arena sized for 288 GB of HBM3E:
// nothing caps it, blocks go back to the driver only when an allocation fails.
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

// Stream-ordered caching arena over hipMalloc (defined in libcudf.so, cudf_amd/cpp/src/memory_resource.cpp): a
// freed block goes to a free list tagged with the stream it was freed on and an event recorded there; the same
// stream takes it back with no synchronisation at all (stream order), another stream first waits on the event
// (hipStreamWaitEvent, asynchronous).  hipMalloc / hipFree -- both of which synchronise the device -- are left
// only on the cold path: a first-time size, or an out-of-memory retry after the cache is emptied.
class pool_memory_resource final : public device_memory_resource {
 public:
  pool_memory_resource();
  ~pool_memory_resource() override;
  pool_memory_resource(pool_memory_resource const&)            = delete;
  pool_memory_resource& operator=(pool_memory_resource const&) = delete;
  // return every cached block to the driver (synchronises the device)
  void release();
  [[nodiscard]] std::size_t cached_bytes() const noexcept;
  [[nodiscard]] std::size_t driver_allocations() const noexcept;  // hipMalloc calls so far

 private:
  void* do_allocate(std::size_t bytes, cuda_stream_view stream) override;
  void do_deallocate(void* p, std::size_t bytes, cuda_stream_view stream) noexcept override;
  struct impl;
  impl* impl_;
};

// The process-wide default: the caching arena.  CUDF_AMD_ALLOC=plain selects hipMalloc / hipFree with a stream
// synchronisation per free, CUDF_AMD_ALLOC=async the HIP runtime's own stream-ordered pool (hipMallocAsync).
device_memory_resource* get_default_resource();
device_memory_resource* set_default_resource(device_memory_resource* r);  // returns the previous one; nullptr restores the built-in

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
