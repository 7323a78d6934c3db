// rmm::mr::pool_memory_resource -- the stream-ordered caching arena behind every default `mr` argument of the API
// (reference: the role rmm::mr::pool_memory_resource / cuda_async_memory_resource play for libcudf,
// cpp/include/cudf/utilities/memory_resource.hpp; libcudf never calls cudaMalloc on a hot path).
//
// Why not hipMallocAsync: on ROCm 7.2 a block handed back by the runtime's pool on the null stream was observed to
// lose the pageable host-to-device copy issued right after (tests/cpp dbg_h2d, CUDF_AMD_ALLOC=async reproduces it).
// This arena keeps the ordering argument in plain sight instead: a block is only ever reused (a) on the stream it
// was freed on -- everything queued there before the free runs before anything queued after the re-allocation --
// or (b) on another stream after that stream was made to wait on an event recorded at the free.
#include <rmm/resource_ref.hpp>

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace rmm {
namespace mr {

namespace {
constexpr std::size_t SMALL_GRAIN = 512;                 // below 1 MiB
constexpr std::size_t LARGE_GRAIN = std::size_t{2} << 20;  // from 1 MiB up: 2 MiB, the driver's own page size
inline std::size_t size_class(std::size_t bytes)
{
  std::size_t const g = bytes < (std::size_t{1} << 20) ? SMALL_GRAIN : LARGE_GRAIN;
  return (bytes + g - 1) / g * g;
}
}  // namespace

struct pool_memory_resource::impl {
  struct block {
    void* p;
    hipStream_t stream;  // the stream it was freed on
    hipEvent_t freed;    // recorded on that stream at the free
  };
  std::mutex m;
  std::multimap<std::size_t, block> free_blocks;      // by size class
  std::unordered_map<void*, std::size_t> live;        // pointer -> size class, for blocks handed out
  std::vector<hipEvent_t> spare_events;
  std::size_t cached{0};
  std::atomic<std::size_t> driver_allocs{0};

  hipEvent_t take_event()
  {
    if (!spare_events.empty()) {
      auto e = spare_events.back();
      spare_events.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    return e;
  }
  // give every cached block back to the driver; the caller holds the lock
  void release_locked()
  {
    if (free_blocks.empty()) return;
    (void)hipDeviceSynchronize();
    for (auto& kv : free_blocks) {
      (void)hipFree(kv.second.p);
      if (kv.second.freed) spare_events.push_back(kv.second.freed);
    }
    free_blocks.clear();
    cached = 0;
  }
};

pool_memory_resource::pool_memory_resource() : impl_{new impl} {}
// Cached blocks are NOT returned at destruction: the built-in instance dies during static destruction, when the
// HIP runtime may already be gone; the driver reclaims the memory with the process.
pool_memory_resource::~pool_memory_resource() = default;

void pool_memory_resource::release()
{
  std::lock_guard<std::mutex> lock(impl_->m);
  impl_->release_locked();
}
std::size_t pool_memory_resource::cached_bytes() const noexcept { return impl_->cached; }
std::size_t pool_memory_resource::driver_allocations() const noexcept { return impl_->driver_allocs.load(); }

void* pool_memory_resource::do_allocate(std::size_t bytes, cuda_stream_view stream)
{
  std::size_t const sz = size_class(bytes);
  {
    std::lock_guard<std::mutex> lock(impl_->m);
    // a cached block of this class or a little above it (at most 1/8 wasted): the same stream first, then one whose
    // last use is already over, then any (the stream waits for it)
    std::size_t const limit = sz + sz / 8;
    auto const lo = impl_->free_blocks.lower_bound(sz);
    auto pick     = impl_->free_blocks.end();
    int pick_rank = 3;
    for (auto it = lo; it != impl_->free_blocks.end() && it->first <= limit && pick_rank > 0; ++it) {
      int rank = 2;
      if (it->second.stream == stream.value()) rank = 0;
      else if (it->second.freed == nullptr || hipEventQuery(it->second.freed) == hipSuccess) rank = 1;
      if (rank < pick_rank) {
        pick      = it;
        pick_rank = rank;
      }
    }
    (void)hipGetLastError();  // hipEventQuery reports hipErrorNotReady through the sticky error as well
    if (pick != impl_->free_blocks.end()) {
      auto const b = pick->second;
      auto const c = pick->first;
      if (pick_rank == 2 && hipStreamWaitEvent(stream.value(), b.freed, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipEventSynchronize(b.freed);
      }
      impl_->free_blocks.erase(pick);
      impl_->cached -= c;
      if (b.freed) impl_->spare_events.push_back(b.freed);
      impl_->live.emplace(b.p, c);
      return b.p;
    }
  }
  void* p = nullptr;
  if (hipMalloc(&p, sz) != hipSuccess) {
    (void)hipGetLastError();
    {
      std::lock_guard<std::mutex> lock(impl_->m);
      impl_->release_locked();  // out of memory with blocks in the cache: hand them back and try once more
    }
    if (hipMalloc(&p, sz) != hipSuccess) {
      (void)hipGetLastError();
      throw std::bad_alloc();
    }
  }
  impl_->driver_allocs.fetch_add(1);
  std::lock_guard<std::mutex> lock(impl_->m);
  impl_->live.emplace(p, sz);
  return p;
}

void pool_memory_resource::do_deallocate(void* p, std::size_t, cuda_stream_view stream) noexcept
{
  std::lock_guard<std::mutex> lock(impl_->m);
  auto const it = impl_->live.find(p);
  if (it == impl_->live.end()) {  // not ours (a buffer adopted from elsewhere): the conservative path
    (void)hipStreamSynchronize(stream.value());
    (void)hipFree(p);
    return;
  }
  std::size_t const c = it->second;
  impl_->live.erase(it);
  hipEvent_t e = impl_->take_event();
  if (e == nullptr || hipEventRecord(e, stream.value()) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipStreamSynchronize(stream.value());  // no event: make the block safe for everybody now
    if (e) impl_->spare_events.push_back(e);
    e = nullptr;
  }
  impl_->free_blocks.emplace(c, impl::block{p, stream.value(), e});
  impl_->cached += c;
}

namespace {
std::atomic<device_memory_resource*> g_user_default{nullptr};
device_memory_resource* builtin_default()
{
  // leaked on purpose: see ~pool_memory_resource
  static pool_memory_resource* pool      = new pool_memory_resource;
  static hip_async_memory_resource* hipp = new hip_async_memory_resource;
  static hip_memory_resource* plain      = new hip_memory_resource;
  static int const choice                = [] {
    char const* e = std::getenv("CUDF_AMD_ALLOC");
    if (e == nullptr) return 0;
    return e[0] == 'p' && e[1] == 'l' ? 2 : (e[0] == 'a' ? 1 : 0);
  }();
  return choice == 2 ? static_cast<device_memory_resource*>(plain)
                     : (choice == 1 ? static_cast<device_memory_resource*>(hipp) : static_cast<device_memory_resource*>(pool));
}
}  // namespace

device_memory_resource* get_default_resource()
{
  auto* u = g_user_default.load(std::memory_order_acquire);
  return u ? u : builtin_default();
}
device_memory_resource* set_default_resource(device_memory_resource* r)
{
  auto* prev = g_user_default.exchange(r, std::memory_order_acq_rel);
  return prev ? prev : builtin_default();
}

}  // namespace mr
}  // namespace rmm
