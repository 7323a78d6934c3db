// The ledger behind cudf_amd::check_device_faults: status words of queued sorts, copied to pinned memory by their own stream.
// reference behaviour matched: cudf::sort returns once its work is queued (cpp/src/sort/sort.cu:52-89); device errors are sticky and
// reach the caller at a later checking call (cpp/include/cudf/utilities/error.hpp:63-86).
#include "common.hpp"

#include <cudf_amd/device_faults.hpp>

#include <mutex>
#include <vector>

namespace cudf {
namespace detail {
namespace {

struct Slot {
  int* pinned    = nullptr;  // hipHostMalloc: the stream writes the sort's status word here
  hipEvent_t ev  = nullptr;  // recorded behind that copy
  bool busy      = false;
};

class Ledger {
 public:
  // queue "status word of the sort that used `tmp` -> pinned slot" on `stream`; never waits unless every slot is in flight
  void post(void const* tmp, rmm::cuda_stream_view stream)
  {
    std::lock_guard<std::mutex> g(m_);
    reap(false);
    Slot* s = nullptr;
    for (auto& c : slots_)
      if (!c.busy) {
        s = &c;
        break;
      }
    if (!s) {
      if (slots_.size() < kMaxSlots) {
        Slot n;
        CUDF_CUDA_TRY(hipHostMalloc(reinterpret_cast<void**>(&n.pinned), sizeof(int), hipHostMallocDefault));
        CUDF_CUDA_TRY(hipEventCreateWithFlags(&n.ev, hipEventDisableTiming));
        slots_.push_back(n);
        s = &slots_.back();
      } else {  // 256 sorts in flight and none finished: wait for the oldest (the only blocking path; a caller that far ahead of the
                // device loses nothing by it)
        CUDF_CUDA_TRY(hipEventSynchronize(slots_[next_wait_ % slots_.size()].ev));
        ++next_wait_;
        reap(false);
        for (auto& c : slots_)
          if (!c.busy) {
            s = &c;
            break;
          }
        CUDF_EXPECTS(s != nullptr, "device fault ledger: no slot became free");
      }
    }
    *s->pinned = 0;
    gx_check(gx_sort_status_async(tmp, s->pinned, gxs(stream)), "gx_sort_status_async");
    CUDF_CUDA_TRY(hipEventRecord(s->ev, stream.value()));
    s->busy = true;
  }

  // number of faults reported by sorts that have completed since the last call (and forget them)
  int take()
  {
    std::lock_guard<std::mutex> g(m_);
    reap(false);
    int const f = faults_;
    faults_     = 0;
    return f;
  }

 private:
  void reap(bool)
  {
    for (auto& c : slots_) {
      if (!c.busy) continue;
      if (hipEventQuery(c.ev) != hipSuccess) continue;  // not run yet (hipErrorNotReady)
      if (*c.pinned == 5) ++faults_;                     // only 5 is a fault: 3 means the LSD passes produced a correct output
      c.busy = false;
    }
  }
  static constexpr std::size_t kMaxSlots = 256;
  std::mutex m_;
  std::vector<Slot> slots_;
  std::size_t next_wait_ = 0;
  int faults_            = 0;
};

Ledger& ledger()
{
  static Ledger* l = new Ledger;  // leaked on purpose: hipHostFree / hipEventDestroy at static-destruction time race the runtime's own teardown
  return *l;
}

}  // namespace

void post_sort_status(rmm::device_buffer const& tmp, rmm::cuda_stream_view stream) { ledger().post(tmp.data(), stream); }

void throw_pending_sort_faults()
{
  int const f = ledger().take();
  if (f > 0)
    throw cudf::cuda_error{"radix sort: a look-back wait made no progress and was abandoned (device-side fault) in " + std::to_string(f) +
                             " earlier sort(s); their results are not sorted -- the process and its HIP context are intact, call the sort again",
                           GX_EINTERNAL};
}

}  // namespace detail
}  // namespace cudf

namespace cudf_amd {

void check_device_faults(rmm::cuda_stream_view stream)
{
  stream.synchronize();
  cudf::detail::throw_pending_sort_faults();
}

void poll_device_faults() { cudf::detail::throw_pending_sort_faults(); }

}  // namespace cudf_amd
