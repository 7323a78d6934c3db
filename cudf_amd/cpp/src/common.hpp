// common.hpp -- helpers shared by the host library (libcudf.so): everything here only validates,
// allocates and forwards to the C ABI of the kernel layer (include/cudf_amd/gx.h).
#pragma once
#include <cudf/column/column.hpp>
#include <cudf/column/column_view.hpp>
#include <cudf/types.hpp>
#include <cudf/utilities/error.hpp>
#include <cudf_amd/gx.h>
#include <rmm/device_buffer.hpp>
#include <rmm/device_uvector.hpp>

#include <cstdint>
#include <memory>
#include <string>

namespace cudf {
namespace detail {

inline gx_stream_t gxs(rmm::cuda_stream_view s) { return reinterpret_cast<gx_stream_t>(s.value()); }

// negative gx_error -> logic_error, positive hipError -> cuda_error (error.hpp:63-86)
inline void gx_check(int rc, char const* what)
{
  if (rc == 0) return;
  if (rc > 0) throw cudf::cuda_error{std::string{what} + ": HIP error " + std::to_string(rc), rc};
  if (rc == GX_EDTYPE) throw cudf::data_type_error{std::string{what} + ": unsupported element type"};
  throw cudf::logic_error{std::string{what} + ": kernel layer rejected the call (" + std::to_string(rc) + ")"};
}

// type_id values of the supported fixed-width types equal gx_dtype by construction (gx.h)
inline int gx_type(data_type t)
{
  auto const id = static_cast<int>(t.id());
  CUDF_EXPECTS(id >= GX_INT8 && id <= GX_BOOL8, "Only fixed-width numeric columns are supported on this path",
               cudf::data_type_error);
  return id;
}

// data pointer of row 0 of the view (offset applied)
inline void const* row0(column_view const& c)
{
  return static_cast<char const*>(c.head<void>()) + static_cast<std::size_t>(c.offset()) * size_of(c.type());
}

// Run a gx entry point that follows the scratch-query convention:  f(tmp, &bytes)
template <typename F>
rmm::device_buffer run_with_scratch(F&& f, char const* what, rmm::cuda_stream_view stream)
{
  std::size_t bytes = 0;
  gx_check(f(nullptr, &bytes), what);
  rmm::device_buffer tmp{bytes ? bytes : 1, stream};
  gx_check(f(tmp.data(), &bytes), what);
  return tmp;
}

// Sorts are stream-ordered (round 6): the device-side status word of a sort's scratch (5 = a look-back wait was
// abandoned, gx_sort.hip spin_guard) is copied to pinned host memory BY THE STREAM behind the sort (post_sort_status: no wait) and
// examined by a later call -- throw_pending_sort_faults at the entry of the next sort and at this layer's own synchronisation points,
// cudf_amd::check_device_faults(stream) on demand (src/device_faults.cpp, include/cudf_amd/device_faults.hpp).  cudf::cuda_error, not
// logic_error (a device runtime fault is not a caller error); only status 5 is a fault.  SortFaultMode opts the sorts issued
// inside its scope into the recoverable form; every other caller of the C-ABI sorts keeps the trap.
void post_sort_status(rmm::device_buffer const& tmp, rmm::cuda_stream_view stream);
void throw_pending_sort_faults();
struct SortFaultMode {
  SortFaultMode() { gx_sort_set_fault_mode(1); }
  ~SortFaultMode() { gx_sort_set_fault_mode(0); }
  SortFaultMode(SortFaultMode const&)            = delete;
  SortFaultMode& operator=(SortFaultMode const&) = delete;
};

// read one int64 from the device (synchronises the stream)
inline int64_t read_i64(int64_t const* dev, rmm::cuda_stream_view stream)
{
  int64_t h = 0;
  CUDF_CUDA_TRY(hipMemcpyAsync(&h, dev, sizeof(h), hipMemcpyDeviceToHost, stream.value()));
  stream.synchronize();
  throw_pending_sort_faults();  // a synchronisation point of this layer: sorts queued before it have run
  return h;
}

// A validity bitmap usable by the kernels for `c`: the kernels index bit i for row i, so a view
// with a non-zero offset gets its bits re-based into a fresh bitmap (`holder` keeps it alive).
bitmask_type const* rebased_mask(column_view const& c, rmm::device_buffer& holder, rmm::cuda_stream_view stream);

}  // namespace detail
}  // namespace cudf
