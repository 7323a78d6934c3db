// cudf_amd/device_faults.hpp -- where a device-side protocol fault of a queued sort surfaces.
//
// cudf::sort / cudf::sorted_order return as soon as their work is queued, like the reference's (cpp/src/sort/sort.cu:52-89): the
// status word of the sort (gx_sort_status: 5 = a look-back wait was abandoned, the output is not sorted) is copied into pinned host
// memory BY THE STREAM, behind the sort, and looked at later -- the way a CUDA sticky error reaches a caller of the reference: at
// the next call that checks (cpp/include/cudf/utilities/error.hpp:63-86).  A fault is reported
//   * by the next cudf::sort / sorted_order / sort_by_key call of the process, once the faulting sort has run, or
//   * by check_device_faults(stream), which waits for `stream` first -- call it before trusting a sort's output on the host when
//     no other cudf call follows;
// as cudf::cuda_error (a std::runtime_error: the caller's arguments were fine), error code GX_EINTERNAL.  The process and its HIP
// context survive, and the same sort succeeds when called again.
#pragma once
#include <cudf/utilities/default_stream.hpp>
#include <rmm/cuda_stream_view.hpp>

namespace cudf_amd {

// waits for everything queued on `stream`, then throws cudf::cuda_error if any sort that has completed reported a fault
void check_device_faults(rmm::cuda_stream_view stream = cudf::get_default_stream());

// the non-waiting form: looks only at sorts that have already run
void poll_device_faults();

}  // namespace cudf_amd
