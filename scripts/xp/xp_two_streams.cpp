// xp_two_streams.cpp -- measurement program (NOT product code): two cudf::sort calls on two streams, issued back to back from one host
// thread, for a rocprofv3 --kernel-trace run ("a C++ case that overlaps two sorts on two streams and shows overlap in a
// rocprof trace"; the pass / fail form of the same thing is the last case of tests/cpp/cudf_api_tests.cpp).  scripts/overlap_summary.py turns
// the trace into: per queue the first start / last end of its kernels, and the time during which kernels of BOTH queues were running.
// The reference returns from cudf::sort once the work is queued (cpp/src/sort/sort.cu:52-89).
//   g++ -std=c++20 -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include scripts/xp/xp_two_streams.cpp -Lcudf_amd -lcudf -lcudf_amd \
//       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../../../cudf_amd' -Wl,-rpath,/opt/rocm/lib -o scripts/xp/bin/xp_two_streams
#include <cudf/column/column.hpp>
#include <cudf/sorting.hpp>
#include <cudf/table/table_view.hpp>
#include <cudf_amd/device_faults.hpp>

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

using namespace cudf;

static std::unique_ptr<column> make_col(std::vector<int64_t> const& v)
{
  rmm::device_buffer data{v.data(), v.size() * sizeof(int64_t), get_default_stream()};
  get_default_stream().synchronize();
  return std::make_unique<column>(data_type{type_id::INT64}, static_cast<size_type>(v.size()), std::move(data), rmm::device_buffer{}, 0);
}

int main(int argc, char** argv)
{
  std::size_t const N = std::size_t{1} << (argc > 1 ? std::atoi(argv[1]) : 27);  // rows per stream (2^27: the cursor path, ~1.5 ms per sort)
  hipStream_t s1, s2;
  if (hipStreamCreateWithFlags(&s1, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) return 2;
  std::vector<int64_t> k(N);
  uint64_t x = 0x9E3779B97F4A7C15ull;
  for (auto& v : k) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = static_cast<int64_t>(x); }
  auto a = make_col(k);
  for (auto& v : k) v = ~v;
  auto b = make_col(k);
  table_view ta{{a->view()}}, tb{{b->view()}};
  { auto w1 = cudf::sort(ta, {}, {}, s1); auto w2 = cudf::sort(tb, {}, {}, s2); }  // warm-up: arena blocks of both streams, module load
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  for (int rep = 0; rep < 3; ++rep) {
    auto const t0 = std::chrono::steady_clock::now();
    auto ra = cudf::sort(ta, {}, {}, s1);
    auto rb = cudf::sort(tb, {}, {}, s2);
    auto const t1 = std::chrono::steady_clock::now();
    (void)hipStreamSynchronize(s1);
    (void)hipStreamSynchronize(s2);
    auto const t2 = std::chrono::steady_clock::now();
    cudf_amd::poll_device_faults();
    std::printf("rep %d: %zu rows per stream; issue of both calls %.3f ms, until both streams drained %.3f ms\n", rep, N,
                std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t0).count());
    // one sort alone, for the comparison
    auto const u0 = std::chrono::steady_clock::now();
    auto rc = cudf::sort(ta, {}, {}, s1);
    (void)hipStreamSynchronize(s1);
    auto const u1 = std::chrono::steady_clock::now();
    std::printf("rep %d: one sort alone %.3f ms\n", rep, std::chrono::duration<double, std::milli>(u1 - u0).count());
  }
  (void)hipStreamDestroy(s1);
  (void)hipStreamDestroy(s2);
  return 0;
}
