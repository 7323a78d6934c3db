// cudf_api_bench -- what a CALLER of the C++ surface pays: cudf::sort, cudf::hash_join::inner_join and
// cudf::groupby::groupby::aggregate timed wall-clock through include/cudf/*.hpp with the default (pooled) memory
// resource, on the same synthetic workloads bench.py times through the C ABI with pre-allocated scratch
// (reference harnesses: cpp/benchmarks/sort/sort.cpp:12-44, cpp/benchmarks/join/join.cu, cpp/benchmarks/groupby/group_sum.cpp).
// bench.py --through-cpp runs this binary and reports the ratio to its own C-ABI numbers.
// Usage: cudf_api_bench [rows=1e9] [steps=5] [warmup=2] [build_rows=rows/10]      -> one JSON line on stdout
#include <cudf/aggregation.hpp>
#include <cudf/column/column_factories.hpp>
#include <cudf/groupby.hpp>
#include <cudf/join/hash_join.hpp>
#include <cudf/sorting.hpp>
#include <cudf/table/table.hpp>
#include <cudf/table/table_view.hpp>
#include <cudf_amd/gx.h>

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

using namespace cudf;

#define HIP_OK(x)                                                                     \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      std::fprintf(stderr, "hip error %d at %s:%d\n", (int)e_, __FILE__, __LINE__);   \
      std::exit(2);                                                                   \
    }                                                                                 \
  } while (0)

static double now_ms()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static std::unique_ptr<column> random_column(type_id t, size_type n, uint64_t seed, int64_t lo, int64_t hi)
{
  auto c = make_fixed_width_column(data_type{t}, n, mask_state::UNALLOCATED);
  if (gx_fill_random(static_cast<int>(t), c->mutable_view().head<void>(), n, seed, lo, hi, nullptr) != 0) std::exit(3);
  return c;
}

template <typename F>
static double time_steps(int warmup, int steps, F&& step)
{
  for (int i = 0; i < warmup; ++i) step();
  HIP_OK(hipDeviceSynchronize());
  double const t0 = now_ms();
  for (int i = 0; i < steps; ++i) step();
  HIP_OK(hipDeviceSynchronize());
  return (now_ms() - t0) / steps;
}

int main(int argc, char** argv)
{
  auto const rows   = static_cast<size_type>(argc > 1 ? std::atof(argv[1]) : 1e9);
  int const steps   = argc > 2 ? std::atoi(argv[2]) : 5;
  int const warmup  = argc > 3 ? std::atoi(argv[3]) : 2;
  auto const nbuild = static_cast<size_type>(argc > 4 ? std::atof(argv[4]) : rows / 10.0);
  auto* pool        = dynamic_cast<rmm::mr::pool_memory_resource*>(rmm::mr::get_default_resource());

  // ---- cudf::sort of one int64 column (keys only)
  double sort_ms = 0;
  {
    auto keys = random_column(type_id::INT64, rows, 42, 0, 0);
    table_view tv{{keys->view()}};
    sort_ms = time_steps(warmup, steps, [&] { auto out = cudf::sort(tv); });
  }
  // ---- cudf::sorted_order of the same kind of column (what the reference's own sort benchmark times, benchmarks/sort/sort.cpp:16-58)
  double order_ms = 0;
  {
    auto keys = random_column(type_id::INT64, rows, 43, 0, 0);
    table_view tv{{keys->view()}};
    order_ms = time_steps(warmup, steps, [&] { auto out = cudf::sorted_order(tv); });
  }
  // ---- hash_join on SURVEY 8d's key distribution, as bench.py's line: build = nbuild DISTINCT random 64-bit keys, mix64(p(i)) with
  //      p a bijection of [0, nbuild) and mix64 the splitmix64 finalizer (a bijection of the 64-bit integers); probe = mix64(u), u
  //      uniform in [0, nbuild / 0.3): a probe row finds a build key with probability 0.3, the rest are random keys outside the set
  double build_ms = 0, probe_ms = 0;
  std::size_t pairs = 0;
  {
    auto mix64 = [](uint64_t x) {
      x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
      x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
      return x ^ (x >> 31);
    };
    std::vector<int64_t> hb(static_cast<std::size_t>(nbuild));
    // i -> i * 2654435761 mod nbuild is a bijection whenever nbuild has no prime factor besides 2 and 5 (1e8 has none)
    for (std::size_t i = 0; i < hb.size(); ++i) {
      uint64_t const p = (static_cast<uint64_t>(i) * 2654435761ull) % static_cast<uint64_t>(nbuild);
      hb[i]            = static_cast<int64_t>(mix64(p));
    }
    auto bk = make_fixed_width_column(data_type{type_id::INT64}, nbuild, mask_state::UNALLOCATED);
    HIP_OK(hipMemcpy(bk->mutable_view().head<void>(), hb.data(), hb.size() * 8, hipMemcpyHostToDevice));
    auto pk = random_column(type_id::INT64, rows, 67890, 0, static_cast<int64_t>(nbuild / 0.3));
    if (gx_mix64_inplace(pk->mutable_view().head<uint64_t>(), rows, nullptr) != 0) std::exit(3);
    HIP_OK(hipDeviceSynchronize());
    table_view bt{{bk->view()}}, pt{{pk->view()}};
    std::unique_ptr<hash_join> hj;
    {  // steady state: the loop below builds the new table while the old one is still alive, so the arena must hold two
       // tables' worth of blocks before the clock starts (a first-time hipMalloc of 4.3 GB has been seen to take 0.7 s)
      auto a = std::make_unique<hash_join>(bt, null_equality::EQUAL);
      auto b = std::make_unique<hash_join>(bt, null_equality::EQUAL);
      HIP_OK(hipDeviceSynchronize());
    }
    build_ms = time_steps(1, 2, [&] { hj = std::make_unique<hash_join>(bt, null_equality::EQUAL); });
    if (std::getenv("CUDF_API_BENCH_TRACE")) {  // per-call view of the build: host time of the call, then time until idle
      for (int i = 0; i < 6; ++i) {
        auto const a0 = pool ? pool->driver_allocations() : 0;
        double const t0 = now_ms();
        if (i >= 3) hj.reset();  // the last three: free the old table before building the new one
        hj = std::make_unique<hash_join>(bt, null_equality::EQUAL);
        double const t1 = now_ms();
        HIP_OK(hipDeviceSynchronize());
        double const t2 = now_ms();
        std::fprintf(stderr, "build %d: call %.3f ms, until idle %.3f ms, driver allocations +%zu\n", i, t1 - t0, t2 - t1,
                     (pool ? pool->driver_allocations() : 0) - a0);
      }
    }
    probe_ms = time_steps(warmup, steps, [&] {
      auto res = hj->inner_join(pt);
      pairs    = res.first->size();
    });
  }
  // ---- groupby(int32 key, 1e6 groups).agg(f64 sum, count)
  double groupby_ms = 0;
  size_type groups  = 0;
  {
    auto gk = random_column(type_id::INT32, rows, 7, 0, 1000000);
    auto gv = random_column(type_id::FLOAT64, rows, 8, 0, 0);
    groupby::groupby gb{table_view{{gk->view()}}};
    groupby_ms = time_steps(warmup, steps, [&] {
      std::vector<groupby::aggregation_request> reqs(1);
      reqs[0].values = gv->view();
      reqs[0].aggregations.emplace_back(make_sum_aggregation<groupby_aggregation>());
      reqs[0].aggregations.emplace_back(make_count_aggregation<groupby_aggregation>());
      auto res = gb.aggregate(reqs);
      groups   = res.first->num_rows();
    });
  }
  std::printf(
    "{\"rows\": %lld, \"build_rows\": %lld, \"steps\": %d, \"warmup\": %d, \"sort_ms\": %.4f, \"sorted_order_ms\": %.4f, \"join_build_ms\": %.4f, "
    "\"join_probe_ms\": %.4f, \"join_pairs\": %zu, \"groupby_ms\": %.4f, \"groups\": %lld, \"memory_resource\": \"%s\", "
    "\"driver_allocations\": %zu, \"cached_bytes\": %zu}\n",
    (long long)rows, (long long)nbuild, steps, warmup, sort_ms, order_ms, build_ms, probe_ms, pairs, groupby_ms, (long long)groups,
    pool ? "pool_memory_resource" : "other", pool ? pool->driver_allocations() : 0, pool ? pool->cached_bytes() : 0);
  return 0;
}
