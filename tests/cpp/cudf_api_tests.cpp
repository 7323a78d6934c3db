// C++ API parity tests: drive the drop-in through the reference's own interface (cudf::sort,
// cudf::hash_join, cudf::groupby::groupby, cudf::reduce, cudf::scan ...) with the literal vectors of
// the reference's gtest suites (file:line cited per case).  No gtest in this image: a minimal
// harness.  Needs a GPU; run by tests/test_gpu_cpp_api.py.
#include <map>
#include <tuple>
#include <cudf/aggregation.hpp>
#include <cudf/column/column_factories.hpp>
#include <cudf/copying.hpp>
#include <cudf/groupby.hpp>
#include <cudf/hashing.hpp>
#include <cudf/interop.hpp>
#include <cudf/join/distinct_hash_join.hpp>
#include <cudf/join/filtered_join.hpp>
#include <cudf/join/hash_join.hpp>
#include <cudf/join/join.hpp>
#include <cudf/reduction.hpp>
#include <cudf/sorting.hpp>
#include <cudf_amd/gx.h>  // gx_sequence_i32: a device fill for the allocator test
#include <chrono>
#include <cudf_amd/device_faults.hpp>
#include <cudf_amd/gx.h>
#include <cudf_amd/gx_knobs.h>  // the look-back fault hooks of the last case

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <functional>
#include <limits>
#include <numeric>
#include <random>
#include <cudf/partitioning.hpp>

#include <thread>
#include <cudf_amd/distributed.hpp>

#include <algorithm>
#include <string>
#include <vector>

using namespace cudf;
static int g_failed = 0, g_run = 0;
#define CHECK(cond)                                                                   \
  do {                                                                                \
    if (!(cond)) {                                                                    \
      std::printf("    CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond);         \
      throw std::runtime_error("check failed");                                       \
    }                                                                                 \
  } while (0)

template <typename T>
std::unique_ptr<column> make_col(std::vector<T> const& v, std::vector<int> const& valid = {})
{
  auto const n = static_cast<size_type>(v.size());
  rmm::device_buffer data{v.data(), v.size() * sizeof(T), get_default_stream()};
  rmm::device_buffer mask{};
  size_type nulls = 0;
  if (!valid.empty()) {
    std::vector<bitmask_type> w(bitmask_allocation_size_bytes(n) / 4, 0u);
    for (size_type i = 0; i < n; ++i) {
      if (valid[i]) w[i / 32] |= 1u << (i % 32); else ++nulls;
    }
    mask = rmm::device_buffer{w.data(), w.size() * 4, get_default_stream()};
  }
  get_default_stream().synchronize();
  return std::make_unique<column>(data_type{type_to_id<T>()}, n, std::move(data), std::move(mask), nulls);
}

template <typename T>
std::vector<T> to_host(column_view const& c)
{
  std::vector<T> h(c.size());
  if (c.size()) (void)hipMemcpy(h.data(), c.data<T>(), h.size() * sizeof(T), hipMemcpyDeviceToHost);
  return h;
}
std::vector<int> valid_host(column_view const& c)
{
  std::vector<int> v(c.size(), 1);
  if (!c.nullable()) return v;
  std::vector<bitmask_type> w(num_bitmask_words(c.size() + c.offset()));
  (void)hipMemcpy(w.data(), c.null_mask(), w.size() * 4, hipMemcpyDeviceToHost);
  for (size_type i = 0; i < c.size(); ++i) v[i] = (w[(i + c.offset()) / 32] >> ((i + c.offset()) % 32)) & 1;
  return v;
}
template <typename T>
std::vector<T> to_host(rmm::device_uvector<T> const& c)
{
  std::vector<T> h(c.size());
  if (c.size()) (void)hipMemcpy(h.data(), c.data(), h.size() * sizeof(T), hipMemcpyDeviceToHost);
  return h;
}
using pairs_t = std::vector<std::pair<int, int>>;
pairs_t sorted_pairs(join_result const& r)
{
  auto l = to_host(*r.first);
  auto rr = to_host(*r.second);
  pairs_t p;
  for (std::size_t i = 0; i < l.size(); ++i) p.emplace_back(l[i], rr[i]);
  std::sort(p.begin(), p.end());
  return p;
}
template <typename Exc, typename F>
bool throws(F&& f)
{
  try {
    f();
  } catch (Exc const&) {
    return true;
  } catch (...) {
    return false;
  }
  return false;
}
void run(char const* name, std::function<void()> f)
{
  ++g_run;
  try {
    f();
    std::printf("[ OK ] %s\n", name);
  } catch (std::exception const& e) {
    ++g_failed;
    std::printf("[FAIL] %s: %s\n", name, e.what());
  }
}


static void dbg_h2d(char const* where)
{
  std::vector<int32_t> kv(300000);
  for (std::size_t i = 0; i < kv.size(); ++i) kv[i] = (int32_t)(i % 5000) + 1;
  auto c  = make_col<int32_t>(kv);
  auto h0 = to_host<int32_t>(c->view());
  std::size_t bad = 0;
  for (std::size_t i = 0; i < kv.size(); ++i) bad += h0[i] != kv[i];
  std::printf("  dbg[%s] make_col: bad %zu first %d ptr %p\n", where, bad, h0[0], c->view().head<void>());
  (void)hipMemcpy(const_cast<void*>(c->view().head<void>()), kv.data(), kv.size() * 4, hipMemcpyHostToDevice);
  h0  = to_host<int32_t>(c->view());
  bad = 0;
  for (std::size_t i = 0; i < kv.size(); ++i) bad += h0[i] != kv[i];
  std::printf("  dbg[%s] after sync hipMemcpy: bad %zu\n", where, bad);
  rmm::device_buffer raw{kv.data(), kv.size() * 4, get_default_stream()};
  get_default_stream().synchronize();
  std::vector<int32_t> h1(kv.size());
  (void)hipMemcpy(h1.data(), raw.data(), kv.size() * 4, hipMemcpyDeviceToHost);
  bad = 0;
  for (std::size_t i = 0; i < kv.size(); ++i) bad += h1[i] != kv[i];
  std::printf("  dbg[%s] raw device_buffer: bad %zu ptr %p\n", where, bad, raw.data());
}

static void on_segv(int sig)
{
  void* frames[64];
  int const n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, STDERR_FILENO);
  _exit(128 + sig);
}

int main()
{
  setvbuf(stdout, nullptr, _IONBF, 0);
  signal(SIGSEGV, on_segv);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    std::printf("no GPU\n");
    return 77;
  }
  if (getenv("CUDF_AMD_DEBUG")) dbg_h2d("start");
  constexpr double NaN = std::numeric_limits<double>::quiet_NaN();
  constexpr double Inf = std::numeric_limits<double>::infinity();

  // ---- sort: cpp/tests/sort/stable_sort_tests.cpp:88-123
  run("sorted_order single column no nulls (stable_sort_tests.cpp:88-104)", [] {
    auto c = make_col<int32_t>({7, 1, -2, 5, 1, 0, 1, -2, 0, 5});
    auto o = stable_sorted_order(table_view{{c->view()}});
    CHECK((to_host<int32_t>(o->view()) == std::vector<int32_t>{2, 7, 5, 8, 1, 4, 6, 3, 9, 0}));
    auto u = make_col<uint32_t>({7, 1, (uint32_t)-2, 5, 1, 0, 1, (uint32_t)-2, 0, 5});
    auto o2 = sorted_order(table_view{{u->view()}});
    CHECK((to_host<int32_t>(o2->view()) == std::vector<int32_t>{5, 8, 1, 4, 6, 3, 9, 0, 2, 7}));
    CHECK(o->type().id() == type_id::INT32 && !o->nullable());
  });
  run("sorted_order with nulls BEFORE (stable_sort_tests.cpp:106-123)", [] {
    auto c = make_col<int64_t>({7, 1, -2, 5, 1, 0, 1, -2, 0, 5}, {1, 1, 0, 0, 1, 0, 1, 0, 1, 0});
    auto o = stable_sorted_order(table_view{{c->view()}}, {order::ASCENDING}, {null_order::BEFORE});
    // the reference compares the GATHERED table (run_stable_sort_test :20-31): null rows are interchangeable
    auto h = to_host<int32_t>(o->view());
    std::vector<int32_t> nulls(h.begin(), h.begin() + 5), rest(h.begin() + 5, h.end());
    std::sort(nulls.begin(), nulls.end());
    CHECK((nulls == std::vector<int32_t>{2, 3, 5, 7, 9}));
    CHECK((rest == std::vector<int32_t>{8, 1, 4, 6, 0}));
  });
  run("sorted_order floats: NaN / Inf / -0.0 (sort_test.cpp:1071-1083)", [&] {
    auto c = make_col<double>({-0.0, -NaN, -NaN, NaN, Inf, -Inf, 7, 5, 6, NaN, Inf, -Inf, -NaN, -NaN, -0.0});
    auto o = sorted_order(table_view{{c->view()}});
    CHECK((to_host<int32_t>(o->view()) == std::vector<int32_t>{5, 11, 0, 14, 7, 8, 6, 4, 10, 1, 2, 3, 9, 12, 13}));
  });
  run("multi-column sorted_order {ASC, DESC} (sort_test.cpp:148-175, numeric columns)", [] {
    auto c1 = make_col<int32_t>({5, 4, 3, 5, 8});
    auto c3 = make_col<int32_t>({10, 40, 70, 5, 2});
    auto o  = sorted_order(table_view{{c1->view(), c3->view()}}, {order::ASCENDING, order::DESCENDING});
    CHECK((to_host<int32_t>(o->view()) == std::vector<int32_t>{2, 1, 0, 3, 4}));
  });
  run("multi-column with nulls AFTER (sort_test.cpp:50-84, numeric columns)", [] {
    auto c1 = make_col<int32_t>({5, 4, 3, 5, 8, 5}, {1, 1, 0, 1, 1, 1});
    auto c3 = make_col<int32_t>({10, 40, 70, 5, 2, 10}, {1, 1, 0, 1, 1, 1});
    auto o  = sorted_order(table_view{{c1->view(), c3->view()}}, {order::ASCENDING, order::DESCENDING},
                           {null_order::AFTER, null_order::AFTER});
    // col1 asc: 4(1) 5(0,3,5) 8(4) null(2); among the 5s col3 desc: 10(0),10(5),5(3) -> stable 0,5,3
    CHECK((to_host<int32_t>(o->view()) == std::vector<int32_t>{1, 0, 5, 3, 4, 2}));
  });
  run("sort / sort_by_key / error messages (sort.cu:31-67, sort_impl.cuh:45)", [] {
    std::mt19937_64 rng(1);
    std::vector<int64_t> v(100003);
    for (auto& x : v) x = (int64_t)rng();
    auto c   = make_col<int64_t>(v);
    auto out = sort(table_view{{c->view()}}, {order::DESCENDING});
    std::sort(v.begin(), v.end(), std::greater<int64_t>());
    CHECK(to_host<int64_t>(out->view().column(0)) == v);
    auto k  = make_col<int32_t>({3, 1, 2});
    auto pv = make_col<double>({30., 10., 20.});
    auto sb = sort_by_key(table_view{{pv->view()}}, table_view{{k->view()}});
    CHECK((to_host<double>(sb->view().column(0)) == std::vector<double>{10., 20., 30.}));
    auto k2 = make_col<int32_t>({3, 1});
    CHECK(throws<std::logic_error>([&] { sort_by_key(table_view{{pv->view()}}, table_view{{k2->view()}}); }));
    CHECK(throws<std::logic_error>([&] { sorted_order(table_view{{k->view()}}, {order::ASCENDING, order::ASCENDING}); }));
    auto e = sorted_order(table_view{});
    CHECK(e->size() == 0 && e->type().id() == type_id::INT32);
  });

  // ---- join: cpp/tests/join/join_tests.cpp:1163-1237, 2040-2123 (single key column forms)
  run("inner_join / left_join / full_join single key (join_tests.cpp:1163-1237)", [] {
    auto l = make_col<int32_t>({3, 1, 2, 0, 2});
    auto r = make_col<int32_t>({2, 2, 0, 4, 3});
    auto ij = inner_join(table_view{{l->view()}}, table_view{{r->view()}});
    CHECK((sorted_pairs(ij) == pairs_t{{0, 4}, {2, 0}, {2, 1}, {3, 2}, {4, 0}, {4, 1}}));
    auto lj = left_join(table_view{{l->view()}}, table_view{{r->view()}});
    CHECK((sorted_pairs(lj) == pairs_t{{0, 4}, {1, JoinNoMatch}, {2, 0}, {2, 1}, {3, 2}, {4, 0}, {4, 1}}));
    auto fj = full_join(table_view{{l->view()}}, table_view{{r->view()}});
    CHECK((sorted_pairs(fj) == pairs_t{{JoinNoMatch, 3}, {0, 4}, {1, JoinNoMatch}, {2, 0}, {2, 1}, {3, 2}, {4, 0}, {4, 1}}));
  });
  run("hash_join object: sequential probes, sizes, errors (join_tests.cpp:2040-2123; hash_join.cu:49-58)", [] {
    auto b = make_col<int64_t>({2, 2, 0, 4, 3});
    for (double lf : {0.5, 1.0}) {
      hash_join hj{table_view{{b->view()}}, nullable_join::NO, null_equality::EQUAL, lf};
      auto p1 = make_col<int64_t>({3, 1, 2, 0, 2});
      CHECK(hj.inner_join_size(table_view{{p1->view()}}) == 6);
      CHECK(sorted_pairs(hj.inner_join(table_view{{p1->view()}})).size() == 6);
      auto p2 = make_col<int64_t>({9, 9, 9, 0, 3});
      CHECK((sorted_pairs(hj.inner_join(table_view{{p2->view()}}, 2)) == pairs_t{{3, 2}, {4, 4}}));
      CHECK(hj.left_join_size(table_view{{p2->view()}}) == 5);
      CHECK(hj.full_join_size(table_view{{p2->view()}}) == 8);
      auto pn = make_col<int64_t>({1, 2}, {1, 0});
      CHECK(throws<std::invalid_argument>([&] { (void)hj.inner_join(table_view{{pn->view()}}); }));
      auto pt = make_col<int32_t>({1, 2});
      CHECK(throws<cudf::data_type_error>([&] { (void)hj.inner_join(table_view{{pt->view()}}); }));
    }
    CHECK(throws<std::invalid_argument>([&] { hash_join hj{table_view{}, null_equality::EQUAL}; }));
    CHECK(throws<std::invalid_argument>([&] { hash_join hj{table_view{{b->view()}}, nullable_join::NO, null_equality::EQUAL, 0.0}; }));
  });
  run("join on nulls, both null_equality values (join_tests.cpp:1421-...)", [] {
    auto l = make_col<int32_t>({1, 2, 0, 7}, {1, 1, 0, 1});
    auto r = make_col<int32_t>({2, 0, 9, 0}, {1, 0, 1, 0});
    auto eq = inner_join(table_view{{l->view()}}, table_view{{r->view()}}, null_equality::EQUAL);
    CHECK((sorted_pairs(eq) == pairs_t{{1, 0}, {2, 1}, {2, 3}}));
    auto ne = inner_join(table_view{{l->view()}}, table_view{{r->view()}}, null_equality::UNEQUAL);
    CHECK((sorted_pairs(ne) == pairs_t{{1, 0}}));
    auto e  = make_col<int32_t>({});
    auto em = inner_join(table_view{{l->view()}}, table_view{{e->view()}});
    CHECK(em.first->size() == 0 && em.second->size() == 0);
    auto le = left_join(table_view{{l->view()}}, table_view{{e->view()}});
    CHECK((sorted_pairs(le) == pairs_t{{0, JoinNoMatch}, {1, JoinNoMatch}, {2, JoinNoMatch}, {3, JoinNoMatch}}));
  });
  run("large join: build on the smaller side, pair orientation kept (join.cu:49-59)", [] {
    std::mt19937 rng(3);
    std::vector<int64_t> big(200000), small(5000);
    for (auto& x : big) x = rng() % 20000;
    for (std::size_t i = 0; i < small.size(); ++i) small[i] = (int64_t)i * 3;
    auto cb = make_col<int64_t>(big);
    auto cs = make_col<int64_t>(small);
    auto a  = sorted_pairs(inner_join(table_view{{cs->view()}}, table_view{{cb->view()}}));  // left = small
    std::size_t expect = 0;
    for (auto x : big) expect += (x % 3 == 0 && x / 3 < 5000);
    CHECK(a.size() == expect);
    for (auto const& p : a) CHECK(small[p.first] == big[p.second]);
  });

  // ---- groupby: cpp/tests/groupby/{sum,count,mean}_tests.cpp, sum_scan_tests.cpp
  auto by_key = [](table const& keys, std::vector<std::unique_ptr<column>> const& res) {
    auto k = to_host<int32_t>(keys.view().column(0));
    std::vector<int> idx(k.size());
    std::iota(idx.begin(), idx.end(), 0);
    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return k[a] < k[b]; });
    return std::make_pair(k, idx);
  };
  run("filtered_join semi / anti (semi_anti_join_tests.cpp:119-136,357-421)", [] {
    auto l = make_col<int32_t>({0, 1, 2});
    auto r = make_col<int32_t>({0, 1, 3});
    filtered_join fj{table_view{{r->view()}}, null_equality::EQUAL, get_default_stream()};
    CHECK((to_host(*fj.semi_join(table_view{{l->view()}})) == std::vector<size_type>{0, 1}));
    CHECK((to_host(*fj.anti_join(table_view{{l->view()}})) == std::vector<size_type>{2}));
    // empty right table (no columns): semi -> nothing, anti -> every left row; empty left -> nothing
    filtered_join fe{table_view{}, null_equality::EQUAL, get_default_stream()};
    CHECK(fe.semi_join(table_view{{l->view()}})->size() == 0);
    CHECK((to_host(*fe.anti_join(table_view{{l->view()}})) == std::vector<size_type>{0, 1, 2}));
    auto e = make_col<int32_t>({});
    CHECK(fj.semi_join(table_view{{e->view()}})->size() == 0);
    CHECK(fj.anti_join(table_view{{e->view()}})->size() == 0);
    CHECK(throws<std::invalid_argument>([&] { filtered_join bad{table_view{{r->view()}}, null_equality::EQUAL, 0.0, get_default_stream()}; }));
    // nulls: a null left row matches a null right row only under EQUAL
    auto ln = make_col<int64_t>({5, 7, 9, 1}, {1, 0, 1, 1});
    auto rn = make_col<int64_t>({9, 4, 5}, {1, 0, 1});
    filtered_join feq{table_view{{rn->view()}}, null_equality::EQUAL, get_default_stream()};
    CHECK((to_host(*feq.semi_join(table_view{{ln->view()}})) == std::vector<size_type>{0, 1, 2}));
    CHECK((to_host(*feq.anti_join(table_view{{ln->view()}})) == std::vector<size_type>{3}));
    filtered_join fne{table_view{{rn->view()}}, null_equality::UNEQUAL, get_default_stream()};
    CHECK((to_host(*fne.semi_join(table_view{{ln->view()}})) == std::vector<size_type>{0, 2}));
    CHECK((to_host(*fne.anti_join(table_view{{ln->view()}})) == std::vector<size_type>{1, 3}));
    // two key columns
    auto l0 = make_col<int32_t>({3, 1, 2, 0, 3}), l1 = make_col<int32_t>({0, 1, 2, 4, 1});
    auto r0 = make_col<int32_t>({2, 2, 0, 4, 3}), r1 = make_col<int32_t>({1, 0, 1, 2, 1});
    filtered_join f2{table_view{{r0->view(), r1->view()}}, null_equality::EQUAL, get_default_stream()};
    CHECK((to_host(*f2.semi_join(table_view{{l0->view(), l1->view()}})) == std::vector<size_type>{4}));
    CHECK((to_host(*f2.anti_join(table_view{{l0->view(), l1->view()}})) == std::vector<size_type>{0, 1, 2, 3}));
  });
  run("distinct_hash_join (distinct_join_tests.cpp:74-93,185-225,541-575,614-649)", [] {
    // IntegerInnerJoin: right = 0..2023, left = 0,2,...,4046
    std::vector<int32_t> rv(2024), lv(2024);
    std::iota(rv.begin(), rv.end(), 0);
    for (int i = 0; i < 2024; ++i) lv[i] = 2 * i;
    auto r = make_col<int32_t>(rv);
    auto l = make_col<int32_t>(lv);
    distinct_hash_join dj{table_view{{r->view()}}};
    auto p = sorted_pairs(dj.inner_join(table_view{{l->view()}}));
    CHECK(p.size() == 1012);
    for (int i = 0; i < 1012; ++i) CHECK(p[i].first == i && p[i].second == 2 * i);
    // PrimitiveLeftJoinNoNulls: two int32 key columns
    auto l0 = make_col<int32_t>({3, 1, 2, 0, 3}), l1 = make_col<int32_t>({0, 1, 2, 4, 1});
    auto r0 = make_col<int32_t>({2, 2, 0, 4, 3}), r1 = make_col<int32_t>({1, 0, 1, 2, 1});
    distinct_hash_join d2{table_view{{r0->view(), r1->view()}}};
    CHECK((to_host(*d2.left_join(table_view{{l0->view(), l1->view()}})) ==
           std::vector<size_type>{JoinNoMatch, JoinNoMatch, JoinNoMatch, JoinNoMatch, 4}));
    // PrimitiveLeftJoinWithNulls: the left second key is null at row 2
    auto m0 = make_col<int32_t>({3, 1, 2, 0, 2}), m1 = make_col<int32_t>({1, 1, -1, 4, 0}, {1, 1, 0, 1, 1});
    CHECK((to_host(*d2.left_join(table_view{{m0->view(), m1->view()}})) ==
           std::vector<size_type>{4, JoinNoMatch, JoinNoMatch, JoinNoMatch, 1}));
    // PrimitiveInnerJoinNoNulls: three int32 key columns (12 bytes: the dictionary encoding)
    auto a0 = make_col<int32_t>({1, 2, 3, 4, 5}), a1 = make_col<int32_t>({0, 0, 3, 4, 5}), a2 = make_col<int32_t>({9, 9, 9, 9, 9});
    auto b0 = make_col<int32_t>({1, 2, 3, 4, 9}), b1 = make_col<int32_t>({0, 0, 0, 4, 4}), b2 = make_col<int32_t>({9, 9, 9, 0, 9});
    distinct_hash_join d3{table_view{{a0->view(), a1->view(), a2->view()}}};
    CHECK((sorted_pairs(d3.inner_join(table_view{{b0->view(), b1->view(), b2->view()}})) == pairs_t{{0, 0}, {1, 1}}));
    CHECK(throws<std::invalid_argument>([&] { distinct_hash_join bad{table_view{{r->view()}}, null_equality::EQUAL, 1.5}; }));
    CHECK(throws<std::invalid_argument>([&] { distinct_hash_join bad{table_view{}}; }));
  });
  run("left / full join, one nullable key column, nulls on BOTH sides (hash_join.cu:77-84; ADVICE r1)", [] {
    // EQUAL: null left rows 1 and 4 pair with null right rows 0 and 3 and do NOT also appear as (row, NoMatch);
    // UNEQUAL: they match nothing.  Same expectation through the direct path (int32), the normalising encoder
    // (float64) and the dictionary encoder (int16).
    auto check = [](auto tag) {
      using T = decltype(tag);
      auto l = make_col<T>({3, 9, 2, 7, 9}, {1, 0, 1, 1, 0});
      auto r = make_col<T>({9, 2, 3, 9, 2}, {0, 1, 1, 0, 1});
      table_view L{{l->view()}}, R{{r->view()}};
      pairs_t const eq{{0, 2}, {1, 0}, {1, 3}, {2, 1}, {2, 4}, {3, JoinNoMatch}, {4, 0}, {4, 3}};
      pairs_t const ne{{0, 2}, {1, JoinNoMatch}, {2, 1}, {2, 4}, {3, JoinNoMatch}, {4, JoinNoMatch}};
      CHECK((sorted_pairs(left_join(L, R, null_equality::EQUAL)) == eq));
      CHECK((sorted_pairs(left_join(L, R, null_equality::UNEQUAL)) == ne));
      CHECK(full_join(L, R, null_equality::EQUAL).first->size() == eq.size());        // every right row is matched
      CHECK(full_join(L, R, null_equality::UNEQUAL).first->size() == ne.size() + 2);  // + null right rows 0 and 3
      hash_join hj{R, null_equality::EQUAL};
      CHECK(hj.left_join_size(L) == eq.size());
      CHECK(hj.inner_join_size(L) == 7);
    };
    check(int32_t{});
    check(double{});
    check(int16_t{});
    check(int64_t{});
  });
  run("partitioned inner / left / full joins + match contexts (join_tests.cpp:3229-3560; hash_join.hpp:259-440)", [] {
    // HashJoinPartitionedInnerJoin's tables (numeric key columns): left {3,1,2,0,2} x {1,1,null,4,1}, right {2,2,0,4,3} x {1,null,1,2,1}
    auto l0 = make_col<int32_t>({3, 1, 2, 0, 2}), l1 = make_col<int32_t>({1, 1, 0, 4, 0}, {1, 1, 0, 1, 1});
    auto r0 = make_col<int32_t>({2, 2, 0, 4, 3}), r1 = make_col<int32_t>({1, 0, 1, 2, 1}, {1, 0, 1, 1, 1});
    table_view L{{l0->view(), l1->view()}}, R{{r0->view(), r1->view()}};
    hash_join hj{R, null_equality::EQUAL};
    auto whole_inner = sorted_pairs(hj.inner_join(L));
    auto whole_left  = sorted_pairs(hj.left_join(L));
    auto whole_full  = sorted_pairs(hj.full_join(L));
    // match counts: inner = matches, left / full = at least one pair per row
    auto ictx = hj.inner_join_match_context(L);
    auto lctx = hj.left_join_match_context(L);
    auto ic   = to_host(*ictx._match_counts);
    auto lc   = to_host(*lctx._match_counts);
    CHECK(ic.size() == 5 && lc.size() == 5);
    std::size_t isum = 0, lsum = 0;
    for (int i = 0; i < 5; ++i) {
      isum += ic[i];
      lsum += lc[i];
      CHECK(lc[i] == std::max(ic[i], 1));
    }
    CHECK(isum == whole_inner.size() && lsum == whole_left.size());
    // row by row, and in chunks of two
    for (int step : {1, 2, 5}) {
      pairs_t inner_all, left_all;
      std::vector<join_result> keep;
      std::vector<device_span<size_type const>> lp, rp;
      join_partition_context pc{std::make_unique<join_match_context>(hj.inner_join_match_context(L)), 0, 0};
      for (int s0 = 0; s0 < 5; s0 += step) {
        pc.left_start_idx = s0;
        pc.left_end_idx   = std::min(5, s0 + step);
        auto pi           = sorted_pairs(hj.partitioned_inner_join(pc));
        inner_all.insert(inner_all.end(), pi.begin(), pi.end());
        auto pl = hj.partitioned_left_join(pc);
        auto sl = sorted_pairs(pl);
        left_all.insert(left_all.end(), sl.begin(), sl.end());
        keep.emplace_back(hj.partitioned_full_join(pc));
        lp.emplace_back(keep.back().first->data(), keep.back().first->size());
        rp.emplace_back(keep.back().second->data(), keep.back().second->size());
      }
      std::sort(inner_all.begin(), inner_all.end());
      std::sort(left_all.begin(), left_all.end());
      CHECK(inner_all == whole_inner);
      CHECK(left_all == whole_left);
      auto fin = hash_join::finalize_partitioned_full_join(lp, rp, 5, 5);
      CHECK(sorted_pairs(fin) == whole_full);
    }
    // single int64 key, larger: chunks of 1000 rows against the whole join
    std::mt19937_64 rng(5);
    std::vector<int64_t> lk(10000), rk(3000);
    for (auto& x : lk) x = static_cast<int64_t>(rng() % 4000);
    for (auto& x : rk) x = static_cast<int64_t>(rng() % 4000);
    auto lcol = make_col<int64_t>(lk), rcol = make_col<int64_t>(rk);
    table_view LL{{lcol->view()}}, RR{{rcol->view()}};
    hash_join big{RR, null_equality::EQUAL};
    auto whole = sorted_pairs(big.inner_join(LL));
    join_partition_context pc{std::make_unique<join_match_context>(big.inner_join_match_context(LL)), 0, 0};
    auto counts = to_host(*pc.left_table_context->_match_counts);
    CHECK(static_cast<std::size_t>(std::accumulate(counts.begin(), counts.end(), int64_t{0})) == whole.size());
    pairs_t all;
    for (int s0 = 0; s0 < 10000; s0 += 1000) {
      pc.left_start_idx = s0;
      pc.left_end_idx   = s0 + 1000;
      auto p            = sorted_pairs(big.partitioned_inner_join(pc));
      for (auto const& pr : p) CHECK(pr.first >= s0 && pr.first < s0 + 1000);
      all.insert(all.end(), p.begin(), p.end());
    }
    std::sort(all.begin(), all.end());
    CHECK(all == whole);
  });
  run("groupby on several int64 key columns at size: one partition pass, rows compared in the LDS tables (gx_groupby_sum_count_wide)", [] {
    // 400 000 rows, keys (i % 371, (7 i) % 113, i % 3): 371 x 113 x 3 tuples, all of them hit; host-side reference in a map
    size_type const n = 400000;
    std::vector<int64_t> a(n), b(n), c3(n), v(n);
    std::map<std::tuple<int64_t, int64_t, int64_t>, std::pair<int64_t, int32_t>> want;
    for (size_type i = 0; i < n; ++i) {
      a[i]  = i % 371 - 100;
      b[i]  = (7LL * i) % 113;
      c3[i] = (i % 3) * (int64_t{1} << 40);
      v[i]  = (i * 31LL) % 1000 - 500;
      auto& w = want[{a[i], b[i], c3[i]}];
      w.first += v[i];
      w.second += 1;
    }
    auto ka = make_col<int64_t>(a), kb = make_col<int64_t>(b), kc = make_col<int64_t>(c3), vals = make_col<int64_t>(v);
    groupby::groupby gb{table_view{{ka->view(), kb->view(), kc->view()}}};
    std::vector<groupby::aggregation_request> reqs(1);
    reqs[0].values = vals->view();
    reqs[0].aggregations.emplace_back(make_sum_aggregation<groupby_aggregation>());
    reqs[0].aggregations.emplace_back(make_count_aggregation<groupby_aggregation>());
    reqs[0].aggregations.emplace_back(make_mean_aggregation<groupby_aggregation>());
    auto [k, res] = gb.aggregate(reqs);
    CHECK(k->num_columns() == 3 && static_cast<std::size_t>(k->num_rows()) == want.size());
    auto h0 = to_host<int64_t>(k->get_column(0).view()), h1 = to_host<int64_t>(k->get_column(1).view()),
         h2 = to_host<int64_t>(k->get_column(2).view());
    auto hs = to_host<int64_t>(res[0].results[0]->view());
    auto hc = to_host<int32_t>(res[0].results[1]->view());
    auto hm = to_host<double>(res[0].results[2]->view());
    std::size_t seen = 0;
    for (std::size_t g = 0; g < h0.size(); ++g) {
      auto it = want.find({h0[g], h1[g], h2[g]});
      CHECK(it != want.end());
      CHECK(it->second.first == hs[g] && it->second.second == hc[g]);
      CHECK(hm[g] == static_cast<double>(hs[g]) / hc[g]);
      it->second.second = -1;  // every group exactly once
      ++seen;
    }
    CHECK(seen == want.size());
    for (auto const& e : want) CHECK(e.second.second == -1);
  });

  run("multi-column join keys (join_tests.cpp:1163-1283,1421-1500: numeric key columns)", [] {
    auto l0 = make_col<int32_t>({3, 1, 2, 0, 2}), l1 = make_col<int32_t>({1, 1, 0, 4, 0});
    auto r0 = make_col<int32_t>({2, 2, 0, 4, 3}), r1 = make_col<int32_t>({1, 0, 1, 2, 1});
    table_view L{{l0->view(), l1->view()}}, R{{r0->view(), r1->view()}};
    CHECK((sorted_pairs(inner_join(L, R)) == pairs_t{{0, 4}, {2, 1}, {4, 1}}));
    CHECK((sorted_pairs(left_join(L, R)) == pairs_t{{0, 4}, {1, JoinNoMatch}, {2, 1}, {3, JoinNoMatch}, {4, 1}}));
    CHECK(full_join(L, R).first->size() == 8);  // 5 left rows + right rows 0, 2, 3 unmatched
    // InnerJoinOnNulls: nulls on both sides of the second key; EQUAL -> 2 pairs, UNEQUAL -> 1
    auto ln = make_col<int32_t>({1, 1, 0, 4, 0}, {1, 1, 0, 1, 1});
    auto rn = make_col<int32_t>({1, 0, 1, 2, 1}, {1, 0, 1, 1, 1});
    table_view LN{{l0->view(), ln->view()}}, RN{{r0->view(), rn->view()}};
    CHECK((sorted_pairs(inner_join(LN, RN, null_equality::EQUAL)) == pairs_t{{0, 4}, {2, 1}}));
    CHECK((sorted_pairs(inner_join(LN, RN, null_equality::UNEQUAL)) == pairs_t{{0, 4}}));
    // wide keys (int64, float64 with NaN / -0.0, int16): 18 bytes per row
    auto w0 = make_col<int64_t>({10, 10, 20, 20, 30});
    auto w1 = make_col<double>({0.0, std::nan(""), 1.5, -0.0, 2.0});
    auto w2 = make_col<int16_t>({1, 2, 3, 4, 5});
    auto v0 = make_col<int64_t>({20, 10, 10, 30});
    auto v1 = make_col<double>({0.0, -std::nan(""), -0.0, 2.5});
    auto v2 = make_col<int16_t>({4, 2, 1, 5});
    table_view W{{w0->view(), w1->view(), w2->view()}}, V{{v0->view(), v1->view(), v2->view()}};
    CHECK((sorted_pairs(inner_join(W, V)) == pairs_t{{0, 2}, {1, 1}, {3, 0}}));   // -0.0 == 0.0, NaN == NaN
    CHECK(throws<cudf::data_type_error>([&] { (void)inner_join(W, table_view{{v0->view(), v1->view(), v0->view()}}); }));
    CHECK(throws<std::invalid_argument>([&] {
      hash_join hj{V, null_equality::EQUAL};
      (void)hj.inner_join(table_view{{w0->view()}});
    }));
  });
  run("groupby key types and multi-column keys (keys_tests.cpp:25-41 typed over int8..double)", [&] {
    auto check_keys = [&](auto tag) {
      using K = decltype(tag);
      auto keys = make_col<K>({1, 2, 3, 1, 2, 2, 1, 3, 3, 2});
      auto vals = make_col<int32_t>({0, 1, 2, 3, 4, 5, 6, 7, 8, 9});
      groupby::groupby gb{table_view{{keys->view()}}};
      std::vector<groupby::aggregation_request> reqs(1);
      reqs[0].values = vals->view();
      reqs[0].aggregations.emplace_back(make_count_aggregation<groupby_aggregation>());
      reqs[0].aggregations.emplace_back(make_sum_aggregation<groupby_aggregation>());
      auto [k, res] = gb.aggregate(reqs);
      auto order    = sorted_order(k->view());
      auto ks       = gather(k->view(), order->view());
      auto cs       = gather(table_view{{res[0].results[0]->view(), res[0].results[1]->view()}}, order->view());
      CHECK((to_host<K>(ks->get_column(0).view()) == std::vector<K>{1, 2, 3}));
      CHECK((to_host<int32_t>(cs->get_column(0).view()) == std::vector<int32_t>{3, 4, 3}));
      CHECK((to_host<int64_t>(cs->get_column(1).view()) == std::vector<int64_t>{9, 19, 17}));
    };
    check_keys(int8_t{});
    check_keys(int16_t{});
    check_keys(int32_t{});
    check_keys(int64_t{});
    check_keys(float{});
    check_keys(double{});
    // two key columns, a null in the second drops the row (null_policy::EXCLUDE)
    auto k0   = make_col<int32_t>({1, 1, 2, 2, 1, 2, 1});
    auto k1   = make_col<int64_t>({7, 8, 7, 7, 7, 9, 8}, {1, 1, 1, 1, 1, 0, 1});
    auto vals = make_col<double>({1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0});
    groupby::groupby gb{table_view{{k0->view(), k1->view()}}};
    std::vector<groupby::aggregation_request> reqs(1);
    reqs[0].values = vals->view();
    reqs[0].aggregations.emplace_back(make_sum_aggregation<groupby_aggregation>());
    reqs[0].aggregations.emplace_back(make_max_aggregation<groupby_aggregation>());
    auto [k, res] = gb.aggregate(reqs);
    CHECK(k->num_columns() == 2 && k->num_rows() == 3);
    auto order = sorted_order(k->view());
    auto ks    = gather(k->view(), order->view());
    auto rs    = gather(table_view{{res[0].results[0]->view(), res[0].results[1]->view()}}, order->view());
    CHECK((to_host<int32_t>(ks->get_column(0).view()) == std::vector<int32_t>{1, 1, 2}));
    CHECK((to_host<int64_t>(ks->get_column(1).view()) == std::vector<int64_t>{7, 8, 7}));
    CHECK((to_host<double>(rs->get_column(0).view()) == std::vector<double>{6.0, 9.0, 7.0}));
    CHECK((to_host<double>(rs->get_column(1).view()) == std::vector<double>{5.0, 7.0, 4.0}));
  });
  run("groupby SUM/COUNT/MEAN basic (sum_tests.cpp:68-80, count_tests.cpp:21-41, mean_tests.cpp:37-56)", [&] {
    auto keys = make_col<int32_t>({1, 2, 3, 1, 2, 2, 1, 3, 3, 2});
    auto vals = make_col<int32_t>({0, 1, 2, 3, 4, 5, 6, 7, 8, 9});
    groupby::groupby gb{table_view{{keys->view()}}};
    std::vector<groupby::aggregation_request> reqs(1);
    reqs[0].values = vals->view();
    reqs[0].aggregations.push_back(make_sum_aggregation<groupby_aggregation>());
    reqs[0].aggregations.push_back(make_count_aggregation<groupby_aggregation>());
    reqs[0].aggregations.push_back(make_mean_aggregation<groupby_aggregation>());
    auto [k, res] = gb.aggregate(reqs);
    auto [kh, idx] = by_key(*k, res[0].results);
    CHECK(kh.size() == 3);
    auto s = to_host<int64_t>(res[0].results[0]->view());
    auto c = to_host<int32_t>(res[0].results[1]->view());
    auto m = to_host<double>(res[0].results[2]->view());
    CHECK(res[0].results[0]->type().id() == type_id::INT64);
    int64_t es[] = {9, 19, 17};
    int32_t ec[] = {3, 4, 3};
    double em[]  = {3., 19. / 4, 17. / 3};
    for (int g = 0; g < 3; ++g) {
      CHECK(kh[idx[g]] == g + 1 && s[idx[g]] == es[g] && c[idx[g]] == ec[g] && m[idx[g]] == em[g]);
    }
  });
  run("groupby with null keys and values (sum_tests.cpp:124-145, count_tests.cpp:103-132, mean_tests.cpp:102-128)", [&] {
    auto keys = make_col<int32_t>({1, 2, 3, 1, 2, 2, 1, 3, 3, 2, 4}, {1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1});
    auto vals = make_col<double>({0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 4}, {0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 0});
    groupby::groupby gb{table_view{{keys->view()}}};
    std::vector<groupby::aggregation_request> reqs(1);
    reqs[0].values = vals->view();
    reqs[0].aggregations.push_back(make_sum_aggregation<groupby_aggregation>());
    reqs[0].aggregations.push_back(make_count_aggregation<groupby_aggregation>());
    reqs[0].aggregations.push_back(make_count_aggregation<groupby_aggregation>(null_policy::INCLUDE));
    reqs[0].aggregations.push_back(make_mean_aggregation<groupby_aggregation>());
    auto [k, res] = gb.aggregate(reqs);
    auto [kh, idx] = by_key(*k, res[0].results);
    CHECK(kh.size() == 4);
    auto s  = to_host<double>(res[0].results[0]->view());
    auto sv = valid_host(res[0].results[0]->view());
    auto cv = to_host<int32_t>(res[0].results[1]->view());
    auto ca = to_host<int32_t>(res[0].results[2]->view());
    auto m  = to_host<double>(res[0].results[3]->view());
    auto mv = valid_host(res[0].results[3]->view());
    double es[] = {9, 14, 10, 0};
    int ev[]    = {1, 1, 1, 0};
    int ecv[]   = {2, 3, 2, 0};
    int eca[]   = {3, 4, 2, 1};
    double em[] = {4.5, 14. / 3, 5., 0};
    for (int g = 0; g < 4; ++g) {
      int i = idx[g];
      CHECK(kh[i] == g + 1 && sv[i] == ev[g] && mv[i] == ev[g] && cv[i] == ecv[g] && ca[i] == eca[g]);
      if (ev[g]) CHECK(s[i] == es[g] && m[i] == em[g]);
    }
    CHECK(res[0].results[0]->null_count() == 1);
  });
  run("groupby edge cases: empty input, all-null keys, size mismatch, int32 sum does not wrap (sum_tests.cpp:82-108,167-188)", [&] {
    auto ek = make_col<int32_t>({});
    auto ev = make_col<int32_t>({});
    groupby::groupby gb0{table_view{{ek->view()}}};
    std::vector<groupby::aggregation_request> r0(1);
    r0[0].values = ev->view();
    r0[0].aggregations.push_back(make_sum_aggregation<groupby_aggregation>());
    auto [k0, res0] = gb0.aggregate(r0);
    CHECK(k0->num_rows() == 0 && res0[0].results[0]->size() == 0 && res0[0].results[0]->type().id() == type_id::INT64);
    auto nk = make_col<int32_t>({1, 2, 3}, {0, 0, 0});
    auto nv = make_col<int32_t>({3, 4, 5});
    groupby::groupby gb1{table_view{{nk->view()}}};
    std::vector<groupby::aggregation_request> r1(1);
    r1[0].values = nv->view();
    r1[0].aggregations.push_back(make_sum_aggregation<groupby_aggregation>());
    auto [k1, res1] = gb1.aggregate(r1);
    CHECK(k1->num_rows() == 0);
    auto sv = make_col<int32_t>({1, 2});
    std::vector<groupby::aggregation_request> r2(1);
    r2[0].values = sv->view();
    r2[0].aggregations.push_back(make_sum_aggregation<groupby_aggregation>());
    CHECK(throws<std::logic_error>([&] { (void)gb1.aggregate(r2); }));
    auto ok = make_col<int32_t>({0, 0});
    auto ov = make_col<int32_t>({std::numeric_limits<int32_t>::min(), std::numeric_limits<int32_t>::min()});
    groupby::groupby gb3{table_view{{ok->view()}}};
    std::vector<groupby::aggregation_request> r3(1);
    r3[0].values = ov->view();
    r3[0].aggregations.push_back(make_sum_aggregation<groupby_aggregation>());
    auto [k3, res3] = gb3.aggregate(r3);
    CHECK(to_host<int64_t>(res3[0].results[0]->view())[0] == -4294967296ll);
  });
  if (getenv("CUDF_AMD_DEBUG")) dbg_h2d("before two requests");
  run("groupby two requests give consistently ordered results", [&] {
    std::mt19937 rng(5);
    std::vector<int32_t> kv(300000);
    std::vector<double> a(kv.size()), b(kv.size());
    for (std::size_t i = 0; i < kv.size(); ++i) {
      kv[i] = rng() % 5000;
      a[i]  = (double)(rng() % 1000);
      b[i]  = (double)(rng() % 7);
    }
    auto keys = make_col<int32_t>(kv);
    { auto h0 = to_host<int32_t>(keys->view()); std::printf("    after make_col: %d %d %d ptr %p\n", h0[0], h0[1], h0[2], keys->view().head<void>()); }
    auto ca = make_col<double>(a);
    auto cb = make_col<double>(b);
    { auto h0 = to_host<int32_t>(keys->view()); std::printf("    after 3 cols: %d %d %d; a ptr %p b ptr %p\n", h0[0], h0[1], h0[2], ca->view().head<void>(), cb->view().head<void>()); }
    groupby::groupby gb{table_view{{keys->view()}}};
    std::vector<groupby::aggregation_request> reqs(2);
    reqs[0].values = ca->view();
    reqs[0].aggregations.push_back(make_sum_aggregation<groupby_aggregation>());
    reqs[1].values = cb->view();
    reqs[1].aggregations.push_back(make_sum_aggregation<groupby_aggregation>());
    auto [k, res] = gb.aggregate(reqs);
    auto kh = to_host<int32_t>(k->view().column(0));
    auto sa = to_host<double>(res[0].results[0]->view());
    auto sb = to_host<double>(res[1].results[0]->view());
    std::vector<double> ea(5000, 0.), eb(5000, 0.);
    for (std::size_t i = 0; i < kv.size(); ++i) {
      ea[kv[i]] += a[i];
      eb[kv[i]] += b[i];
    }
    if (kh.size() != 5000) {
      auto mn = reduce(keys->view(), *make_min_aggregation<reduce_aggregation>(), data_type{type_id::INT32});
      auto mx = reduce(keys->view(), *make_max_aggregation<reduce_aggregation>(), data_type{type_id::INT32});
      std::printf("    groups: %zu sums %zu %zu; key[0]=%d sum=%g; device keys min %d max %d; host kv[0..2]=%d %d %d\n", kh.size(),
                  sa.size(), sb.size(), kh[0], sa[0], static_cast<numeric_scalar<int32_t>&>(*mn).value(),
                  static_cast<numeric_scalar<int32_t>&>(*mx).value(), kv[0], kv[1], kv[2]);
    }
    CHECK(kh.size() == 5000);
    for (std::size_t g = 0; g < kh.size(); ++g) CHECK(sa[g] == ea[kh[g]] && sb[g] == eb[kh[g]]);  // small integers: exact
  });
  run("groupby MIN / MAX incl. null group, mixed with SUM (min_tests.cpp:37-120, max_tests.cpp:37-120)", [&] {
    auto keys = make_col<int32_t>({1, 2, 3, 1, 2, 2, 1, 3, 3, 2});
    auto vals = make_col<double>({0, 1, 2, 3, 4, 5, 6, 7, 8, 9});
    groupby::groupby gb{table_view{{keys->view()}}};
    std::vector<groupby::aggregation_request> reqs(1);
    reqs[0].values = vals->view();
    reqs[0].aggregations.push_back(make_min_aggregation<groupby_aggregation>());
    reqs[0].aggregations.push_back(make_max_aggregation<groupby_aggregation>());
    reqs[0].aggregations.push_back(make_sum_aggregation<groupby_aggregation>());
    auto [k, res] = gb.aggregate(reqs);
    CHECK((to_host<int32_t>(k->view().column(0)) == std::vector<int32_t>{1, 2, 3}));  // two hash passes -> key order
    CHECK((to_host<double>(res[0].results[0]->view()) == std::vector<double>{0, 1, 2}));
    CHECK((to_host<double>(res[0].results[1]->view()) == std::vector<double>{6, 9, 8}));
    CHECK((to_host<double>(res[0].results[2]->view()) == std::vector<double>{9, 19, 17}));
    auto k2 = make_col<int32_t>({1, 2, 3, 1, 2, 2, 1, 3, 3, 2, 4}, {1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1});
    auto v2 = make_col<int32_t>({0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 4}, {0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 0});
    groupby::groupby gb2{table_view{{k2->view()}}};
    std::vector<groupby::aggregation_request> r2(1);
    r2[0].values = v2->view();
    r2[0].aggregations.push_back(make_min_aggregation<groupby_aggregation>());
    r2[0].aggregations.push_back(make_max_aggregation<groupby_aggregation>());
    auto [kk, rr] = gb2.aggregate(r2);
    auto [kh, idx] = by_key(*kk, rr[0].results);
    auto mn = to_host<int32_t>(rr[0].results[0]->view());
    auto mx = to_host<int32_t>(rr[0].results[1]->view());
    auto vv = valid_host(rr[0].results[0]->view());
    int emn[] = {3, 1, 2, 0}, emx[] = {6, 9, 8, 0}, ev[] = {1, 1, 1, 0};
    CHECK(kh.size() == 4 && rr[0].results[0]->type().id() == type_id::INT32);
    for (int g = 0; g < 4; ++g) {
      int i = idx[g];
      CHECK(kh[i] == g + 1 && vv[i] == ev[g]);
      if (ev[g]) CHECK(mn[i] == emn[g] && mx[i] == emx[g]);
    }
  });
  run("groupby VARIANCE / STD / M2 / ARGMIN / ARGMAX (var_tests.cpp:24-137, std_tests.cpp:24-134, argmin_tests.cpp:23-108, argmax_tests.cpp:22-107)", [&] {
    auto run_one = [&](column_view keys, column_view vals, std::unique_ptr<groupby_aggregation> agg) {
      groupby::groupby gb{table_view{{keys}}};
      std::vector<groupby::aggregation_request> reqs(1);
      reqs[0].values = vals;
      reqs[0].aggregations.emplace_back(std::move(agg));
      auto [k, res] = gb.aggregate(reqs);
      auto order    = sorted_order(k->view());
      auto ks       = gather(k->view(), order->view());
      auto rs       = gather(table_view{{res[0].results[0]->view()}}, order->view());
      return std::pair{std::move(ks), std::move(rs)};
    };
    auto near = [](std::vector<double> const& a, std::vector<double> const& b, std::vector<int> const& valid) {
      if (a.size() != b.size()) return false;
      for (std::size_t i = 0; i < a.size(); ++i)
        if (valid[i] && std::fabs(a[i] - b[i]) > 1e-12 * std::max(1.0, std::fabs(b[i]))) return false;
      return true;
    };
    auto keys = make_col<int32_t>({1, 2, 3, 1, 2, 2, 1, 3, 3, 2});
    for (int t = 0; t < 3; ++t) {  // value types int32, int64, double
      std::unique_ptr<column> vals = t == 0   ? make_col<int32_t>({0, 1, 2, 3, 4, 5, 6, 7, 8, 9})
                                     : t == 1 ? make_col<int64_t>({0, 1, 2, 3, 4, 5, 6, 7, 8, 9})
                                              : make_col<double>({0, 1, 2, 3, 4, 5, 6, 7, 8, 9});
      auto [k, r] = run_one(keys->view(), vals->view(), make_variance_aggregation<groupby_aggregation>());
      CHECK((to_host<int32_t>(k->get_column(0).view()) == std::vector<int32_t>{1, 2, 3}));
      CHECK(r->get_column(0).type().id() == type_id::FLOAT64);
      CHECK(near(to_host<double>(r->get_column(0).view()), {9.0, 131.0 / 12, 31.0 / 3}, {1, 1, 1}));
      auto [k2, r2] = run_one(keys->view(), vals->view(), make_std_aggregation<groupby_aggregation>());
      CHECK(near(to_host<double>(r2->get_column(0).view()), {3.0, std::sqrt(131.0 / 12), std::sqrt(31.0 / 3)}, {1, 1, 1}));
      auto [k3, r3] = run_one(keys->view(), vals->view(), make_m2_aggregation<groupby_aggregation>());
      CHECK(near(to_host<double>(r3->get_column(0).view()), {18.0, 131.0 / 4, 62.0 / 3}, {1, 1, 1}));
    }
    // null keys and values; ddof = 1 and 2
    auto kn = make_col<int32_t>({1, 2, 3, 1, 2, 2, 1, 3, 3, 2, 4}, {1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1});
    auto vn = make_col<int64_t>({0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 3}, {0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1});
    {
      auto [k, r] = run_one(kn->view(), vn->view(), make_variance_aggregation<groupby_aggregation>());
      CHECK((to_host<int32_t>(k->get_column(0).view()) == std::vector<int32_t>{1, 2, 3, 4}));
      CHECK((valid_host(r->get_column(0).view()) == std::vector<int>{1, 1, 1, 0}));
      CHECK(near(to_host<double>(r->get_column(0).view()), {4.5, 49.0 / 3, 18.0, 0.0}, {1, 1, 1, 0}));
      auto [k2, r2] = run_one(kn->view(), vn->view(), make_variance_aggregation<groupby_aggregation>(2));
      CHECK((valid_host(r2->get_column(0).view()) == std::vector<int>{0, 1, 0, 0}));
      CHECK(near(to_host<double>(r2->get_column(0).view()), {0.0, 98.0 / 3, 0.0, 0.0}, {0, 1, 0, 0}));
      auto [k3, r3] = run_one(kn->view(), vn->view(), make_std_aggregation<groupby_aggregation>());
      CHECK(near(to_host<double>(r3->get_column(0).view()), {3 / std::sqrt(2.0), 7 / std::sqrt(3.0), 3 * std::sqrt(2.0), 0.0}, {1, 1, 1, 0}));
    }
    // ARGMIN / ARGMAX
    auto va = make_col<int32_t>({9, 8, 7, 6, 5, 4, 3, 2, 1, 0});
    {
      auto [k, r] = run_one(keys->view(), va->view(), make_argmin_aggregation<groupby_aggregation>());
      CHECK(r->get_column(0).type().id() == type_id::INT32);
      CHECK((to_host<int32_t>(r->get_column(0).view()) == std::vector<int32_t>{6, 9, 8}));
      auto [k2, r2] = run_one(keys->view(), va->view(), make_argmax_aggregation<groupby_aggregation>());
      CHECK((to_host<int32_t>(r2->get_column(0).view()) == std::vector<int32_t>{0, 1, 2}));
    }
    {
      auto vb = make_col<double>({9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 4}, {1, 1, 1, 1, 1, 0, 0, 1, 1, 1, 0});
      auto [k, r] = run_one(kn->view(), vb->view(), make_argmin_aggregation<groupby_aggregation>());
      CHECK((to_host<int32_t>(k->get_column(0).view()) == std::vector<int32_t>{1, 2, 3, 4}));
      CHECK((valid_host(r->get_column(0).view()) == std::vector<int>{1, 1, 1, 0}));
      auto h = to_host<int32_t>(r->get_column(0).view());
      CHECK(h[0] == 3 && h[1] == 9 && h[2] == 8);
      auto kx = make_col<int32_t>({1, 2, 3, 1, 2, 2, 1, 3, 3, 2, 4}, {1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1});
      auto vx = make_col<int64_t>({9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 4}, {0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0});
      auto [k2, r2] = run_one(kx->view(), vx->view(), make_argmax_aggregation<groupby_aggregation>());
      CHECK((valid_host(r2->get_column(0).view()) == std::vector<int>{1, 1, 1, 0}));
      auto h2 = to_host<int32_t>(r2->get_column(0).view());
      CHECK(h2[0] == 3 && h2[1] == 4 && h2[2] == 7);
    }
    // all of them in one request, mixed with SUM: every result in the same (key) order
    {
      auto vals = make_col<double>({0, 1, 2, 3, 4, 5, 6, 7, 8, 9});
      groupby::groupby gb{table_view{{keys->view()}}};
      std::vector<groupby::aggregation_request> reqs(1);
      reqs[0].values = vals->view();
      reqs[0].aggregations.emplace_back(make_sum_aggregation<groupby_aggregation>());
      reqs[0].aggregations.emplace_back(make_variance_aggregation<groupby_aggregation>());
      reqs[0].aggregations.emplace_back(make_argmax_aggregation<groupby_aggregation>());
      reqs[0].aggregations.emplace_back(make_sum_of_squares_aggregation<groupby_aggregation>());
      auto [k, res] = gb.aggregate(reqs);
      CHECK((to_host<int32_t>(k->get_column(0).view()) == std::vector<int32_t>{1, 2, 3}));
      CHECK((to_host<double>(res[0].results[0]->view()) == std::vector<double>{9, 19, 17}));
      CHECK(near(to_host<double>(res[0].results[1]->view()), {9.0, 131.0 / 12, 31.0 / 3}, {1, 1, 1}));
      CHECK((to_host<int32_t>(res[0].results[2]->view()) == std::vector<int32_t>{6, 9, 8}));
      CHECK((to_host<double>(res[0].results[3]->view()) == std::vector<double>{45, 123, 117}));
    }
  });
  run("groupby SUM scan (sum_scan_tests.cpp:33-48,118-139)", [&] {
    auto keys = make_col<int32_t>({1, 2, 3, 1, 2, 2, 1, 3, 3, 2});
    auto vals = make_col<int32_t>({0, 1, 2, 3, 4, 5, 6, 7, 8, 9});
    groupby::groupby gb{table_view{{keys->view()}}};
    std::vector<groupby::scan_request> reqs(1);
    reqs[0].values = vals->view();
    reqs[0].aggregations.push_back(make_sum_aggregation<groupby_scan_aggregation>());
    auto [k, res] = gb.scan(reqs);
    CHECK((to_host<int32_t>(k->view().column(0)) == std::vector<int32_t>{1, 1, 1, 2, 2, 2, 2, 3, 3, 3}));
    CHECK((to_host<int64_t>(res[0].results[0]->view()) == std::vector<int64_t>{0, 3, 9, 1, 5, 10, 19, 2, 9, 17}));
    auto k2 = make_col<int32_t>({1, 2, 3, 1, 2, 2, 1, 3, 3, 2, 4}, {1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1});
    auto v2 = make_col<int32_t>({0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 4}, {0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 0});
    groupby::groupby gb2{table_view{{k2->view()}}};
    std::vector<groupby::scan_request> r2(1);
    r2[0].values = v2->view();
    r2[0].aggregations.push_back(make_sum_aggregation<groupby_scan_aggregation>());
    auto [kk, rr] = gb2.scan(r2);
    CHECK((to_host<int32_t>(kk->view().column(0)) == std::vector<int32_t>{1, 1, 1, 2, 2, 2, 2, 3, 3, 4}));
    auto o = to_host<int64_t>(rr[0].results[0]->view());
    auto v = valid_host(rr[0].results[0]->view());
    int64_t e[] = {0, 3, 9, 1, 5, 0, 14, 2, 10, 0};
    int ev[]    = {0, 1, 1, 1, 1, 0, 1, 1, 1, 0};
    for (int i = 0; i < 10; ++i) {
      CHECK(v[i] == ev[i]);
      if (ev[i]) CHECK(o[i] == e[i]);
    }
  });

  run("rank: FIRST / DENSE / MIN / MAX / AVERAGE, keep / top / bottom, percentage (rank_test.cpp:60-430)", [] {
    auto col1 = make_col<int32_t>({5, 4, 3, 5, 8, 5});
    auto col2 = make_col<int32_t>({5, 4, 3, 5, 8, 5}, {1, 1, 0, 1, 1, 1});
    using V   = std::vector<int32_t>;
    using D   = std::vector<double>;
    auto ri   = [](column_view c, rank_method m, order o, null_policy np, null_order no) {
      return to_host<int32_t>(rank(c, m, o, np, no, false)->view());
    };
    auto rd = [](column_view c, rank_method m, order o, null_policy np, null_order no, bool pct) {
      return to_host<double>(rank(c, m, o, np, no, pct)->view());
    };
    auto const A = order::ASCENDING, Dn = order::DESCENDING;
    auto const EX = null_policy::EXCLUDE, IN = null_policy::INCLUDE;
    auto const AF = null_order::AFTER, BE = null_order::BEFORE;
    CHECK((ri(col1->view(), rank_method::FIRST, A, EX, AF) == V{3, 2, 1, 4, 6, 5}));
    {  // keep: the null row stays null, the others are ranked among themselves
      auto r = rank(col2->view(), rank_method::FIRST, A, EX, AF, false);
      CHECK((valid_host(r->view()) == std::vector<int>{1, 1, 0, 1, 1, 1}));
      auto h = to_host<int32_t>(r->view());
      CHECK(h[0] == 2 && h[1] == 1 && h[3] == 3 && h[4] == 5 && h[5] == 4);
    }
    CHECK((ri(col2->view(), rank_method::FIRST, A, IN, BE) == V{3, 2, 1, 4, 6, 5}));   // first_asc_top
    CHECK((ri(col2->view(), rank_method::FIRST, A, IN, AF) == V{2, 1, 6, 3, 5, 4}));   // first_asc_bottom
    CHECK((ri(col1->view(), rank_method::FIRST, Dn, EX, BE) == V{2, 5, 6, 3, 1, 4}));  // first_desc_keep
    CHECK((ri(col1->view(), rank_method::DENSE, A, EX, AF) == V{3, 2, 1, 3, 4, 3}));   // dense_asc_keep
    CHECK((ri(col1->view(), rank_method::MIN, A, EX, AF) == V{3, 2, 1, 3, 6, 3}));     // min_asc_keep
    CHECK((ri(col1->view(), rank_method::MAX, A, EX, AF) == V{5, 2, 1, 5, 6, 5}));     // max_asc_keep
    CHECK((rd(col1->view(), rank_method::AVERAGE, A, EX, AF, false) == D{4, 2, 1, 4, 6, 4}));  // average_asc_keep
    CHECK((rd(col1->view(), rank_method::DENSE, A, EX, AF, true) == D{0.75, 0.5, 0.25, 0.75, 1., 0.75}));  // dense_asc_keep_pct
    CHECK((rd(col1->view(), rank_method::MIN, Dn, EX, BE, true) == D{1.0 / 3.0, 5.0 / 6.0, 1., 1.0 / 3.0, 1.0 / 6.0, 1.0 / 3.0}));
    {  // min_desc_keep_pct with the null row: divided by the 5 ranked rows
      auto r = rank(col2->view(), rank_method::MIN, Dn, EX, BE, true);
      auto h = to_host<double>(r->view());
      CHECK((valid_host(r->view()) == std::vector<int>{1, 1, 0, 1, 1, 1}));
      CHECK(h[0] == 0.4 && h[1] == 1. && h[3] == 0.4 && h[4] == 0.2 && h[5] == 0.4);
    }
    {  // dense percentage with a null: 3 distinct ranked values
      auto h = to_host<double>(rank(col2->view(), rank_method::DENSE, A, EX, AF, true)->view());
      CHECK(h[0] == 2.0 / 3.0 && h[1] == 1.0 / 3.0 && h[4] == 1.);
    }
  });
  run("top_k / top_k_order (top_k.cu:118-165) and segmented sorts (segmented_sort_tests.cpp:70-157)", [] {
    auto c = make_col<int64_t>({7, -3, 12, 5, 12, 0, 9});
    CHECK((to_host<int64_t>(top_k(c->view(), 3)->view()) == std::vector<int64_t>{12, 12, 9}));
    CHECK((to_host<int32_t>(top_k_order(c->view(), 3)->view()) == std::vector<int32_t>{2, 4, 6}));
    CHECK((to_host<int64_t>(top_k(c->view(), 2, order::ASCENDING)->view()) == std::vector<int64_t>{-3, 0}));
    CHECK(top_k(c->view(), 0)->size() == 0 && top_k(c->view(), 100)->size() == 7);
    auto cn = make_col<double>({1.5, 9.0, 2.5, 7.0}, {1, 0, 1, 1});  // the null never makes the top
    CHECK((to_host<double>(top_k(cn->view(), 2)->view()) == std::vector<double>{7.0, 2.5}));
    CHECK(throws<std::invalid_argument>([&] { (void)top_k(c->view(), -1); }));
    // segments             {0   1   2} {3   4} {5} {6   7   8   9  10}{11  12}{13}{14  15}
    auto col1 = make_col<int32_t>({10, 36, 14, 32, 49, 23, 10, 34, 12, 45, 12, 37, 43, 26, 21, 16});
    auto col2 = make_col<int32_t>({10, 63, 41, 23, 94, 32, 10, 43, 21, 54, 22, 73, 34, 62, 12, 61});
    auto segs = make_col<int32_t>({0, 3, 5, 5, 5, 6, 11, 13, 14, 16});
    table_view in1{{col1->view()}}, in2{{col1->view(), col2->view()}};
    using V = std::vector<int32_t>;
    CHECK((to_host<int32_t>(segmented_sort_by_key(in1, in1, segs->view(), {order::ASCENDING})->view().column(0)) ==
           V{10, 14, 36, 32, 49, 23, 10, 12, 12, 34, 45, 37, 43, 26, 16, 21}));
    CHECK((to_host<int32_t>(segmented_sort_by_key(in1, in1, segs->view(), {order::DESCENDING})->view().column(0)) ==
           V{36, 14, 10, 49, 32, 23, 45, 34, 12, 12, 10, 43, 37, 26, 21, 16}));
    auto r2 = segmented_sort_by_key(in2, in2, segs->view(), {order::ASCENDING, order::DESCENDING});
    CHECK((to_host<int32_t>(r2->view().column(1)) == V{10, 41, 63, 23, 94, 32, 10, 22, 21, 43, 54, 73, 34, 62, 61, 12}));
    // offsets that do not cover every row: the rows outside keep their place (segmented_sort_impl.cuh:190-196)
    auto part = make_col<int32_t>({3, 7});
    CHECK((to_host<int32_t>(segmented_sorted_order(in1, part->view())->view()) ==
           V{0, 1, 2, 6, 5, 3, 4, 7, 8, 9, 10, 11, 12, 13, 14, 15}));
    // nulls inside segments, both precedences (segmented_sort_tests.cpp:106-130)
    auto n1 = make_col<int32_t>({1, 3, 2, 4, 5, 23, 6, 8, 7, 9, 7, 37, 43, 26, 21, 16}, {1, 1, 0, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1});
    table_view inn{{n1->view()}};
    auto aa = segmented_sort_by_key(inn, inn, segs->view(), {}, {null_order::AFTER});
    CHECK((valid_host(aa->view().column(0)) == std::vector<int>{1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1}));
    auto ha = to_host<int32_t>(aa->view().column(0));
    CHECK(ha[0] == 1 && ha[1] == 3 && ha[6] == 6 && ha[7] == 7 && ha[8] == 7 && ha[9] == 8 && ha[14] == 16 && ha[15] == 21);
    auto ab = segmented_sort_by_key(inn, inn, segs->view(), {}, {null_order::BEFORE});
    CHECK((valid_host(ab->view().column(0)) == std::vector<int>{0, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1}));
  });

  // ---- sort-based groupby: the reference's tests force this path with an extra NTH_ELEMENT(0) request
  // (groupby_test_util.cpp:60-64); with it the keys come out SORTED and are compared without re-sorting (:77-80)
  run("sort-path groupby: SUM/COUNT/MIN/MAX/MEAN/PRODUCT, keys sorted (sum_tests.cpp:68-80, count_tests.cpp:21-41, min/max_tests.cpp:37-57, force_use_sort_impl::YES)", [&] {
    auto keys = make_col<int32_t>({1, 2, 3, 1, 2, 2, 1, 3, 3, 2});
    auto vals = make_col<int32_t>({0, 1, 2, 3, 4, 5, 6, 7, 8, 9});
    groupby::groupby gb{table_view{{keys->view()}}};
    std::vector<groupby::aggregation_request> reqs(1);
    reqs[0].values = vals->view();
    reqs[0].aggregations.push_back(make_sum_aggregation<groupby_aggregation>());
    reqs[0].aggregations.push_back(make_count_aggregation<groupby_aggregation>());
    reqs[0].aggregations.push_back(make_min_aggregation<groupby_aggregation>());
    reqs[0].aggregations.push_back(make_max_aggregation<groupby_aggregation>());
    reqs[0].aggregations.push_back(make_mean_aggregation<groupby_aggregation>());
    reqs[0].aggregations.push_back(make_product_aggregation<groupby_aggregation>());
    reqs[0].aggregations.push_back(make_nth_element_aggregation<groupby_aggregation>(0));
    reqs[0].aggregations.push_back(make_nth_element_aggregation<groupby_aggregation>(-1));
    reqs[0].aggregations.push_back(make_argmax_aggregation<groupby_aggregation>());
    reqs[0].aggregations.push_back(make_variance_aggregation<groupby_aggregation>());
    auto [k, res] = gb.aggregate(reqs);
    CHECK((to_host<int32_t>(k->view().column(0)) == std::vector<int32_t>{1, 2, 3}));
    auto const& r = res[0].results;
    CHECK(r[0]->type().id() == type_id::INT64 && r[5]->type().id() == type_id::INT64);
    CHECK((to_host<int64_t>(r[0]->view()) == std::vector<int64_t>{9, 19, 17}));
    CHECK((to_host<int32_t>(r[1]->view()) == std::vector<int32_t>{3, 4, 3}));
    CHECK((to_host<int32_t>(r[2]->view()) == std::vector<int32_t>{0, 1, 2}));
    CHECK((to_host<int32_t>(r[3]->view()) == std::vector<int32_t>{6, 9, 8}));
    CHECK((to_host<double>(r[4]->view()) == std::vector<double>{3., 19. / 4, 17. / 3}));
    CHECK((to_host<int64_t>(r[5]->view()) == std::vector<int64_t>{0, 180, 112}));
    CHECK((to_host<int32_t>(r[6]->view()) == std::vector<int32_t>{0, 1, 2}));   // first value of each group
    CHECK((to_host<int32_t>(r[7]->view()) == std::vector<int32_t>{6, 9, 8}));   // last value of each group
    CHECK((to_host<int32_t>(r[8]->view()) == std::vector<int32_t>{6, 9, 8}));   // ARGMAX: row index of the maximum
    {  // var_tests.cpp:37-57 (the reference compares within 4 ulp: column_utilities.hpp:38)
      auto v      = to_host<double>(r[9]->view());
      double ev[] = {9., 131. / 12, 31. / 3};
      for (int g = 0; g < 3; ++g) CHECK(std::abs(v[g] - ev[g]) <= 4 * std::numeric_limits<double>::epsilon() * ev[g]);
    }
    // double SUM through the segmented reduce (double-double accumulation): exact on these values
    auto dv = make_col<double>({0.1, 1.5, 2.25, 3.0, 4.5, 5.125, 6.0, 7.5, 8.0, 9.75});
    std::vector<groupby::aggregation_request> r2(1);
    r2[0].values = dv->view();
    r2[0].aggregations.push_back(make_sum_aggregation<groupby_aggregation>());
    r2[0].aggregations.push_back(make_nth_element_aggregation<groupby_aggregation>(0));
    auto [k2, res2] = gb.aggregate(r2);
    CHECK((to_host<double>(res2[0].results[0]->view()) == std::vector<double>{0.1 + 3.0 + 6.0, 1.5 + 4.5 + 5.125 + 9.75, 2.25 + 7.5 + 8.0}));
  });
  run("sort-path groupby with null keys and values (sum_tests.cpp:124-145, count_tests.cpp:103-132, max_tests.cpp:100-120)", [&] {
    auto keys = make_col<int32_t>({1, 2, 3, 1, 2, 2, 1, 3, 3, 2, 4}, {1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1});
    auto vals = make_col<double>({0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 4}, {0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 0});
    groupby::groupby gb{table_view{{keys->view()}}};
    std::vector<groupby::aggregation_request> reqs(1);
    reqs[0].values = vals->view();
    reqs[0].aggregations.push_back(make_sum_aggregation<groupby_aggregation>());
    reqs[0].aggregations.push_back(make_count_aggregation<groupby_aggregation>());
    reqs[0].aggregations.push_back(make_count_aggregation<groupby_aggregation>(null_policy::INCLUDE));
    reqs[0].aggregations.push_back(make_max_aggregation<groupby_aggregation>());
    reqs[0].aggregations.push_back(make_nth_element_aggregation<groupby_aggregation>(0));
    auto [k, res] = gb.aggregate(reqs);
    CHECK((to_host<int32_t>(k->view().column(0)) == std::vector<int32_t>{1, 2, 3, 4}));
    auto const& r = res[0].results;
    auto s = to_host<double>(r[0]->view());
    CHECK((valid_host(r[0]->view()) == std::vector<int>{1, 1, 1, 0}));
    CHECK(s[0] == 9. && s[1] == 14. && s[2] == 10.);
    CHECK((to_host<int32_t>(r[1]->view()) == std::vector<int32_t>{2, 3, 2, 0}));
    CHECK((to_host<int32_t>(r[2]->view()) == std::vector<int32_t>{3, 4, 2, 1}));
    auto mx = to_host<double>(r[3]->view());
    CHECK((valid_host(r[3]->view()) == std::vector<int>{1, 1, 1, 0}));
    CHECK(mx[0] == 6. && mx[1] == 9. && mx[2] == 8.);
  });
  run("pre-sorted keys: sorted::YES, descending, nullable, nulls kept (keys_tests.cpp:109-203)", [&] {
    auto sum_of = [&](std::unique_ptr<column> const& keys, std::unique_ptr<column> const& vals, null_policy np,
                      std::vector<order> const& ord = {}) {
      groupby::groupby gb{table_view{{keys->view()}}, np, sorted::YES, ord, {null_order::BEFORE}};
      std::vector<groupby::aggregation_request> reqs(1);
      reqs[0].values = vals->view();
      reqs[0].aggregations.push_back(make_sum_aggregation<groupby_aggregation>());
      return gb.aggregate(reqs);
    };
    auto vals = make_col<int32_t>({0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 4});
    {
      auto keys     = make_col<int32_t>({1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4});
      auto [k, res] = sum_of(keys, vals, null_policy::EXCLUDE);
      CHECK((to_host<int32_t>(k->view().column(0)) == std::vector<int32_t>{1, 2, 3, 4}));
      CHECK((to_host<int64_t>(res[0].results[0]->view()) == std::vector<int64_t>{3, 18, 24, 4}));
    }
    {
      auto keys     = make_col<int32_t>({4, 3, 3, 3, 2, 2, 2, 2, 1, 1, 1});
      auto [k, res] = sum_of(keys, vals, null_policy::EXCLUDE, {order::DESCENDING});
      CHECK((to_host<int32_t>(k->view().column(0)) == std::vector<int32_t>{4, 3, 2, 1}));
      CHECK((to_host<int64_t>(res[0].results[0]->view()) == std::vector<int64_t>{0, 6, 22, 21}));
    }
    {  // nulls to exclude: the helper re-sorts (sort_helper.cu:47-54)
      auto keys     = make_col<int32_t>({1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4}, {1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 1});
      auto [k, res] = sum_of(keys, vals, null_policy::EXCLUDE);
      CHECK((to_host<int32_t>(k->view().column(0)) == std::vector<int32_t>{1, 2, 3, 4}));
      CHECK(k->view().column(0).null_count() == 0);
      CHECK((to_host<int64_t>(res[0].results[0]->view()) == std::vector<int64_t>{3, 15, 17, 4}));
    }
    {  // nulls kept: every run of adjacent equal rows (null == null) is a group
      auto keys     = make_col<int32_t>({1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4}, {1, 1, 1, 0, 0, 1, 1, 0, 1, 1, 1});
      auto [k, res] = sum_of(keys, vals, null_policy::INCLUDE);
      CHECK((valid_host(k->view().column(0)) == std::vector<int>{1, 0, 1, 0, 1, 1}));
      auto kh = to_host<int32_t>(k->view().column(0));
      CHECK(kh[0] == 1 && kh[2] == 2 && kh[4] == 3 && kh[5] == 4);
      CHECK((to_host<int64_t>(res[0].results[0]->view()) == std::vector<int64_t>{3, 7, 11, 7, 17, 4}));
    }
    {  // unsorted keys with nulls KEPT (null_policy::INCLUDE, keys_tests.cpp:86-107): the null key is a group
      auto keys = make_col<int32_t>({1, 2, 3, 1, 2, 2, 1, 3, 3, 2, 4}, {1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1});
      groupby::groupby gb{table_view{{keys->view()}}, null_policy::INCLUDE};
      std::vector<groupby::aggregation_request> reqs(1);
      reqs[0].values = vals->view();
      reqs[0].aggregations.push_back(make_sum_aggregation<groupby_aggregation>());
      auto [k, res] = gb.aggregate(reqs);
      CHECK((valid_host(k->view().column(0)) == std::vector<int>{1, 1, 1, 1, 0}));   // nulls AFTER (sort_helper.cu:92-94)
      auto kh = to_host<int32_t>(k->view().column(0));
      CHECK(kh[0] == 1 && kh[1] == 2 && kh[2] == 3 && kh[3] == 4);
      CHECK((to_host<int64_t>(res[0].results[0]->view()) == std::vector<int64_t>{9, 19, 10, 4, 7}));
    }
    {  // groupby::scan on pre-sorted keys skips the sort (two key columns)
      auto k0 = make_col<int32_t>({1, 1, 1, 2, 2, 3});
      auto k1 = make_col<int64_t>({5, 5, 6, 6, 6, 6});
      auto v  = make_col<int32_t>({1, 2, 3, 4, 5, 6});
      groupby::groupby gb{table_view{{k0->view(), k1->view()}}, null_policy::EXCLUDE, sorted::YES};
      std::vector<groupby::scan_request> reqs(1);
      reqs[0].values = v->view();
      reqs[0].aggregations.push_back(make_sum_aggregation<groupby_scan_aggregation>());
      auto [k, res] = gb.scan(reqs);
      CHECK((to_host<int64_t>(res[0].results[0]->view()) == std::vector<int64_t>{1, 3, 3, 4, 9, 6}));
      CHECK((to_host<int64_t>(k->view().column(1)) == std::vector<int64_t>{5, 5, 6, 6, 6, 6}));
    }
  });
  run("groupby::get_groups / shift / replace_nulls (group_shift / shift_tests.cpp:34-140, replace_nulls_tests.cpp:40-105, groupby.cu:261-362)", [&] {
    auto key = make_col<int32_t>({1, 2, 1, 2, 2, 1, 1, 2, 1, 2, 1, 2, 1});
    auto val = make_col<int32_t>({3, 4, 5, 6, 7, 8, 9, 0, 1, 2, 3, 4, 5});
    groupby::groupby gb{table_view{{key->view()}}};
    auto grp = gb.get_groups(table_view{{val->view()}});
    CHECK((grp.offsets == std::vector<size_type>{0, 7, 13}));
    CHECK((to_host<int32_t>(grp.keys->view().column(0)) == std::vector<int32_t>{1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2}));
    CHECK((to_host<int32_t>(grp.values->view().column(0)) == std::vector<int32_t>{3, 5, 8, 9, 1, 3, 5, 4, 6, 7, 0, 2, 4}));
    // forward shift by 3, valid fill 42 (ForwardShiftWithoutNull_ValidScalar)
    numeric_scalar<int32_t> fill{42};
    std::vector<size_type> off{3};
    auto [sk, sv] = gb.shift(table_view{{val->view()}}, off, {fill});
    CHECK((to_host<int32_t>(sv->view().column(0)) == std::vector<int32_t>{42, 42, 42, 3, 5, 8, 9, 42, 42, 42, 4, 6, 7}));
    CHECK(sv->view().column(0).null_count() == 0);
    // backward shift by 1, null fill (BackwardShiftWithoutNull_NullScalar)
    auto key2 = make_col<int32_t>({1, 2, 1, 2, 2, 1, 1});
    auto val2 = make_col<int32_t>({3, 4, 5, 6, 7, 8, 9});
    groupby::groupby gb2{table_view{{key2->view()}}};
    numeric_scalar<int32_t> nullfill{0, false};
    std::vector<size_type> off2{-1};
    auto [sk2, sv2] = gb2.shift(table_view{{val2->view()}}, off2, {nullfill});
    auto h2 = to_host<int32_t>(sv2->view().column(0));
    CHECK((valid_host(sv2->view().column(0)) == std::vector<int>{1, 1, 1, 0, 1, 1, 0}));
    CHECK(h2[0] == 5 && h2[1] == 8 && h2[2] == 9 && h2[4] == 6 && h2[5] == 7);
    // forward shift with nulls in the values (ForwardShiftWithNull_ValidScalar)
    auto val3 = make_col<int32_t>({3, 4, 5, 6, 7, 8, 9, 0, 1, 2, 3, 4, 5}, {1, 0, 1, 0, 1, 0, 0, 1, 0, 1, 1, 0, 1});
    auto [sk3, sv3] = gb.shift(table_view{{val3->view()}}, off, {fill});
    CHECK((valid_host(sv3->view().column(0)) == std::vector<int>{1, 1, 1, 1, 1, 0, 0, 1, 1, 1, 0, 0, 1}));
    auto h3 = to_host<int32_t>(sv3->view().column(0));
    CHECK(h3[0] == 42 && h3[3] == 3 && h3[4] == 5 && h3[7] == 42 && h3[12] == 7);
    // replace_nulls: PrecedingFill / FollowingFill / leading and trailing nulls
    auto rk = make_col<int32_t>({0, 1, 0, 1, 0, 1});
    auto rv = make_col<int32_t>({42, 7, 24, 10, 1, 1000}, {1, 1, 1, 0, 0, 0});
    groupby::groupby gbr{table_view{{rk->view()}}};
    std::vector<replace_policy> pol{replace_policy::PRECEDING};
    auto [pk, pv] = gbr.replace_nulls(table_view{{rv->view()}}, pol);
    CHECK((to_host<int32_t>(pk->view().column(0)) == std::vector<int32_t>{0, 0, 0, 1, 1, 1}));
    CHECK((to_host<int32_t>(pv->view().column(0)) == std::vector<int32_t>{42, 24, 24, 7, 7, 7}));
    CHECK(pv->view().column(0).null_count() == 0);
    auto fk = make_col<int32_t>({0, 0, 1, 1, 0, 1, 1, 1});
    auto fv = make_col<int32_t>({2, 4, 8, 16, 32, 64, 128, 256}, {1, 0, 1, 0, 1, 0, 1, 1});
    groupby::groupby gbf{table_view{{fk->view()}}};
    std::vector<replace_policy> polf{replace_policy::FOLLOWING};
    auto [qk, qv] = gbf.replace_nulls(table_view{{fv->view()}}, polf);
    CHECK((to_host<int32_t>(qv->view().column(0)) == std::vector<int32_t>{2, 32, 32, 8, 128, 128, 128, 256}));
    auto tv = make_col<int32_t>({2, 4, 8, 16, 32, 64, 128, 256}, {1, 0, 0, 0, 0, 1, 0, 0});
    auto [tk, tvv] = gbf.replace_nulls(table_view{{tv->view()}}, polf);
    CHECK((valid_host(tvv->view().column(0)) == std::vector<int>{1, 0, 0, 1, 1, 1, 0, 0}));
    auto th = to_host<int32_t>(tvv->view().column(0));
    CHECK(th[0] == 2 && th[3] == 64 && th[4] == 64 && th[5] == 64);
    auto lv = make_col<int32_t>({42, 7, 24, 10, 1, 1000}, {0, 0, 1, 0, 0, 0});
    auto [lk, lvv] = gbr.replace_nulls(table_view{{lv->view()}}, pol);
    CHECK((valid_host(lvv->view().column(0)) == std::vector<int>{0, 1, 1, 0, 0, 0}));
  });

  // ---- reduce / scan: cpp/tests/reductions/{reduction_tests.cpp,scan_tests.cpp}
  run("reduce SUM/MIN/MAX with nulls; all-null -> invalid scalar", [] {
    auto c = make_col<int32_t>({6, -14, 13, 64, 0, -13, -20, 45}, {1, 0, 1, 1, 1, 1, 0, 1});
    auto s = reduce(c->view(), *make_sum_aggregation<reduce_aggregation>(), data_type{type_id::INT64});
    CHECK(s->is_valid() && static_cast<numeric_scalar<int64_t>&>(*s).value() == 6 + 13 + 64 + 0 - 13 + 45);
    auto mn = reduce(c->view(), *make_min_aggregation<reduce_aggregation>(), data_type{type_id::INT32});
    auto mx = reduce(c->view(), *make_max_aggregation<reduce_aggregation>(), data_type{type_id::INT32});
    CHECK(static_cast<numeric_scalar<int32_t>&>(*mn).value() == -13 && static_cast<numeric_scalar<int32_t>&>(*mx).value() == 64);
    auto n = make_col<double>({1., 2.}, {0, 0});
    auto sn = reduce(n->view(), *make_sum_aggregation<reduce_aggregation>(), data_type{type_id::FLOAT64});
    CHECK(!sn->is_valid());
    CHECK(throws<cudf::data_type_error>([&] { (void)reduce(c->view(), *make_min_aggregation<reduce_aggregation>(), data_type{type_id::INT64}); }));
  });
  run("scan inclusive / exclusive, null policies (scan_inclusive.cu:36-61,198-216)", [] {
    auto c = make_col<int32_t>({1, 2, 3, 4, 5}, {1, 1, 0, 1, 1});
    auto inc = scan(c->view(), *make_sum_aggregation<scan_aggregation>(), scan_type::INCLUSIVE);
    auto h = to_host<int32_t>(inc->view());
    auto v = valid_host(inc->view());
    CHECK(h[0] == 1 && h[1] == 3 && h[3] == 7 && h[4] == 12 && (v == std::vector<int>{1, 1, 0, 1, 1}));
    auto exc = scan(c->view(), *make_sum_aggregation<scan_aggregation>(), scan_type::EXCLUSIVE);
    auto he = to_host<int32_t>(exc->view());
    CHECK(he[0] == 0 && he[1] == 1 && he[3] == 3 && he[4] == 7);
    auto poison = scan(c->view(), *make_max_aggregation<scan_aggregation>(), scan_type::INCLUSIVE, null_policy::INCLUDE);
    CHECK((valid_host(poison->view()) == std::vector<int>{1, 1, 0, 0, 0}));
    auto w = make_col<int8_t>({100, 100, 100});
    auto ws = scan(w->view(), *make_sum_aggregation<scan_aggregation>(), scan_type::INCLUSIVE);
    CHECK(to_host<int8_t>(ws->view())[2] == (int8_t)(300 & 0xFF));  // output dtype == input dtype, wraps
  });

  // ---- hashing / partitioning / gather / data model
  run("murmurhash3_x86_32 + hash_partition (murmurhash3_x86_32_test.cpp; partitioning.hpp:103-110)", [] {
    auto c = make_col<int32_t>({0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11});
    auto h = hashing::murmurhash3_x86_32(table_view{{c->view()}});
    auto hh = to_host<uint32_t>(h->view());
    auto [t, offs] = hash_partition(table_view{{c->view()}}, {0}, 4);
    auto out = to_host<int32_t>(t->view().column(0));
    // num_partitions + 1 offsets, the last one = rows (partitioning.cu:684-688; hash_partition_test.cpp:422-431)
    CHECK(offs.size() == 5 && offs[0] == 0 && offs[4] == 12 && t->num_rows() == 12);
    for (int p = 0; p < 4; ++p)
      for (int i = offs[p]; i < offs[p + 1]; ++i) CHECK((int)(hh[out[i]] % 4) == p);
    auto sorted = out;
    std::sort(sorted.begin(), sorted.end());
    CHECK((sorted == to_host<int32_t>(c->view())));
    // hash_partition_test.cpp:73-141: zero partitions / zero rows / nothing to hash -> EMPTY table, num_partitions + 1 zeros
    auto [t0, o0] = hash_partition(table_view{{c->view()}}, {0}, 0);
    CHECK(t0->num_rows() == 0 && t0->num_columns() == 1 && o0.size() == 1);
    auto e = make_col<int32_t>({});
    auto [t1, o1] = hash_partition(table_view{{e->view()}}, {0}, 3);
    CHECK(t1->num_rows() == 0 && o1.size() == 4 && o1[3] == 0);
    auto [t2, o2] = hash_partition(table_view{{c->view()}}, std::vector<size_type>{}, 3);
    CHECK(t2->num_rows() == 0 && t2->num_columns() == 1 && t2->view().column(0).type() == c->type() && o2.size() == 4);
    CHECK(throws<std::out_of_range>([&] { (void)hash_partition(table_view{{c->view()}}, {-1}, 3); }));                       // :49-60
    auto k9 = make_col<int16_t>({1, 2, 3, 4, 5, 6, 7, 8, 9});
    CHECK(throws<std::invalid_argument>([&] { (void)hash_partition(table_view{{c->view()}}, table_view{{k9->view()}}, 3); }));  // :62-71
    // the keys-table overload and HASH_IDENTITY over the externally computed row hashes agree with the direct form (:411-423)
    auto [t3, o3] = hash_partition(table_view{{c->view()}}, table_view{{h->view()}}, 4, hash_id::HASH_IDENTITY);
    CHECK((o3 == offs) && (to_host<int32_t>(t3->view().column(0)) == out));
    // cudf::partition by an explicit map (partitioning.hpp:44-78, the documented example)
    auto vals = make_col<int32_t>({10, 20, 30, 40, 50});
    auto pmap = make_col<int32_t>({1, 0, 1, 2, 0});
    auto [pt, po] = partition(table_view{{vals->view()}}, pmap->view(), 4);
    CHECK((po == std::vector<size_type>{0, 2, 4, 5, 5}) && (to_host<int32_t>(pt->view().column(0)) == std::vector<int32_t>{20, 50, 10, 30, 40}));
  });
  run("reduce with an initial value (reduction.hpp:124-130; reduction_tests.cpp:122-235,330-421)", [] {
    auto c = make_col<int32_t>({6, -14, 13, 64, 0, -13, -20, 45});
    numeric_scalar<int32_t> init{100};
    auto s = reduce(c->view(), *make_sum_aggregation<reduce_aggregation>(), data_type{type_id::INT32}, std::cref<scalar>(init));
    CHECK(s->is_valid() && static_cast<numeric_scalar<int32_t>&>(*s).value() == 181);
    auto mm = make_col<int64_t>({5, 0, -120, -111, 0, 64, 63, 99, 123, -16}, {1, 1, 0, 1, 1, 1, 0, 1, 0, 1});
    numeric_scalar<int64_t> i64{100};
    auto mn = reduce(mm->view(), *make_min_aggregation<reduce_aggregation>(), data_type{type_id::INT64}, std::cref<scalar>(i64));
    auto mx = reduce(mm->view(), *make_max_aggregation<reduce_aggregation>(), data_type{type_id::INT64}, std::cref<scalar>(i64));
    CHECK(static_cast<numeric_scalar<int64_t>&>(*mn).value() == -111 && static_cast<numeric_scalar<int64_t>&>(*mx).value() == 100);
    init.set_valid_async(false);
    auto inv = reduce(c->view(), *make_sum_aggregation<reduce_aggregation>(), data_type{type_id::INT32}, std::cref<scalar>(init));
    CHECK(!inv->is_valid());                                                                       // invalid init -> invalid result (simple.cuh:80-83)
    CHECK(throws<cudf::data_type_error>([&] { (void)reduce(c->view(), *make_sum_aggregation<reduce_aggregation>(), data_type{type_id::INT64}, std::cref<scalar>(i64)); }));
    numeric_scalar<int32_t> one{1};
    CHECK(throws<std::invalid_argument>([&] { (void)reduce(c->view(), *make_mean_aggregation<reduce_aggregation>(), data_type{type_id::FLOAT64}, std::cref<scalar>(one)); }));
  });
  run("Arrow C Device Data Interface round trip (interop.hpp:477-606,838-885)", [] {
    std::vector<std::unique_ptr<column>> cols;
    cols.emplace_back(make_col<int64_t>({5, -3, 9, 0, 7}));
    cols.emplace_back(make_col<double>({0.5, 1.5, 2.5, 3.5, 4.5}, {1, 0, 1, 1, 0}));
    cols.emplace_back(make_col<uint16_t>({1, 2, 3, 4, 5}));
    table t{std::move(cols)};
    std::vector<column_metadata> meta{{"a"}, {"b"}, {"c"}};
    auto schema = to_arrow_schema(t.view(), meta);
    CHECK(std::string{schema->format} == "+s" && schema->n_children == 3);
    CHECK(std::string{schema->children[0]->format} == "l" && std::string{schema->children[0]->name} == "a");
    CHECK(std::string{schema->children[1]->format} == "g" && (schema->children[1]->flags & ARROW_FLAG_NULLABLE));
    CHECK(std::string{schema->children[2]->format} == "S" && !(schema->children[2]->flags & ARROW_FLAG_NULLABLE));
    auto dev = to_arrow_device(std::move(t));
    CHECK(dev->device_type == ARROW_DEVICE_ROCM && dev->sync_event != nullptr);
    CHECK(dev->array.length == 5 && dev->array.n_children == 3 && dev->array.n_buffers == 1);
    CHECK(dev->array.children[1]->null_count == 2 && dev->array.children[1]->n_buffers == 2);
    CHECK(dev->array.children[0]->buffers[0] == nullptr && dev->array.children[0]->buffers[1] != nullptr);
    // import: zero-copy views over the exported buffers
    auto tv = from_arrow_device(schema.get(), dev.get());
    CHECK(tv->num_columns() == 3 && tv->num_rows() == 5);
    CHECK(tv->column(0).head<void>() == dev->array.children[0]->buffers[1]);
    CHECK((to_host<int64_t>(tv->column(0)) == std::vector<int64_t>{5, -3, 9, 0, 7}));
    CHECK((valid_host(tv->column(1)) == std::vector<int>{1, 0, 1, 1, 0}) && tv->column(1).null_count() == 2);
    CHECK((to_host<uint16_t>(tv->column(2)) == std::vector<uint16_t>{1, 2, 3, 4, 5}));
    // the imported view feeds the hot path directly
    auto order = sorted_order(table_view{{tv->column(0)}});
    CHECK((to_host<int32_t>(order->view()) == std::vector<int32_t>{1, 3, 0, 4, 2}));
    // a single column, null_count left to the consumer (-1), and a sliced (offset) non-owning export
    auto c  = make_col<int32_t>({10, 20, 30, 40, 50, 60}, {1, 1, 0, 1, 0, 1});
    column_view sliced{c->type(), 4, c->view().head<void>(), c->view().null_mask(), 2, 1};
    auto dcol = to_arrow_device(sliced);
    CHECK(dcol->array.offset == 1 && dcol->array.length == 4);
    dcol->array.null_count = -1;
    ArrowSchema cs{};
    cs.format = "i";
    auto cv = from_arrow_device_column(&cs, dcol.get());
    CHECK(cv->size() == 4 && cv->offset() == 1 && cv->null_count() == 2);
    CHECK((to_host<int32_t>(*cv) == std::vector<int32_t>{20, 30, 40, 50}));
    CHECK((valid_host(*cv) == std::vector<int>{1, 0, 1, 0}));
    // errors
    CHECK(throws<std::invalid_argument>([&] { (void)from_arrow_device(nullptr, dev.get()); }));
    CHECK(throws<cudf::data_type_error>([&] { (void)from_arrow_device(&cs, dcol.get()); }));  // not a struct
    ArrowDeviceArray host_side = *dcol;
    host_side.device_type      = ARROW_DEVICE_CPU;
    host_side.sync_event       = nullptr;
    CHECK(throws<std::invalid_argument>([&] { (void)from_arrow_device_column(&cs, &host_side); }));
    ArrowSchema bad{};
    bad.format = "u";  // utf8 string
    CHECK(throws<cudf::data_type_error>([&] { (void)from_arrow_device_column(&bad, dcol.get()); }));
    auto b8 = make_col<uint8_t>({1, 0});
    column_view b{data_type{type_id::BOOL8}, 2, b8->view().head<void>(), nullptr, 0};  // Arrow packs booleans into bits
    CHECK(throws<cudf::data_type_error>([&] { (void)to_arrow_device(b); }));
    CHECK(throws<std::invalid_argument>([&] { (void)to_arrow_schema(table_view{{c->view()}}, std::vector<column_metadata>{}); }));
  });
  run("gather NULLIFY / column & table ownership (gather.cuh:506-577, column.hpp:248-269)", [] {
    auto c = make_col<double>({1., 2., 3.}, {1, 0, 1});
    auto m = make_col<int32_t>({2, JoinNoMatch, 1, 0});
    auto g = gather(table_view{{c->view()}}, m->view(), out_of_bounds_policy::NULLIFY);
    CHECK((valid_host(g->view().column(0)) == std::vector<int>{1, 0, 0, 1}));
    CHECK(g->get_column(0).null_count() == 2);
    column copy{*c};
    CHECK(copy.size() == 3 && copy.null_count() == 1);
    auto contents = copy.release();
    CHECK(copy.size() == 0 && copy.type().id() == type_id::EMPTY && contents.data->size() == 24);
    CHECK(throws<cudf::logic_error>([&] { table_view bad{{c->view(), m->view()}}; }));
    column_view sl{c->type(), 2, c->view().head<void>(), c->view().null_mask(), 1, 1};
    CHECK(sl.null_count(0, 2) == 1 && sl.data<double>() == c->view().data<double>() + 1);
  });

  run("pool_memory_resource: stream-ordered reuse, cross-stream reuse behind an event, H2D right after a recycle", [] {
    auto* pool = dynamic_cast<rmm::mr::pool_memory_resource*>(rmm::mr::get_default_resource());
    if (pool == nullptr) return;  // CUDF_AMD_ALLOC selected another resource
    // (1) the same request again and again: one driver allocation per distinct size, none afterwards
    std::vector<int64_t> kv(200000);
    for (std::size_t i = 0; i < kv.size(); ++i) kv[i] = static_cast<int64_t>((i * 2654435761u) % 1000003) - 500000;
    auto keys = make_col<int64_t>(kv);
    auto want = kv;
    std::sort(want.begin(), want.end());
    std::size_t after_first = 0;
    for (int it = 0; it < 4; ++it) {
      auto out = cudf::sort(table_view{{keys->view()}});
      CHECK(to_host<int64_t>(out->view().column(0)) == want);
      if (it == 0) after_first = pool->driver_allocations();
    }
    CHECK(pool->driver_allocations() == after_first);
    // (2) a pageable host-to-device copy into a block that was just recycled (what hipMallocAsync lost on ROCm 7.2)
    for (int it = 0; it < 6; ++it) {
      std::vector<int32_t> a(300000);
      for (std::size_t i = 0; i < a.size(); ++i) a[i] = static_cast<int32_t>(i % 5000) + it;
      auto c = make_col<int32_t>(a);
      CHECK(to_host<int32_t>(c->view()) == a);
      auto s = cudf::sort(table_view{{c->view()}});  // scratch and output come from, and go back to, the pool
      auto const hs = to_host<int32_t>(s->view().column(0));
      CHECK(hs.size() == a.size() && std::is_sorted(hs.begin(), hs.end()));
    }
    // (3) freed on one stream, taken on another: the taker is ordered behind the last use on the first stream
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking) == hipSuccess);
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) == hipSuccess);
    std::size_t const n = std::size_t{48} << 20;  // 192 MiB of int32: the fill on s1 is still running when s2 asks
    for (int it = 0; it < 3; ++it) {
      void* p1 = nullptr;
      {
        rmm::device_buffer b1{n * 4, rmm::cuda_stream_view{s1}};
        p1 = b1.data();
        for (int r = 0; r < 4; ++r) CHECK(gx_sequence_i32(static_cast<int32_t*>(b1.data()), (int64_t)n, 1000 + r, s1) == 0);
      }  // freed on s1 with the fills still queued
      rmm::device_buffer b2{n * 4, rmm::cuda_stream_view{s2}};
      CHECK(b2.data() == p1);  // the only cached block of that size
      CHECK(gx_sequence_i32(static_cast<int32_t*>(b2.data()), (int64_t)n, 7, s2) == 0);
      std::vector<int32_t> h(n);
      CHECK(hipMemcpyAsync(h.data(), b2.data(), n * 4, hipMemcpyDeviceToHost, s2) == hipSuccess);
      CHECK(hipStreamSynchronize(s2) == hipSuccess);
      CHECK(hipStreamSynchronize(s1) == hipSuccess);
      bool ok = true;
      for (std::size_t i = 0; i < n; i += 4097) ok = ok && h[i] == static_cast<int32_t>(7 + i);
      CHECK(ok && h[n - 1] == static_cast<int32_t>(7 + n - 1));
    }
    (void)hipStreamDestroy(s1);
    (void)hipStreamDestroy(s2);
    // (4) release() hands everything back; the next request allocates again
    pool->release();
    CHECK(pool->cached_bytes() == 0);
    auto const before = pool->driver_allocations();
    { rmm::device_buffer b{1 << 20, get_default_stream()}; }
    CHECK(pool->driver_allocations() == before + 1 && pool->cached_bytes() >= (1u << 20));
  });

  // ---- the sharded operators (include/cudf_amd/distributed.hpp -> gxd.h -> distributed.cpp, RCCL): a 1-rank communicator with
  // the exchange path FORCED, so partition passes, count exchange, grouped send / recv (to self: a device copy), chunked probe,
  // segment tables and the merge all run; the multi-rank logic itself is covered with gloo on the CPU (tests/test_distributed_cpu.py)
  run("cudf_amd::distributed::{sort, hash_join, groupby_sum_count} on a 1-rank RCCL communicator, exchange forced", [] {
    namespace D = cudf_amd::distributed;
    D::communicator comm{D::make_unique_id(), 1, 0};
    CHECK(comm.world() == 1 && comm.rank() == 0);
    // sort: 3e6 pseudo-random keys (several chunks), result = std::sort
    std::vector<int64_t> k(3'000'017);
    uint64_t x = 88172645463325252ull;
    for (auto& v : k) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      v = static_cast<int64_t>(x);
    }
    auto kc = make_col<int64_t>(k);
    auto sorted = D::sort(kc->view(), comm, true);
    auto want = k;
    std::sort(want.begin(), want.end());
    CHECK((to_host<int64_t>(sorted->view()) == want));
    // join: build 2e5 keys (every key twice), probe 2.5e6 keys; pairs = global row ids, checked against a host hash map
    std::vector<int64_t> b(200'000), p(2'500'003);
    for (std::size_t i = 0; i < b.size(); ++i) b[i] = static_cast<int64_t>((i / 2) * 7 + 1);
    for (std::size_t i = 0; i < p.size(); ++i) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      p[i] = static_cast<int64_t>(x % 1'400'000);
    }
    auto bc = make_col<int64_t>(b);
    auto pc = make_col<int64_t>(p);
    D::hash_join hj{bc->view(), comm, true};
    for (int rep = 0; rep < 2; ++rep) {  // build once, probe twice
      auto [l, r] = hj.inner_join(pc->view());
      auto hl = to_host<int64_t>(l->view()), hr = to_host<int64_t>(r->view());
      std::size_t expect = 0;
      for (auto v : p) expect += (v % 7 == 1 && (v - 1) / 7 < 100'000) ? 2 : 0;
      CHECK(hl.size() == expect);
      bool ok = true;
      std::vector<uint8_t> seen(b.size() * 0 + p.size(), 0);
      for (std::size_t i = 0; i < hl.size() && ok; ++i) {
        ok = hl[i] >= 0 && hl[i] < (int64_t)p.size() && hr[i] >= 0 && hr[i] < (int64_t)b.size() && p[hl[i]] == b[hr[i]];
        if (ok) seen[hl[i]] += 1 + (hr[i] & 1);  // both build copies of the key: 1 + 2
      }
      CHECK(ok);
      for (std::size_t i = 0; i < p.size() && ok; ++i) ok = seen[i] == ((p[i] % 7 == 1 && (p[i] - 1) / 7 < 100'000) ? 3 : 0);
      CHECK(ok);
    }
    // groupby: 1e6 rows, 5000 groups, integer-valued doubles (sums exact in any order)
    std::vector<int32_t> gk(1'000'000);
    std::vector<double> gv(gk.size());
    std::vector<double> ws(5000, 0.0);
    std::vector<int64_t> wc(5000, 0);
    for (std::size_t i = 0; i < gk.size(); ++i) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      gk[i] = static_cast<int32_t>(x % 5000);
      gv[i] = static_cast<double>((x >> 20) % 100);
      ws[gk[i]] += gv[i];
      wc[gk[i]] += 1;
    }
    auto gkc = make_col<int32_t>(gk);
    auto gvc = make_col<double>(gv);
    auto t = D::groupby_sum_count(gkc->view(), gvc->view(), comm, true);
    CHECK(t->num_rows() == 5000);
    auto ok_ = to_host<int32_t>(t->view().column(0));
    auto os_ = to_host<double>(t->view().column(1));
    auto oc_ = to_host<int64_t>(t->view().column(2));
    bool good = true;
    for (int g = 0; g < 5000 && good; ++g) good = ok_[g] == g && os_[g] == ws[g] && oc_[g] == wc[g];  // keys ascending
    CHECK(good);
  });

  run("cudf_amd::distributed::sort / hash_join over 3 LOGICAL ranks on one device (loopback transport, one thread per rank)", [] {
    namespace D = cudf_amd::distributed;
    constexpr int W = 3;
    auto comms = D::communicator::loopback(W);
    CHECK((int)comms.size() == W && comms[2]->rank() == 2 && comms[0]->world() == W);
    // shards of different sizes, one of them empty
    std::vector<std::vector<int64_t>> shard(W), bshard(W), pshard(W);
    std::vector<int64_t> all, ball, pall;
    uint64_t x = 0x9E3779B97F4A7C15ull;
    auto next = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    const std::size_t sizes[W] = {1'500'003, 0, 700'001};
    for (int r = 0; r < W; ++r)
      for (std::size_t i = 0; i < sizes[r]; ++i) shard[r].push_back(static_cast<int64_t>(next()));
    for (int r = 0; r < W; ++r) all.insert(all.end(), shard[r].begin(), shard[r].end());
    for (std::size_t i = 0; i < 300'000; ++i) bshard[i % 2 ? 0 : 2].push_back(static_cast<int64_t>(i * 5 + 2));  // distinct build keys on ranks 0 and 2
    for (int r = 0; r < W; ++r)
      for (std::size_t i = 0; i < 400'000 + 100'000 * r; ++i) pshard[r].push_back(static_cast<int64_t>(next() % 2'000'000));
    std::vector<std::vector<int64_t>> sorted(W), jl(W), jr(W);
    std::vector<std::string> errs(W);
    std::vector<std::thread> th;
    for (int r = 0; r < W; ++r)
      th.emplace_back([&, r] {
        try {
          hipStream_t s = nullptr;
          if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) throw std::runtime_error("stream");
          rmm::cuda_stream_view sv{s};
          {  // (every column / buffer made on `s` goes back to the pool before the stream is destroyed)
            auto kc  = make_col<int64_t>(shard[r]);
            auto out = D::sort(kc->view(), *comms[r], true, sv);
            sv.synchronize();
            sorted[r] = to_host<int64_t>(out->view());
            auto bc = make_col<int64_t>(bshard[r]);
            auto pc = make_col<int64_t>(pshard[r]);
            D::hash_join hj{bc->view(), *comms[r], true, sv};
            auto [l, rr] = hj.inner_join(pc->view(), sv);
            sv.synchronize();
            jl[r] = to_host<int64_t>(l->view());
            jr[r] = to_host<int64_t>(rr->view());
          }
          sv.synchronize();
          (void)hipStreamDestroy(s);
        } catch (std::exception const& e) {
          errs[r] = e.what();
        }
      });
    for (auto& t : th) t.join();
    for (int r = 0; r < W; ++r) CHECK(errs[r].empty());
    std::vector<int64_t> got;
    for (int r = 0; r < W; ++r) got.insert(got.end(), sorted[r].begin(), sorted[r].end());
    std::sort(all.begin(), all.end());
    CHECK((got == all));  // rank order = key order
    // pairs: (global probe row, global build row) -- global row = first row of the owning rank's shard + local row
    for (int r = 0; r < W; ++r) {
      ball.insert(ball.end(), bshard[r].begin(), bshard[r].end());
      pall.insert(pall.end(), pshard[r].begin(), pshard[r].end());
    }
    std::size_t expect = 0, have = 0;
    for (auto v : pall) expect += (v % 5 == 2 && v / 5 < 300'000) ? 1 : 0;
    bool ok = true;
    std::vector<uint8_t> hit(pall.size(), 0);
    for (int r = 0; r < W; ++r) {
      have += jl[r].size();
      for (std::size_t i = 0; i < jl[r].size() && ok; ++i) {
        ok = jl[r][i] >= 0 && jl[r][i] < (int64_t)pall.size() && jr[r][i] >= 0 && jr[r][i] < (int64_t)ball.size() && pall[jl[r][i]] == ball[jr[r][i]] &&
             !hit[jl[r][i]];
        if (ok) hit[jl[r][i]] = 1;
      }
    }
    CHECK(ok && have == expect);
  });

  run("ONE cudf::hash_join probed from 4 host threads on 4 streams at once (hash_join.hpp:63-68: probes are const and thread-safe)", [] {
    // build once on the default stream; every thread then probes the same object concurrently on a stream of its own: a large
    // probe (>= 2^22 rows: the partitioned probe, scratch from the shared pooled mr), inner_join_size, a left join and a small
    // probe (the direct kernel), each checked against the closed form of its own probe column
    constexpr std::size_t NB = 3'000'000;
    std::vector<int64_t> bk(NB);
    for (std::size_t i = 0; i < NB; ++i) bk[i] = static_cast<int64_t>((i * 2654435761ull) % NB) * 5 + 2;  // distinct (a bijection mod NB... 2654435761 is odd, NB is not a power of two: checked below)
    {
      std::vector<uint8_t> seen(NB, 0);
      bool distinct = true;
      for (auto v : bk) {
        auto j = static_cast<std::size_t>((v - 2) / 5);
        if (seen[j]) distinct = false;
        seen[j] = 1;
      }
      if (!distinct)
        for (std::size_t i = 0; i < NB; ++i) bk[i] = static_cast<int64_t>(NB - 1 - i) * 5 + 2;
    }
    std::vector<int32_t> row_of(NB);
    for (std::size_t i = 0; i < NB; ++i) row_of[static_cast<std::size_t>((bk[i] - 2) / 5)] = static_cast<int32_t>(i);
    auto bc = make_col<int64_t>(bk);
    get_default_stream().synchronize();
    cudf::hash_join hj{table_view{{bc->view()}}, null_equality::EQUAL};
    get_default_stream().synchronize();
    constexpr int T = 4;
    std::vector<std::string> errs(T);
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        try {
          hipStream_t s = nullptr;
          if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) throw std::runtime_error("stream");
          rmm::cuda_stream_view sv{s};
          {
            uint64_t x = 0x9E3779B97F4A7C15ull * (t + 1);
            auto next  = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
            const std::size_t np = 4'400'000 + 150'001 * t;
            std::vector<int64_t> pk(np);
            for (auto& v : pk) v = static_cast<int64_t>(next() % (5 * NB * 3));  // a third of the key space is populated at stride 5
            std::size_t expect = 0;
            for (auto v : pk) expect += (v % 5 == 2 && static_cast<std::size_t>(v / 5) < NB) ? 1 : 0;
            rmm::device_buffer pd{pk.data(), np * sizeof(int64_t), sv};
            column_view pc{data_type{type_id::INT64}, static_cast<size_type>(np), pd.data(), nullptr, 0};
            table_view pt{{pc}};
            for (int rep = 0; rep < 2; ++rep) {
              auto [l, r] = hj.inner_join(pt, std::nullopt, sv);
              sv.synchronize();
              auto hl = to_host(*l);
              auto hr = to_host(*r);
              if (hl.size() != expect) throw std::runtime_error("inner_join: wrong number of pairs");
              std::vector<uint8_t> hit(np, 0);
              for (std::size_t i = 0; i < hl.size(); ++i) {
                auto const v = pk[hl[i]];
                if (hit[hl[i]] || v % 5 != 2 || row_of[static_cast<std::size_t>(v / 5)] != hr[i]) throw std::runtime_error("inner_join: wrong pair");
                hit[hl[i]] = 1;
              }
              if (hj.inner_join_size(pt, sv) != expect) throw std::runtime_error("inner_join_size");
            }
            auto [ll, lr] = hj.left_join(pt, std::nullopt, sv);
            sv.synchronize();
            if (ll->size() != np) throw std::runtime_error("left_join: one row per probe row expected (distinct build keys)");
            // the direct kernel on a slice of the same column
            column_view small{data_type{type_id::INT64}, 100'000, pd.data(), nullptr, 0};
            auto [sl, sr] = hj.inner_join(table_view{{small}}, std::nullopt, sv);
            sv.synchronize();
            std::size_t se = 0;
            for (std::size_t i = 0; i < 100'000; ++i) se += (pk[i] % 5 == 2 && static_cast<std::size_t>(pk[i] / 5) < NB) ? 1 : 0;
            if (sl->size() != se) throw std::runtime_error("small probe: wrong number of pairs");
          }
          sv.synchronize();
          (void)hipStreamDestroy(s);
        } catch (std::exception const& e) {
          errs[t] = e.what();
        }
      });
    for (auto& t : th) t.join();
    for (int t = 0; t < T; ++t) {
      if (!errs[t].empty()) std::printf("    thread %d: %s\n", t, errs[t].c_str());
      CHECK(errs[t].empty());
    }
  });

  run("a lost look-back chain is an EXCEPTION, not a dead process -- and the sorts stay stream-ordered (gx_sort.hip spin_guard; utilities/error.hpp:63-86)", [] {
    // TEST HOOK: tile 3 of every look-back pass never publishes its granules; the wait limit is cut from 30 s to 150 ms.  The
    // successors' waits are abandoned and the scratch's status word says so.  Round 6: cudf::sorted_order / cudf::sort RETURN (their
    // work is queued, as the reference's is: sort.cu:52-89); the fault surfaces as cudf::cuda_error at cudf_amd::check_device_faults
    // (which waits for the stream) or at the next sort call once the faulting sort has run -- and the same calls succeed right after,
    // in the same process, on the same HIP context.
    constexpr std::size_t N = 3'000'000;
    std::vector<int64_t> k(N);
    uint64_t x = 88172645463325252ull;
    for (auto& v : k) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = static_cast<int64_t>(x); }
    auto c = make_col<int64_t>(k);
    table_view t{{c->view()}};
    gx_sort_set_spin_limit_ms(150);
    gx_sort_inject_lost_tile(3);
    bool threw_order = false, threw_sort = false, returned_order = false, returned_sort = false;
    try {
      auto o         = cudf::sorted_order(t);
      returned_order = true;  // no throw from the call that queued the faulting work
      cudf_amd::check_device_faults(get_default_stream());
    } catch (cudf::cuda_error const& e) {
      threw_order = e.error_code() == GX_EINTERNAL;
    }
    try {
      auto o        = cudf::sort(t);
      returned_sort = true;
      get_default_stream().synchronize();
      auto o2 = cudf::sort(t);  // the NEXT sort call reports the completed one's fault
    } catch (cudf::cuda_error const& e) {
      threw_sort = e.error_code() == GX_EINTERNAL;
    }
    gx_sort_inject_lost_tile(-1);
    gx_sort_set_spin_limit_ms(0);
    CHECK(returned_order);
    CHECK(returned_sort);
    CHECK(threw_order);
    CHECK(threw_sort);
    try { cudf_amd::check_device_faults(get_default_stream()); } catch (cudf::cuda_error const&) {}  // (the second faulting sort of the block above, if it ran)
    auto o  = cudf::sorted_order(t);
    auto so = cudf::sort(t);
    cudf_amd::check_device_faults(get_default_stream());  // waits; nothing to report
    auto ho = to_host<int32_t>(o->view());
    std::vector<int32_t> ref(N);
    std::iota(ref.begin(), ref.end(), 0);
    std::stable_sort(ref.begin(), ref.end(), [&](int32_t a, int32_t b) { return k[a] < k[b]; });
    CHECK(std::equal(ref.begin(), ref.end(), ho.begin()));
    std::vector<int64_t> hs(N);
    CHECK(hipMemcpy(hs.data(), so->view().column(0).head<int64_t>(), N * sizeof(int64_t), hipMemcpyDeviceToHost) == hipSuccess);
    std::sort(k.begin(), k.end());
    CHECK(hs == k);
  });

  run("round_robin_partition: the nine documented examples (partitioning.hpp:183-275), errors (round_robin.cu:160-166); minmax (reduction_tests.cpp:122-243)", [] {
    auto iota = [](int n) { std::vector<int32_t> v(n); std::iota(v.begin(), v.end(), 0); return v; };
    struct Ex { int rows, parts, start; std::vector<int32_t> table; std::vector<size_type> offsets; };
    std::vector<Ex> const ex = {
      {13, 3, 0, {0, 3, 6, 9, 12, 1, 4, 7, 10, 2, 5, 8, 11}, {0, 5, 9, 13}},
      {13, 3, 1, {2, 5, 8, 11, 0, 3, 6, 9, 12, 1, 4, 7, 10}, {0, 4, 9, 13}},
      {11, 3, 0, {0, 3, 6, 9, 1, 4, 7, 10, 2, 5, 8}, {0, 4, 8, 11}},
      {11, 3, 1, {2, 5, 8, 0, 3, 6, 9, 1, 4, 7, 10}, {0, 3, 7, 11}},
      {11, 3, 2, {1, 4, 7, 10, 2, 5, 8, 0, 3, 6, 9}, {0, 4, 7, 11}},
      {11, 15, 2, {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10}, {0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 11, 11}},
      {11, 15, 10, {5, 6, 7, 8, 9, 10, 0, 1, 2, 3, 4}, {0, 1, 2, 3, 4, 5, 6, 6, 6, 6, 6, 7, 8, 9, 10, 11}},
      {11, 15, 14, {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 0}, {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10, 10, 10, 10, 11}},
      {11, 11, 2, {9, 10, 0, 1, 2, 3, 4, 5, 6, 7, 8}, {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11}},
    };
    for (auto const& e : ex) {
      auto c = make_col<int32_t>(iota(e.rows));
      std::vector<double> dv(e.rows);
      for (int i = 0; i < e.rows; ++i) dv[i] = 0.5 * i;
      auto d = make_col<double>(dv);
      auto [t, offs] = round_robin_partition(table_view{{c->view(), d->view()}}, e.parts, e.start);
      CHECK(offs == e.offsets);
      CHECK(to_host<int32_t>(t->get_column(0).view()) == e.table);
      auto hd = to_host<double>(t->get_column(1).view());
      for (int i = 0; i < e.rows; ++i) CHECK(hd[i] == 0.5 * e.table[i]);
    }
    auto c = make_col<int32_t>(iota(5));
    auto e0 = make_col<int32_t>({});
    auto [te, oe] = round_robin_partition(table_view{{e0->view()}}, 5, 0);              // round_robin_test.cpp:44-55 EmptyInput
    CHECK(te->num_rows() == 0 && oe.size() == 6);
    CHECK(throws<cudf::logic_error>([&] { (void)round_robin_partition(table_view{{c->view()}}, 0, 0); }));
    CHECK(throws<cudf::logic_error>([&] { (void)round_robin_partition(table_view{{c->view()}}, 3, 3); }));
    CHECK(throws<cudf::logic_error>([&] { (void)round_robin_partition(table_view{{c->view()}}, 3, -1); }));
    // at size: against the closed form
    int const n = 1'000'003, P = 37, S = 11;
    auto big = make_col<int32_t>(iota(n));
    auto [tb, ob] = round_robin_partition(table_view{{big->view()}}, P, S);
    auto hb = to_host<int32_t>(tb->get_column(0).view());
    std::size_t pos = 0;
    bool ok = ob.size() == static_cast<std::size_t>(P) + 1 && ob.back() == n;
    for (int p = 0; p < P && ok; ++p) {
      ok = ok && static_cast<std::size_t>(ob[p]) == pos;
      for (int i = ((p - S) % P + P) % P; i < n; i += P) ok = ok && hb[pos++] == i;
    }
    CHECK(ok);
    // minmax: reduction_tests.cpp:125-243 (values, the null mask of the test, an all-null column)
    std::vector<int> iv{5, 0, -120, -111, 0, 64, 63, 99, 123, -16};
    std::vector<int> valid{1, 1, 0, 1, 1, 1, 0, 1, 0, 1};
    auto mm_check = [&](auto tag) {
      using T = decltype(tag);
      std::vector<T> v(iv.begin(), iv.end());
      auto col = make_col<T>(v);
      auto [lo, hi] = cudf::minmax(col->view());
      CHECK(lo->is_valid() && hi->is_valid());
      CHECK(static_cast<numeric_scalar<T>*>(lo.get())->value() == T(-120) && static_cast<numeric_scalar<T>*>(hi.get())->value() == T(123));
      auto coln = make_col<T>(v, valid);
      auto [ln, hn] = cudf::minmax(coln->view());
      CHECK(static_cast<numeric_scalar<T>*>(ln.get())->value() == T(-111) && static_cast<numeric_scalar<T>*>(hn.get())->value() == T(99));
      auto colz = make_col<T>(v, std::vector<int>(v.size(), 0));
      auto [lz, hz] = cudf::minmax(colz->view());
      CHECK(!lz->is_valid() && !hz->is_valid());
    };
    mm_check(int32_t{});
    mm_check(int64_t{});
    mm_check(double{});
    mm_check(float{});
  });

  run("is_sorted (sorting.hpp:83-86; is_sorted_tests.cpp:273-466, the fixed-width numeric cases)", [] {
    using O = cudf::order;
    using N = cudf::null_order;
    auto check_type = [](auto tag) {
      using T = decltype(tag);
      std::vector<T> asc = std::is_signed_v<T> ? std::vector<T>{std::numeric_limits<T>::lowest(), T(-100), T(-10), T(-1), T(0), T(1), T(10), T(100), std::numeric_limits<T>::max()}
                                               : std::vector<T>{std::numeric_limits<T>::lowest(), T(0), T(1), T(10), T(100), std::numeric_limits<T>::max()};
      std::vector<T> desc(asc.rbegin(), asc.rend());
      auto a = make_col<T>(asc);
      auto d = make_col<T>(desc);
      CHECK(is_sorted(table_view{{a->view()}}, {O::ASCENDING}, {}));      // :299-311 Ascending
      CHECK(!is_sorted(table_view{{d->view()}}, {O::ASCENDING}, {}));     // :313-326 AscendingFalse
      CHECK(is_sorted(table_view{{d->view()}}, {O::DESCENDING}, {}));     // :328-340 Descending
      CHECK(!is_sorted(table_view{{a->view()}}, {O::DESCENDING}, {}));    // :342-355 DescendingFalse
      auto na = make_col<T>({T(0), T(0)}, {1, 0});                        // nulls_after  :73-77
      auto nb = make_col<T>({T(0), T(0)}, {0, 1});                        // nulls_before :79-83
      CHECK(is_sorted(table_view{{na->view()}}, {}, {N::AFTER}));         // :357-369
      CHECK(!is_sorted(table_view{{nb->view()}}, {}, {N::AFTER}));        // :371-383
      CHECK(is_sorted(table_view{{nb->view()}}, {}, {N::BEFORE}));        // :385-397
      CHECK(!is_sorted(table_view{{na->view()}}, {}, {N::BEFORE}));       // :399-411
      CHECK(throws<cudf::logic_error>([&] { (void)is_sorted(table_view{{a->view(), a->view()}}, {O::ASCENDING}, {}); }));             // :413-424 OrderArgsTooFew
      CHECK(throws<cudf::logic_error>([&] { (void)is_sorted(table_view{{a->view()}}, {O::ASCENDING, O::ASCENDING}, {}); }));           // :426-436
      CHECK(throws<cudf::logic_error>([&] { (void)is_sorted(table_view{{nb->view(), nb->view()}}, {}, {N::BEFORE}); }));               // :438-449
      CHECK(throws<cudf::logic_error>([&] { (void)is_sorted(table_view{{nb->view()}}, {}, {N::BEFORE, N::BEFORE}); }));                // :451-461
      auto e = make_col<T>({});
      CHECK(is_sorted(table_view{{e->view(), e->view()}}, {O::ASCENDING, O::DESCENDING}, {}));   // :281-297 NoRows
    };
    check_type(int8_t{});
    check_type(int32_t{});
    check_type(int64_t{});
    check_type(uint16_t{});
    check_type(uint64_t{});
    check_type(float{});
    check_type(double{});
    CHECK(is_sorted(table_view{std::vector<column_view>{}}, {}, {}));       // :273-279 NoColumns
    // floats: -0.0 and +0.0 are equal, NaN is the greatest value and every NaN equals every other (the sort's comparator)
    double const nan = std::numeric_limits<double>::quiet_NaN();
    auto f = make_col<double>({-1.0, 0.0, -0.0, 0.0, 3.5, nan, -nan});
    CHECK(is_sorted(table_view{{f->view()}}, {O::ASCENDING}, {}));
    auto g = make_col<double>({-1.0, nan, 3.5});
    CHECK(!is_sorted(table_view{{g->view()}}, {O::ASCENDING}, {}));
    // two columns, mixed directions; then a tie in the first column broken the wrong way by the second
    auto c1 = make_col<int32_t>({1, 1, 2, 2, 3});
    auto c2 = make_col<int64_t>({9, 7, 5, 5, 0});
    CHECK(is_sorted(table_view{{c1->view(), c2->view()}}, {O::ASCENDING, O::DESCENDING}, {}));
    CHECK(!is_sorted(table_view{{c1->view(), c2->view()}}, {O::ASCENDING, O::ASCENDING}, {}));
    // at size: a sorted column with ties; one pair swapped far inside
    std::vector<int64_t> big(3'000'001);
    for (std::size_t i = 0; i < big.size(); ++i) big[i] = static_cast<int64_t>(i / 3) - 500'000;
    auto b1 = make_col<int64_t>(big);
    CHECK(is_sorted(table_view{{b1->view()}}, {}, {}));
    std::swap(big[2'000'000], big[2'000'003]);
    auto b2 = make_col<int64_t>(big);
    CHECK(!is_sorted(table_view{{b2->view()}}, {}, {}));
    std::vector<int> valid(big.size(), 1);
    valid[0] = valid[1] = 0;
    std::swap(big[2'000'000], big[2'000'003]);
    auto b3 = make_col<int64_t>(big, valid);
    CHECK(is_sorted(table_view{{b3->view()}}, {}, {N::BEFORE}));
    CHECK(!is_sorted(table_view{{b3->view()}}, {}, {N::AFTER}));
  });

  run("two sorts on two streams are queued back to back: cudf::sort does not wait for the device (sort.cu:52-89)", [] {
    // 2^26 random int64 keys per stream (cursor path).  Both calls are issued from one host thread without any synchronisation in
    // between; the host time of the two calls must be a fraction of the device time of the two sorts (round 5 read a status word
    // back inside every call: issue time == device time), and both results must be sorted permutations of their inputs.
    constexpr std::size_t N = std::size_t{1} << 26;
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking) == hipSuccess);
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) == hipSuccess);
    std::vector<int64_t> k(N);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (auto& v : k) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = static_cast<int64_t>(x); }
    auto a = make_col<int64_t>(k);
    for (auto& v : k) v = ~v;
    auto b = make_col<int64_t>(k);
    table_view ta{{a->view()}}, tb{{b->view()}};
    { auto w1 = cudf::sort(ta, {}, {}, s1); auto w2 = cudf::sort(tb, {}, {}, s2); }  // warm-up: arena blocks of both streams, module load
    CHECK(hipDeviceSynchronize() == hipSuccess);
    auto const t0 = std::chrono::steady_clock::now();
    auto ra = cudf::sort(ta, {}, {}, s1);
    auto rb = cudf::sort(tb, {}, {}, s2);
    auto const t1 = std::chrono::steady_clock::now();
    CHECK(hipStreamSynchronize(s1) == hipSuccess);
    CHECK(hipStreamSynchronize(s2) == hipSuccess);
    auto const t2 = std::chrono::steady_clock::now();
    cudf_amd::poll_device_faults();
    double const issue_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    double const total_ms = std::chrono::duration<double, std::milli>(t2 - t0).count();
    std::printf("    issue of both calls %.3f ms, until both streams drained %.3f ms\n", issue_ms, total_ms);
    CHECK(issue_ms < 0.6 * total_ms);
    std::vector<int64_t> ha(N), hb(N);
    CHECK(hipMemcpy(ha.data(), ra->view().column(0).head<int64_t>(), N * 8, hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(hipMemcpy(hb.data(), rb->view().column(0).head<int64_t>(), N * 8, hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(std::is_sorted(ha.begin(), ha.end()));
    CHECK(std::is_sorted(hb.begin(), hb.end()));
    uint64_t sa = 0, sb = 0, sk = 0;
    for (std::size_t i = 0; i < N; ++i) { sa += static_cast<uint64_t>(ha[i]); sb += static_cast<uint64_t>(hb[i]); sk += static_cast<uint64_t>(k[i]); }
    CHECK(sb == sk);                          // b's keys are the ones left in k
    CHECK(sa == static_cast<uint64_t>(0) - sk - N);  // a = ~b element-wise: sum(~v) = -sum(v) - N
    ra.reset();
    rb.reset();
    CHECK(hipStreamDestroy(s1) == hipSuccess);
    CHECK(hipStreamDestroy(s2) == hipSuccess);
  });

  std::printf("%d run, %d failed\n", g_run, g_failed);
  return g_failed ? 1 : 0;
}
