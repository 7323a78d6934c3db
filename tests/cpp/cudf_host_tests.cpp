// Host-only checks of the C++ drop-in surface: everything here is validated / computed BEFORE the first
// device call, so the binary runs without a GPU (tests/test_cpp_api.py::test_host_side_validation).
// The reference's tests pin the same behaviour: column_view_test.cpp / table_view tests (constructor checks,
// src/column/column_view.cpp:101-132), join_tests.cpp + hash_join.cu:49-58 (argument errors),
// groupby.cu:226-230 (size mismatch), sort_test.cpp MismatchInColumnOrderSize / MismatchInNullPrecedenceSize.
#include <cudf/aggregation.hpp>
#include <cudf/column/column_view.hpp>
#include <cudf/groupby.hpp>
#include <cudf/interop.hpp>
#include <cudf/join/distinct_hash_join.hpp>
#include <cudf/join/filtered_join.hpp>
#include <cudf/join/hash_join.hpp>
#include <cudf/sorting.hpp>
#include <cudf/table/table_view.hpp>
#include <cudf/types.hpp>

#include <cstdio>
#include <functional>
#include <stdexcept>
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

int main()
{
  // a "device pointer" that is never dereferenced on the host
  void const* fake = reinterpret_cast<void const*>(0x10000);
  auto const* fake_mask = reinterpret_cast<bitmask_type const*>(0x20000);

  run("types: sizes, traits, type_to_id (types.hpp:99-216, 278-341)", [] {
    CHECK(size_of(data_type{type_id::INT8}) == 1 && size_of(data_type{type_id::UINT16}) == 2);
    CHECK(size_of(data_type{type_id::FLOAT32}) == 4 && size_of(data_type{type_id::INT64}) == 8);
    CHECK(is_floating_point(data_type{type_id::FLOAT64}) && !is_floating_point(data_type{type_id::INT32}));
    CHECK(type_to_id<int32_t>() == type_id::INT32 && type_to_id<double>() == type_id::FLOAT64);
    CHECK(JoinNoMatch == std::numeric_limits<size_type>::min());
  });
  run("column_view / table_view constructor checks (column_view.cpp:101-132, table_view.cpp)", [&] {
    column_view ok{data_type{type_id::INT32}, 5, fake, fake_mask, 2, 1};
    CHECK(ok.size() == 5 && ok.offset() == 1 && ok.null_count() == 2 && ok.nullable() && ok.has_nulls());
    CHECK(ok.data<int32_t>() == static_cast<int32_t const*>(fake) + 1);
    CHECK(throws<cudf::logic_error>([&] { column_view{data_type{type_id::INT32}, -1, fake, nullptr, 0}; }));
    CHECK(throws<cudf::logic_error>([&] { column_view{data_type{type_id::INT32}, 3, nullptr, nullptr, 0}; }));             // null data
    CHECK(throws<cudf::logic_error>([&] { column_view{data_type{type_id::INT32}, 3, fake, nullptr, 1}; }));     // nulls without a mask
    CHECK(throws<cudf::logic_error>([&] { column_view{data_type{type_id::INT32}, 3, fake, nullptr, 0, -1}; })); // negative offset
    column_view a{data_type{type_id::INT64}, 4, fake, nullptr, 0}, b{data_type{type_id::INT64}, 5, fake, nullptr, 0};
    CHECK(throws<cudf::logic_error>([&] { table_view{{a, b}}; }));                                              // column size mismatch
    table_view t{{a, a}};
    CHECK(t.num_columns() == 2 && t.num_rows() == 4);
    CHECK(t.select({1}).num_columns() == 1);
  });
  run("sort argument checks (sort_impl.cuh:45-52; sort_test.cpp Mismatch*)", [&] {
    column_view a{data_type{type_id::INT64}, 4, fake, nullptr, 0};
    table_view t{{a, a}};
    CHECK(throws<std::invalid_argument>([&] { (void)sorted_order(t, {order::ASCENDING}); }));
    CHECK(throws<std::invalid_argument>([&] { (void)sorted_order(t, {}, {null_order::BEFORE}); }));
    CHECK(throws<std::invalid_argument>([&] { (void)sort(t, {order::ASCENDING, order::ASCENDING, order::ASCENDING}); }));
    column_view v{data_type{type_id::INT32}, 3, fake, nullptr, 0};
    CHECK(throws<std::invalid_argument>([&] { (void)sort_by_key(table_view{{v}}, t); }));                       // 3 value rows, 4 key rows
  });
  run("join argument checks (hash_join.cu:49-58, filtered_join.hpp:74-76, distinct_hash_join.hpp)", [&] {
    column_view a{data_type{type_id::INT64}, 0, nullptr, nullptr, 0};
    CHECK(throws<std::invalid_argument>([&] { hash_join hj{table_view{}, null_equality::EQUAL}; }));
    CHECK(throws<std::invalid_argument>([&] { hash_join hj{table_view{{a}}, nullable_join::NO, null_equality::EQUAL, 0.0}; }));
    CHECK(throws<std::invalid_argument>([&] { hash_join hj{table_view{{a}}, nullable_join::NO, null_equality::EQUAL, 1.5}; }));
    CHECK(throws<std::invalid_argument>([&] { distinct_hash_join dj{table_view{}}; }));
    CHECK(throws<std::invalid_argument>([&] { distinct_hash_join dj{table_view{{a}}, null_equality::EQUAL, -1.0}; }));
    CHECK(throws<std::invalid_argument>([&] { filtered_join fj{table_view{{a}}, null_equality::EQUAL, 0.0, get_default_stream()}; }));
  });
  run("groupby request checks (groupby.cu:226-230)", [&] {
    column_view keys{data_type{type_id::INT32}, 4, fake, nullptr, 0}, vals{data_type{type_id::FLOAT64}, 5, fake, nullptr, 0};
    groupby::groupby gb{table_view{{keys}}};
    std::vector<groupby::aggregation_request> reqs(1);
    reqs[0].values = vals;
    reqs[0].aggregations.emplace_back(make_sum_aggregation<groupby_aggregation>());
    CHECK(throws<std::invalid_argument>([&] { (void)gb.aggregate(reqs); }));
  });
  run("aggregation objects: kinds, ddof, equality, clone (aggregation.hpp)", [] {
    auto v1 = make_variance_aggregation<groupby_aggregation>();
    auto v2 = make_variance_aggregation<groupby_aggregation>(2);
    auto s1 = make_std_aggregation<groupby_aggregation>();
    CHECK(v1->kind == aggregation::VARIANCE && s1->kind == aggregation::STD);
    CHECK(!v1->is_equal(*v2) && !v1->is_equal(*s1) && v1->is_equal(*make_variance_aggregation<groupby_aggregation>(1)));
    auto c = v2->clone();
    CHECK(c->kind == aggregation::VARIANCE && c->is_equal(*v2) && c->do_hash() == v2->do_hash());
    CHECK(make_count_aggregation<groupby_aggregation>()->kind == aggregation::COUNT_VALID);
    CHECK(make_count_aggregation<groupby_aggregation>(null_policy::INCLUDE)->kind == aggregation::COUNT_ALL);
    CHECK(make_argmin_aggregation<groupby_aggregation>()->kind == aggregation::ARGMIN);
    CHECK(make_m2_aggregation<groupby_aggregation>()->kind == aggregation::M2);
  });
  run("Arrow schema export is host-only (interop.hpp:477-480)", [&] {
    column_view a{data_type{type_id::INT64}, 4, fake, nullptr, 0}, b{data_type{type_id::FLOAT32}, 4, fake, fake_mask, 1};
    std::vector<column_metadata> meta{{"k"}, {"v"}};
    auto s = to_arrow_schema(table_view{{a, b}}, meta);
    CHECK(std::string{s->format} == "+s" && s->n_children == 2 && s->release != nullptr);
    CHECK(std::string{s->children[0]->format} == "l" && std::string{s->children[0]->name} == "k" && s->children[0]->flags == 0);
    CHECK(std::string{s->children[1]->format} == "f" && (s->children[1]->flags & ARROW_FLAG_NULLABLE));
    CHECK(throws<std::invalid_argument>([&] { (void)to_arrow_schema(table_view{{a}}, meta); }));
    column_view bad{data_type{type_id::BOOL8}, 4, fake, nullptr, 0};
    CHECK(throws<cudf::data_type_error>([&] { (void)to_arrow_schema(table_view{{bad}}, std::vector<column_metadata>{{"x"}}); }));
    CHECK(throws<std::invalid_argument>([&] { (void)from_arrow_device(nullptr, nullptr); }));
  });
  std::printf("%d run, %d failed\n", g_run, g_failed);
  return g_failed ? 1 : 0;
}
