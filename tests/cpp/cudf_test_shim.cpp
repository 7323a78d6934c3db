// tests/cpp/cudf_test_shim.cpp -- TEST INFRASTRUCTURE: a flat extern "C" face over the cudf:: C++ surface
// (include/cudf/*.hpp -> cudf_amd/libcudf.so -> the gx_* C ABI -> HIP kernels), so that the Python parity tests can
// drive cudf::rank / top_k / segmented sort / sort-path groupby / groupby::scan / shift / replace_nulls / get_groups /
// hash_join match contexts / full_join on 1e5..1e7-row inputs and compare them with the NumPy oracle
// (oracle/cudf_oracle.py).  Nothing in the product links or loads this file.
//
// Conventions: every pointer is a DEVICE pointer unless it ends in _host; validity = Arrow bitmap (uint32 words,
// LSB first, 1 = valid) or NULL; outputs are caller-allocated with room for `n` rows (masks: ceil(n/32) words);
// the return value is 0 or -1 with the exception text in shim_last_error().
#include <cudf/aggregation.hpp>
#include <cudf/column/column.hpp>
#include <cudf/column/column_view.hpp>
#include <cudf/groupby.hpp>
#include <cudf/join/hash_join.hpp>
#include <cudf/join/join.hpp>
#include <cudf/null_mask.hpp>
#include <cudf/partitioning.hpp>
#include <cudf/reduction.hpp>
#include <cudf/scalar/scalar.hpp>
#include <cudf/sorting.hpp>
#include <cudf/table/table_view.hpp>

#include <hip/hip_runtime_api.h>

#include <cstring>
#include <functional>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

template <typename F>
int guarded(F&& f)
{
  try {
    f();
    return 0;
  } catch (std::exception const& e) {
    g_err = e.what();
    return -1;
  }
}

cudf::column_view view(int dtype, void const* data, uint32_t const* valid, int n, int nulls)
{
  return cudf::column_view{cudf::data_type{static_cast<cudf::type_id>(dtype)}, n, data, valid, valid ? nulls : 0};
}

void d2d(void* dst, void const* src, std::size_t bytes)
{
  if (bytes == 0) return;
  if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, cudf::get_default_stream().value()) != hipSuccess)
    throw std::runtime_error("shim: device copy failed");
}

// copy a result column out: data, optional validity, null count
void emit(cudf::column_view const& c, void* out, uint32_t* out_valid, int* out_nulls_host)
{
  d2d(out, c.head<char>(), static_cast<std::size_t>(c.size()) * cudf::size_of(c.type()));
  if (out_nulls_host) *out_nulls_host = c.null_count();
  if (out_valid) {
    if (c.size() == 0) return;
    if (c.nullable()) {
      if (c.offset() != 0) throw std::runtime_error("shim: sliced result column");
      d2d(out_valid, c.null_mask(), static_cast<std::size_t>((c.size() + 31) / 32) * 4);  // allocations are padded to 64 B
    } else if (hipMemsetAsync(out_valid, 0xFF, static_cast<std::size_t>((c.size() + 31) / 32) * 4,
                              cudf::get_default_stream().value()) != hipSuccess) {
      throw std::runtime_error("shim: memset failed");
    }
  }
}

void shim_sync() { cudf::get_default_stream().synchronize(); }

std::unique_ptr<cudf::groupby_aggregation> make_agg(int kind, int param)
{
  using A = cudf::aggregation;
  switch (static_cast<A::Kind>(kind)) {
    case A::SUM: return cudf::make_sum_aggregation<cudf::groupby_aggregation>();
    case A::PRODUCT: return cudf::make_product_aggregation<cudf::groupby_aggregation>();
    case A::MIN: return cudf::make_min_aggregation<cudf::groupby_aggregation>();
    case A::MAX: return cudf::make_max_aggregation<cudf::groupby_aggregation>();
    case A::COUNT_VALID: return cudf::make_count_aggregation<cudf::groupby_aggregation>(cudf::null_policy::EXCLUDE);
    case A::COUNT_ALL: return cudf::make_count_aggregation<cudf::groupby_aggregation>(cudf::null_policy::INCLUDE);
    case A::MEAN: return cudf::make_mean_aggregation<cudf::groupby_aggregation>();
    case A::SUM_OF_SQUARES: return cudf::make_sum_of_squares_aggregation<cudf::groupby_aggregation>();
    case A::M2: return cudf::make_m2_aggregation<cudf::groupby_aggregation>();
    case A::VARIANCE: return cudf::make_variance_aggregation<cudf::groupby_aggregation>(param);
    case A::STD: return cudf::make_std_aggregation<cudf::groupby_aggregation>(param);
    case A::ARGMAX: return cudf::make_argmax_aggregation<cudf::groupby_aggregation>();
    case A::ARGMIN: return cudf::make_argmin_aggregation<cudf::groupby_aggregation>();
    case A::NTH_ELEMENT: return cudf::make_nth_element_aggregation<cudf::groupby_aggregation>(param, cudf::null_policy::INCLUDE);
    default: throw std::invalid_argument("shim: aggregation kind");
  }
}

}  // namespace

extern "C" {

const char* shim_last_error() { return g_err.c_str(); }

// cudf::rank (sorting.hpp; reference cpp/src/sort/rank.cu:59-369).  out: INT32, or FLOAT64 when percentage != 0 or
// method == AVERAGE.
int shim_rank(int dtype, const void* data, const uint32_t* valid, int n, int nulls, int method, int descending, int null_include,
              int nulls_before, int percentage, void* out, uint32_t* out_valid, int* out_nulls_host)
{
  return guarded([&] {
    auto r = cudf::rank(view(dtype, data, valid, n, nulls), static_cast<cudf::rank_method>(method),
                        descending ? cudf::order::DESCENDING : cudf::order::ASCENDING,
                        null_include ? cudf::null_policy::INCLUDE : cudf::null_policy::EXCLUDE,
                        nulls_before ? cudf::null_order::BEFORE : cudf::null_order::AFTER, percentage != 0);
    emit(r->view(), out, out_valid, out_nulls_host);
    shim_sync();
  });
}

// cudf::top_k + cudf::top_k_order (reference cpp/src/sort/top_k.cu:118-165)
int shim_top_k(int dtype, const void* data, const uint32_t* valid, int n, int nulls, int k, int descending, void* out_vals,
               int32_t* out_idx, int* out_n_host)
{
  return guarded([&] {
    auto const o = descending ? cudf::order::DESCENDING : cudf::order::ASCENDING;
    auto col     = view(dtype, data, valid, n, nulls);
    auto v       = cudf::top_k(col, k, o);
    auto i       = cudf::top_k_order(col, k, o);
    if (v->size() != i->size()) throw std::runtime_error("shim: top_k and top_k_order sizes differ");
    emit(v->view(), out_vals, nullptr, nullptr);
    emit(i->view(), out_idx, nullptr, nullptr);
    *out_n_host = v->size();
    shim_sync();
  });
}

// cudf::(stable_)segmented_sorted_order of a key table (reference cpp/src/sort/segmented_sort_impl.cuh:152-324)
int shim_segmented_sorted_order(int ncols, const int* dtypes_host, const void* const* datas_host, const uint32_t* const* valids_host,
                                const int* nulls_host, int n, const int32_t* offsets, int noffsets, const int* descending_host,
                                const int* nulls_before_host, int stable, int32_t* out)
{
  return guarded([&] {
    std::vector<cudf::column_view> cols;
    std::vector<cudf::order> ord;
    std::vector<cudf::null_order> prec;
    for (int i = 0; i < ncols; ++i) {
      cols.push_back(view(dtypes_host[i], datas_host[i], valids_host[i], n, nulls_host[i]));
      ord.push_back(descending_host[i] ? cudf::order::DESCENDING : cudf::order::ASCENDING);
      prec.push_back(nulls_before_host[i] ? cudf::null_order::BEFORE : cudf::null_order::AFTER);
    }
    cudf::column_view off{cudf::data_type{cudf::type_id::INT32}, noffsets, offsets, nullptr, 0};
    auto r = stable ? cudf::stable_segmented_sorted_order(cudf::table_view{cols}, off, ord, prec)
                    : cudf::segmented_sorted_order(cudf::table_view{cols}, off, ord, prec);
    emit(r->view(), out, nullptr, nullptr);
    shim_sync();
  });
}

// cudf::sorted_order / stable_sorted_order of a key table (reference cpp/src/sort/sort_impl.cuh:31-95, sort.cu:22-29, stable_sort.cu)
int shim_table_sorted_order(int ncols, const int* dtypes_host, const void* const* datas_host, const uint32_t* const* valids_host,
                            const int* nulls_host, int n, const int* descending_host, const int* nulls_before_host, int stable, int32_t* out)
{
  return guarded([&] {
    std::vector<cudf::column_view> cols;
    std::vector<cudf::order> ord;
    std::vector<cudf::null_order> prec;
    for (int i = 0; i < ncols; ++i) {
      cols.push_back(view(dtypes_host[i], datas_host[i], valids_host ? valids_host[i] : nullptr, n, nulls_host ? nulls_host[i] : 0));
      ord.push_back(descending_host[i] ? cudf::order::DESCENDING : cudf::order::ASCENDING);
      prec.push_back(nulls_before_host && !nulls_before_host[i] ? cudf::null_order::AFTER : cudf::null_order::BEFORE);
    }
    auto r = stable ? cudf::stable_sorted_order(cudf::table_view{cols}, ord, prec) : cudf::sorted_order(cudf::table_view{cols}, ord, prec);
    emit(r->view(), out, nullptr, nullptr);
    shim_sync();
  });
}

// cudf::groupby::groupby(keys, null_policy, sorted).aggregate({values, {agg}}) with ONE key column.
// force_sort != 0 adds an NTH_ELEMENT(0) aggregation to the request, the way the reference's tests force the
// sort-based path (cpp/tests/groupby/groupby_test_util.hpp force_use_sort_impl).  out_* have room for n groups.
int shim_groupby_aggregate(int key_dtype, const void* keys, const uint32_t* kvalid, int knulls, int val_dtype, const void* vals,
                           const uint32_t* vvalid, int vnulls, int n, int kind, int param, int null_include, int keys_sorted,
                           int force_sort, void* out_keys, uint32_t* out_keys_valid, int* out_keys_nulls_host, void* out_vals,
                           uint32_t* out_vals_valid, int* out_vals_nulls_host, int* out_groups_host, int* out_val_dtype_host)
{
  return guarded([&] {
    auto kc = view(key_dtype, keys, kvalid, n, knulls);
    cudf::groupby::groupby gb(cudf::table_view{{kc}}, null_include ? cudf::null_policy::INCLUDE : cudf::null_policy::EXCLUDE,
                              keys_sorted ? cudf::sorted::YES : cudf::sorted::NO);
    std::vector<cudf::groupby::aggregation_request> reqs(1);
    reqs[0].values = view(val_dtype, vals, vvalid, n, vnulls);
    reqs[0].aggregations.push_back(make_agg(kind, param));
    if (force_sort) reqs[0].aggregations.push_back(cudf::make_nth_element_aggregation<cudf::groupby_aggregation>(0));
    auto res      = gb.aggregate(reqs);
    auto const& k = res.first->get_column(0);
    auto const& v = *res.second[0].results[0];
    emit(k.view(), out_keys, out_keys_valid, out_keys_nulls_host);
    emit(v.view(), out_vals, out_vals_valid, out_vals_nulls_host);
    *out_groups_host    = k.size();
    *out_val_dtype_host = static_cast<int>(v.type().id());
    shim_sync();
  });
}

// groupby::scan with ONE key column and one aggregation (reference cpp/src/groupby/sort/scan.cpp:60-238)
int shim_groupby_scan(int key_dtype, const void* keys, const uint32_t* kvalid, int knulls, int val_dtype, const void* vals,
                      const uint32_t* vvalid, int vnulls, int n, int kind, int null_include, int keys_sorted, void* out_keys,
                      void* out_vals, uint32_t* out_vals_valid, int* out_vals_nulls_host, int* out_rows_host, int* out_val_dtype_host)
{
  return guarded([&] {
    auto kc = view(key_dtype, keys, kvalid, n, knulls);
    cudf::groupby::groupby gb(cudf::table_view{{kc}}, null_include ? cudf::null_policy::INCLUDE : cudf::null_policy::EXCLUDE,
                              keys_sorted ? cudf::sorted::YES : cudf::sorted::NO);
    std::vector<cudf::groupby::scan_request> reqs(1);
    reqs[0].values = view(val_dtype, vals, vvalid, n, vnulls);
    using A        = cudf::aggregation;
    switch (static_cast<A::Kind>(kind)) {
      case A::SUM: reqs[0].aggregations.push_back(cudf::make_sum_aggregation<cudf::groupby_scan_aggregation>()); break;
      case A::MIN: reqs[0].aggregations.push_back(cudf::make_min_aggregation<cudf::groupby_scan_aggregation>()); break;
      case A::MAX: reqs[0].aggregations.push_back(cudf::make_max_aggregation<cudf::groupby_scan_aggregation>()); break;
      case A::COUNT_VALID: reqs[0].aggregations.push_back(cudf::make_count_aggregation<cudf::groupby_scan_aggregation>(cudf::null_policy::EXCLUDE)); break;
      case A::COUNT_ALL: reqs[0].aggregations.push_back(cudf::make_count_aggregation<cudf::groupby_scan_aggregation>(cudf::null_policy::INCLUDE)); break;
      default: throw std::invalid_argument("shim: scan kind");
    }
    auto res      = gb.scan(reqs);
    auto const& k = res.first->get_column(0);
    auto const& v = *res.second[0].results[0];
    emit(k.view(), out_keys, nullptr, nullptr);
    emit(v.view(), out_vals, out_vals_valid, out_vals_nulls_host);
    *out_rows_host      = k.size();
    *out_val_dtype_host = static_cast<int>(v.type().id());
    shim_sync();
  });
}

// groupby::shift of one value column (reference cpp/src/groupby/groupby.cu:306-346); fill = 8 value bytes + validity
int shim_groupby_shift(int key_dtype, const void* keys, const uint32_t* kvalid, int knulls, int val_dtype, const void* vals,
                       const uint32_t* vvalid, int vnulls, int n, int offset, unsigned long long fill_bits, int fill_valid,
                       void* out_keys, void* out_vals, uint32_t* out_vals_valid, int* out_vals_nulls_host, int* out_rows_host)
{
  return guarded([&] {
    auto kc = view(key_dtype, keys, kvalid, n, knulls);
    auto vc = view(val_dtype, vals, vvalid, n, vnulls);
    cudf::groupby::groupby gb(cudf::table_view{{kc}});
    std::unique_ptr<cudf::scalar> fill;
    switch (static_cast<cudf::type_id>(val_dtype)) {
      case cudf::type_id::INT32: {
        int32_t x;
        std::memcpy(&x, &fill_bits, 4);
        fill = std::make_unique<cudf::numeric_scalar<int32_t>>(x, fill_valid != 0);
        break;
      }
      case cudf::type_id::INT64: {
        int64_t x;
        std::memcpy(&x, &fill_bits, 8);
        fill = std::make_unique<cudf::numeric_scalar<int64_t>>(x, fill_valid != 0);
        break;
      }
      case cudf::type_id::FLOAT64: {
        double x;
        std::memcpy(&x, &fill_bits, 8);
        fill = std::make_unique<cudf::numeric_scalar<double>>(x, fill_valid != 0);
        break;
      }
      case cudf::type_id::FLOAT32: {
        float x;
        std::memcpy(&x, &fill_bits, 4);
        fill = std::make_unique<cudf::numeric_scalar<float>>(x, fill_valid != 0);
        break;
      }
      default: throw std::invalid_argument("shim: shift value dtype");
    }
    std::vector<cudf::size_type> offs{offset};
    std::vector<std::reference_wrapper<cudf::scalar const>> fills{*fill};
    auto res = gb.shift(cudf::table_view{{vc}}, offs, fills);
    emit(res.first->get_column(0).view(), out_keys, nullptr, nullptr);
    emit(res.second->get_column(0).view(), out_vals, out_vals_valid, out_vals_nulls_host);
    *out_rows_host = res.first->num_rows();
    shim_sync();
  });
}

// groupby::replace_nulls of one value column (reference cpp/src/groupby/groupby.cu:285-321, sort/group_replace_nulls.cu)
int shim_groupby_replace_nulls(int key_dtype, const void* keys, const uint32_t* kvalid, int knulls, int val_dtype, const void* vals,
                               const uint32_t* vvalid, int vnulls, int n, int following, void* out_keys, void* out_vals,
                               uint32_t* out_vals_valid, int* out_vals_nulls_host, int* out_rows_host)
{
  return guarded([&] {
    auto kc = view(key_dtype, keys, kvalid, n, knulls);
    auto vc = view(val_dtype, vals, vvalid, n, vnulls);
    cudf::groupby::groupby gb(cudf::table_view{{kc}});
    std::vector<cudf::replace_policy> pol{following ? cudf::replace_policy::FOLLOWING : cudf::replace_policy::PRECEDING};
    auto res = gb.replace_nulls(cudf::table_view{{vc}}, pol);
    emit(res.first->get_column(0).view(), out_keys, nullptr, nullptr);
    emit(res.second->get_column(0).view(), out_vals, out_vals_valid, out_vals_nulls_host);
    *out_rows_host = res.first->num_rows();
    shim_sync();
  });
}

// groupby::get_groups (reference cpp/src/groupby/groupby.cu:261-283); out_offsets_host has room for n + 1 entries
int shim_groupby_get_groups(int key_dtype, const void* keys, const uint32_t* kvalid, int knulls, int val_dtype, const void* vals,
                            const uint32_t* vvalid, int vnulls, int n, int null_include, void* out_keys, void* out_vals,
                            int32_t* out_offsets_host, int* out_groups_host, int* out_rows_host)
{
  return guarded([&] {
    auto kc = view(key_dtype, keys, kvalid, n, knulls);
    auto vc = view(val_dtype, vals, vvalid, n, vnulls);
    cudf::groupby::groupby gb(cudf::table_view{{kc}}, null_include ? cudf::null_policy::INCLUDE : cudf::null_policy::EXCLUDE);
    auto g = gb.get_groups(cudf::table_view{{vc}});
    emit(g.keys->get_column(0).view(), out_keys, nullptr, nullptr);
    emit(g.values->get_column(0).view(), out_vals, nullptr, nullptr);
    std::memcpy(out_offsets_host, g.offsets.data(), g.offsets.size() * sizeof(int32_t));
    *out_groups_host = static_cast<int>(g.offsets.size()) - 1;
    *out_rows_host   = g.keys->num_rows();
    shim_sync();
  });
}

// hash_join::{inner,left,full}_join_match_context: matches per left row (reference hash_join.hpp:259-340)
int shim_join_match_counts(int dtype, const void* left, const uint32_t* lvalid, int lnulls, int nl, const void* right,
                           const uint32_t* rvalid, int rnulls, int nr, int kind, int nulls_equal, int32_t* out_counts)
{
  return guarded([&] {
    auto lc = view(dtype, left, lvalid, nl, lnulls);
    auto rc = view(dtype, right, rvalid, nr, rnulls);
    cudf::hash_join hj(cudf::table_view{{rc}}, nulls_equal ? cudf::null_equality::EQUAL : cudf::null_equality::UNEQUAL);
    auto ctx = kind == 0 ? hj.inner_join_match_context(cudf::table_view{{lc}})
                         : kind == 1 ? hj.left_join_match_context(cudf::table_view{{lc}}) : hj.full_join_match_context(cudf::table_view{{lc}});
    d2d(out_counts, ctx._match_counts->data(), static_cast<std::size_t>(nl) * sizeof(int32_t));
    shim_sync();
  });
}

// cudf::full_join (left join + complement: reference cpp/src/join/join_utils.cu:86-157); out_* have room for cap pairs
int shim_full_join(int dtype, const void* left, const uint32_t* lvalid, int lnulls, int nl, const void* right, const uint32_t* rvalid,
                   int rnulls, int nr, int nulls_equal, long long cap, int32_t* out_l, int32_t* out_r, long long* out_pairs_host)
{
  return guarded([&] {
    auto lc  = view(dtype, left, lvalid, nl, lnulls);
    auto rc  = view(dtype, right, rvalid, nr, rnulls);
    auto res = cudf::full_join(cudf::table_view{{lc}}, cudf::table_view{{rc}},
                               nulls_equal ? cudf::null_equality::EQUAL : cudf::null_equality::UNEQUAL);
    auto const m = static_cast<long long>(res.first->size());
    *out_pairs_host = m;
    if (m > cap) throw std::runtime_error("shim: full_join output exceeds the caller's capacity");
    d2d(out_l, res.first->data(), static_cast<std::size_t>(m) * sizeof(int32_t));
    d2d(out_r, res.second->data(), static_cast<std::size_t>(m) * sizeof(int32_t));
    shim_sync();
  });
}

// cudf::hash_partition (partitioning.hpp:103-110) of `ncols` columns with `n` rows on the key columns key_idx_host[0 .. nkeys);
// which == 1: the keys-table overload (partitioning.hpp:138-145) with the selected columns as the key table.  Outputs: the
// regrouped columns (caller-allocated, n rows each), the offsets vector (out_offsets_host has room for max(parts, 0) + 2) and its
// length, the rows of the output table, its number of columns.  Exceptions come back as -1 with the type name first.
int shim_hash_partition(int ncols, const int* dtypes_host, const void* const* datas_host, const uint32_t* const* valids_host,
                        const int* nulls_host, int n, int nkeys, const int* key_idx_host, int parts, unsigned seed, int which,
                        const uint32_t* ext_key /* which == 2: a UINT32 key column of its own, hashed with HASH_IDENTITY */,
                        void* const* out_datas_host, uint32_t* const* out_valids_host, int* out_nulls_host, int* out_offsets_host,
                        int* out_noffsets_host, int* out_rows_host, int* out_cols_host)
{
  auto body = [&] {
    std::vector<cudf::column_view> cols;
    for (int i = 0; i < ncols; ++i) cols.push_back(view(dtypes_host[i], datas_host[i], valids_host ? valids_host[i] : nullptr, n, nulls_host ? nulls_host[i] : 0));
    cudf::table_view input{cols};
    std::vector<cudf::size_type> idx(key_idx_host, key_idx_host + nkeys);
    auto res = which == 2   ? cudf::hash_partition(input, cudf::table_view{{view(static_cast<int>(cudf::type_id::UINT32), ext_key, nullptr, n, 0)}}, parts,
                                                   cudf::hash_id::HASH_IDENTITY, seed)
               : which == 3 ? cudf::hash_partition(input, input.select(idx), parts, cudf::hash_id::HASH_IDENTITY, seed)  // (any numeric key table)
               : which == 1 ? cudf::hash_partition(input, input.select(idx), parts, cudf::hash_id::HASH_MURMUR3, seed)
                            : cudf::hash_partition(input, idx, parts, cudf::hash_id::HASH_MURMUR3, seed);
    *out_rows_host     = res.first->num_rows();
    *out_cols_host     = res.first->num_columns();
    *out_noffsets_host = static_cast<int>(res.second.size());
    for (std::size_t i = 0; i < res.second.size(); ++i) out_offsets_host[i] = res.second[i];
    for (int i = 0; i < res.first->num_columns(); ++i)
      emit(res.first->get_column(i).view(), out_datas_host[i], out_valids_host ? out_valids_host[i] : nullptr, out_nulls_host ? out_nulls_host + i : nullptr);
    shim_sync();
  };
  try {
    body();
    return 0;
  } catch (std::out_of_range const& e) {
    g_err = std::string{"std::out_of_range: "} + e.what();
  } catch (std::invalid_argument const& e) {
    g_err = std::string{"std::invalid_argument: "} + e.what();
  } catch (std::exception const& e) {
    g_err = e.what();
  }
  return -1;
}

// the keys-table overload with a key table of a DIFFERENT row count (hash_partition_test.cpp:62-71: std::invalid_argument)
int shim_hash_partition_key_rows(int dtype, const void* data, int n, int key_dtype, const void* keys, int nk, int parts)
{
  try {
    auto res = cudf::hash_partition(cudf::table_view{{view(dtype, data, nullptr, n, 0)}}, cudf::table_view{{view(key_dtype, keys, nullptr, nk, 0)}}, parts);
    shim_sync();
    return 0;
  } catch (std::invalid_argument const& e) {
    g_err = std::string{"std::invalid_argument: "} + e.what();
  } catch (std::exception const& e) {
    g_err = e.what();
  }
  return -1;
}

// cudf::partition by an INT32 map (partitioning.hpp:44-78)
int shim_partition(int dtype, const void* data, int n, const int32_t* map, int parts, void* out, int* out_offsets_host, int* out_noffsets_host,
                   int* out_rows_host)
{
  return guarded([&] {
    auto res = cudf::partition(cudf::table_view{{view(dtype, data, nullptr, n, 0)}}, view(static_cast<int>(cudf::type_id::INT32), map, nullptr, n, 0), parts);
    *out_rows_host     = res.first->num_rows();
    *out_noffsets_host = static_cast<int>(res.second.size());
    for (std::size_t i = 0; i < res.second.size(); ++i) out_offsets_host[i] = res.second[i];
    emit(res.first->get_column(0).view(), out, nullptr, nullptr);
    shim_sync();
  });
}

// cudf::partition by a map of ANY type (partitioning.cu:780-842: the integral types but bool; others throw cudf::logic_error)
int shim_partition_typed(int dtype, const void* data, int n, int map_dtype, const void* map, const uint32_t* map_valid, int map_nulls, int map_rows, int parts,
                         void* out, int* out_offsets_host, int* out_noffsets_host, int* out_rows_host)
{
  return guarded([&] {
    auto res = cudf::partition(cudf::table_view{{view(dtype, data, nullptr, n, 0)}}, view(map_dtype, map, map_valid, map_rows, map_nulls), parts);
    *out_rows_host     = res.first->num_rows();
    *out_noffsets_host = static_cast<int>(res.second.size());
    for (std::size_t i = 0; i < res.second.size(); ++i) out_offsets_host[i] = res.second[i];
    if (res.first->num_rows() > 0) emit(res.first->get_column(0).view(), out, nullptr, nullptr);
    shim_sync();
  });
}

// cudf::reduce with an initial value (reduction.hpp:124-130).  kind: aggregation::Kind; init_bits_host: the initial value's bytes
// (of init_dtype -- a type other than the column's must throw cudf::data_type_error); has_init 0: the overload is called with
// an empty optional.  Result: out_bits_host (8 bytes, the scalar's value in out_dtype) and out_valid_host.
int shim_reduce_init(int dtype, const void* data, const uint32_t* valid, int nulls, int n, int kind, int out_dtype, int has_init, int init_dtype,
                     unsigned long long init_bits_host, int init_valid, unsigned long long* out_bits_host, int* out_valid_host)
{
  auto body = [&] {
    auto c = view(dtype, data, valid, n, nulls);
    using A = cudf::aggregation;
    std::unique_ptr<cudf::reduce_aggregation> agg;
    switch (static_cast<A::Kind>(kind)) {
      case A::SUM: agg = cudf::make_sum_aggregation<cudf::reduce_aggregation>(); break;
      case A::PRODUCT: agg = cudf::make_product_aggregation<cudf::reduce_aggregation>(); break;
      case A::MIN: agg = cudf::make_min_aggregation<cudf::reduce_aggregation>(); break;
      case A::MAX: agg = cudf::make_max_aggregation<cudf::reduce_aggregation>(); break;
      case A::MEAN: agg = cudf::make_mean_aggregation<cudf::reduce_aggregation>(); break;
      case A::COUNT_VALID: agg = cudf::make_count_aggregation<cudf::reduce_aggregation>(cudf::null_policy::EXCLUDE); break;
      case A::COUNT_ALL: agg = cudf::make_count_aggregation<cudf::reduce_aggregation>(cudf::null_policy::INCLUDE); break;
      case A::ANY: agg = cudf::make_any_aggregation<cudf::reduce_aggregation>(); break;
      case A::ALL: agg = cudf::make_all_aggregation<cudf::reduce_aggregation>(); break;
      default: throw std::runtime_error("shim: reduce kind");
    }
    std::unique_ptr<cudf::scalar> init;
    auto mk = [&](auto tag) {
      using T = decltype(tag);
      T v;
      std::memcpy(&v, &init_bits_host, sizeof(T));
      init = std::make_unique<cudf::numeric_scalar<T>>(v, init_valid != 0);
    };
    if (has_init) {
      switch (static_cast<cudf::type_id>(init_dtype)) {
        case cudf::type_id::INT8: mk(int8_t{}); break;
        case cudf::type_id::BOOL8: mk(bool{}); break;
        case cudf::type_id::INT16: mk(int16_t{}); break;
        case cudf::type_id::INT32: mk(int32_t{}); break;
        case cudf::type_id::INT64: mk(int64_t{}); break;
        case cudf::type_id::UINT8: mk(uint8_t{}); break;
        case cudf::type_id::FLOAT32: mk(float{}); break;
        case cudf::type_id::FLOAT64: mk(double{}); break;
        default: throw std::runtime_error("shim: init dtype");
      }
    }
    std::optional<std::reference_wrapper<cudf::scalar const>> oi;
    if (init) oi = std::cref(*init);
    auto r          = cudf::reduce(c, *agg, cudf::data_type{static_cast<cudf::type_id>(out_dtype)}, oi);
    *out_valid_host = r->is_valid() ? 1 : 0;
    *out_bits_host  = 0;
    if (auto const* p = r->device_value_ptr()) {
      if (hipMemcpy(out_bits_host, p, cudf::size_of(r->type()), hipMemcpyDeviceToHost) != hipSuccess) throw std::runtime_error("shim: copy failed");
    }
    shim_sync();
  };
  try {
    body();
    return 0;
  } catch (cudf::data_type_error const& e) {
    g_err = std::string{"cudf::data_type_error: "} + e.what();
  } catch (std::invalid_argument const& e) {
    g_err = std::string{"std::invalid_argument: "} + e.what();
  } catch (cudf::logic_error const& e) {
    g_err = std::string{"cudf::logic_error: "} + e.what();
  } catch (std::exception const& e) {
    g_err = e.what();
  }
  return -1;
}

}  // extern "C"
