// cudf/aggregation.hpp -- aggregation descriptors and their factories
// (reference: cpp/include/cudf/aggregation.hpp:73-330).  The Kind enumerators keep the reference's
// order so integer values stay interchangeable; the hot path implements SUM, PRODUCT, MIN, MAX, COUNT_VALID,
// COUNT_ALL, MEAN, ANY and ALL (cudf::reduce), SUM_OF_SQUARES, M2, VARIANCE, STD, ARGMIN, ARGMAX and (sort path) NTH_ELEMENT; others throw
// cudf::logic_error where they are used.
#pragma once
#include <cudf/types.hpp>
#include <cudf/utilities/error.hpp>

#include <memory>

namespace cudf {

class aggregation {
 public:
  enum Kind {
    SUM, SUM_WITH_OVERFLOW, PRODUCT, MIN, MAX, COUNT_VALID, COUNT_ALL, ANY, ALL, SUM_OF_SQUARES, MEAN, M2, VARIANCE,
    STD, MEDIAN, QUANTILE, ARGMAX, ARGMIN, NUNIQUE, NTH_ELEMENT, ROW_NUMBER, EWMA, RANK, COLLECT_LIST, COLLECT_SET,
    LEAD, LAG, PTX, CUDA, HOST_UDF, MERGE_LISTS, MERGE_SETS, MERGE_M2, COVARIANCE, CORRELATION, TDIGEST,
    MERGE_TDIGEST, HISTOGRAM, MERGE_HISTOGRAM, BITWISE_AGG, TOP_K, INVALID
  };

  aggregation() : kind{Kind::INVALID} { CUDF_FAIL("No-parameter aggregation constructor should never be called"); }
  explicit aggregation(aggregation::Kind a) : kind{a} {}
  virtual ~aggregation() = default;
  Kind kind;
  [[nodiscard]] virtual bool is_equal(aggregation const& other) const { return kind == other.kind; }
  [[nodiscard]] virtual std::size_t do_hash() const { return static_cast<std::size_t>(kind); }
  [[nodiscard]] virtual std::unique_ptr<aggregation> clone() const { return std::make_unique<aggregation>(kind); }
};

class rolling_aggregation : public virtual aggregation {};
class groupby_aggregation : public virtual aggregation {};
class groupby_scan_aggregation : public virtual aggregation {};
class reduce_aggregation : public virtual aggregation {};
class scan_aggregation : public virtual aggregation {};
class segmented_reduce_aggregation : public virtual aggregation {};

namespace detail {
// one concrete type usable through every API-facing base (the reference derives one class per
// kind from the bases it is legal for: cpp/include/cudf/detail/aggregation/aggregation.hpp)
class simple_aggregation final : public groupby_aggregation,
                                 public groupby_scan_aggregation,
                                 public reduce_aggregation,
                                 public scan_aggregation,
                                 public segmented_reduce_aggregation,
                                 public rolling_aggregation {
 public:
  explicit simple_aggregation(aggregation::Kind k) : aggregation(k) {}
  [[nodiscard]] std::unique_ptr<aggregation> clone() const override
  {
    return std::unique_ptr<aggregation>(static_cast<groupby_aggregation*>(new simple_aggregation(kind)));
  }
};
// VARIANCE / STD carry the delta degrees of freedom (reference: detail/aggregation/aggregation.hpp
// std_var_aggregation, _ddof)
class std_var_aggregation final : public groupby_aggregation,
                                  public reduce_aggregation,
                                  public segmented_reduce_aggregation,
                                  public rolling_aggregation {
 public:
  std_var_aggregation(aggregation::Kind k, size_type ddof) : aggregation(k), _ddof{ddof} {}
  size_type _ddof;
  [[nodiscard]] bool is_equal(aggregation const& other) const override
  {
    auto const* o = dynamic_cast<std_var_aggregation const*>(&other);
    return o != nullptr && o->kind == kind && o->_ddof == _ddof;
  }
  [[nodiscard]] std::size_t do_hash() const override { return static_cast<std::size_t>(kind) * 31u + static_cast<std::size_t>(_ddof); }
  [[nodiscard]] std::unique_ptr<aggregation> clone() const override
  {
    return std::unique_ptr<aggregation>(static_cast<groupby_aggregation*>(new std_var_aggregation(kind, _ddof)));
  }
};
// NTH_ELEMENT carries the index and the null handling (reference: detail/aggregation/aggregation.hpp
// nth_element_aggregation, _n / _null_handling); sort-path only -- the reference's tests use it to force that path
class nth_element_aggregation final : public groupby_aggregation, public reduce_aggregation, public rolling_aggregation {
 public:
  nth_element_aggregation(size_type n, null_policy null_handling) : aggregation(aggregation::NTH_ELEMENT), _n{n}, _null_handling{null_handling} {}
  size_type _n;
  null_policy _null_handling;
  [[nodiscard]] bool is_equal(aggregation const& other) const override
  {
    auto const* o = dynamic_cast<nth_element_aggregation const*>(&other);
    return o != nullptr && o->_n == _n && o->_null_handling == _null_handling;
  }
  [[nodiscard]] std::size_t do_hash() const override { return static_cast<std::size_t>(kind) * 31u + static_cast<std::size_t>(_n); }
  [[nodiscard]] std::unique_ptr<aggregation> clone() const override
  {
    return std::unique_ptr<aggregation>(static_cast<groupby_aggregation*>(new nth_element_aggregation(_n, _null_handling)));
  }
};

template <typename Base>
std::unique_ptr<Base> make_std_var(aggregation::Kind k, size_type ddof)
{
  return std::unique_ptr<Base>(static_cast<Base*>(new std_var_aggregation(k, ddof)));
}
template <>
inline std::unique_ptr<aggregation> make_std_var<aggregation>(aggregation::Kind k, size_type ddof)
{
  return std::unique_ptr<aggregation>(static_cast<groupby_aggregation*>(new std_var_aggregation(k, ddof)));
}

template <typename Base>
std::unique_ptr<Base> make_simple(aggregation::Kind k)
{
  return std::unique_ptr<Base>(static_cast<Base*>(new simple_aggregation(k)));
}
template <>
inline std::unique_ptr<aggregation> make_simple<aggregation>(aggregation::Kind k)
{
  return std::unique_ptr<aggregation>(static_cast<groupby_aggregation*>(new simple_aggregation(k)));
}
}  // namespace detail

template <typename Base = aggregation>
std::unique_ptr<Base> make_sum_aggregation() { return detail::make_simple<Base>(aggregation::SUM); }
template <typename Base = aggregation>
std::unique_ptr<Base> make_product_aggregation() { return detail::make_simple<Base>(aggregation::PRODUCT); }
template <typename Base = aggregation>
std::unique_ptr<Base> make_min_aggregation() { return detail::make_simple<Base>(aggregation::MIN); }
template <typename Base = aggregation>
std::unique_ptr<Base> make_max_aggregation() { return detail::make_simple<Base>(aggregation::MAX); }
template <typename Base = aggregation>
std::unique_ptr<Base> make_count_aggregation(null_policy null_handling = null_policy::EXCLUDE)
{
  return detail::make_simple<Base>(null_handling == null_policy::INCLUDE ? aggregation::COUNT_ALL : aggregation::COUNT_VALID);
}
template <typename Base = aggregation>
std::unique_ptr<Base> make_mean_aggregation() { return detail::make_simple<Base>(aggregation::MEAN); }
// ANY / ALL reductions (aggregation.hpp:318-326 of the reference): cudf::reduce only, BOOL8 output
template <typename Base = aggregation>
std::unique_ptr<Base> make_any_aggregation() { return detail::make_simple<Base>(aggregation::ANY); }
template <typename Base = aggregation>
std::unique_ptr<Base> make_all_aggregation() { return detail::make_simple<Base>(aggregation::ALL); }
// include/cudf/aggregation.hpp:335-375 of the reference: SUM_OF_SQUARES, M2, VARIANCE(ddof = 1), STD(ddof = 1)
template <typename Base = aggregation>
std::unique_ptr<Base> make_sum_of_squares_aggregation() { return detail::make_simple<Base>(aggregation::SUM_OF_SQUARES); }
template <typename Base = aggregation>
std::unique_ptr<Base> make_m2_aggregation() { return detail::make_simple<Base>(aggregation::M2); }
template <typename Base = aggregation>
std::unique_ptr<Base> make_variance_aggregation(size_type ddof = 1) { return detail::make_std_var<Base>(aggregation::VARIANCE, ddof); }
template <typename Base = aggregation>
std::unique_ptr<Base> make_std_aggregation(size_type ddof = 1) { return detail::make_std_var<Base>(aggregation::STD, ddof); }
// NTH_ELEMENT: the n-th value of each group, negative n from the end (aggregation.hpp:448-470)
template <typename Base = aggregation>
std::unique_ptr<Base> make_nth_element_aggregation(size_type n, null_policy null_handling = null_policy::INCLUDE)
{
  return std::unique_ptr<Base>(static_cast<Base*>(new detail::nth_element_aggregation(n, null_handling)));
}
template <>
inline std::unique_ptr<aggregation> make_nth_element_aggregation<aggregation>(size_type n, null_policy null_handling)
{
  return std::unique_ptr<aggregation>(static_cast<groupby_aggregation*>(new detail::nth_element_aggregation(n, null_handling)));
}
// ARGMAX / ARGMIN: row index (size_type) of the group's maximum / minimum (aggregation.hpp:430-446)
template <typename Base = aggregation>
std::unique_ptr<Base> make_argmax_aggregation() { return detail::make_simple<Base>(aggregation::ARGMAX); }
template <typename Base = aggregation>
std::unique_ptr<Base> make_argmin_aggregation() { return detail::make_simple<Base>(aggregation::ARGMIN); }

}  // namespace cudf
