// cudf/types.hpp -- core enums and aliases of the cudf API surface kept by this drop-in
// (reference: cpp/include/cudf/types.hpp:70-216,278-341).  Only the fixed-width subset that the
// hot path supports is functional; other type_ids exist so values stay ABI-compatible.
#pragma once
#include <cstddef>
#include <cstdint>

namespace cudf {

using size_type         = int32_t;
using bitmask_type      = uint32_t;
using valid_type        = uint8_t;
using thread_index_type = int64_t;

enum class order : bool { ASCENDING, DESCENDING };
enum class null_policy : bool { EXCLUDE, INCLUDE };
enum class nan_policy : bool { NAN_IS_NULL, NAN_IS_VALID };
enum class nan_equality { ALL_EQUAL, UNEQUAL };
enum class null_equality : bool { EQUAL, UNEQUAL };
enum class null_order : bool { AFTER, BEFORE };
enum class sorted : bool { NO, YES };
enum class mask_state : int32_t { UNALLOCATED, UNINITIALIZED, ALL_VALID, ALL_NULL };
enum class scan_type : bool { INCLUSIVE, EXCLUSIVE };

struct order_info {
  sorted is_sorted;
  order ordering;
  null_order null_ordering;
};

enum class type_id : int32_t {
  EMPTY,
  INT8,
  INT16,
  INT32,
  INT64,
  UINT8,
  UINT16,
  UINT32,
  UINT64,
  FLOAT32,
  FLOAT64,
  BOOL8,
  TIMESTAMP_DAYS,
  TIMESTAMP_SECONDS,
  TIMESTAMP_MILLISECONDS,
  TIMESTAMP_MICROSECONDS,
  TIMESTAMP_NANOSECONDS,
  DURATION_DAYS,
  DURATION_SECONDS,
  DURATION_MILLISECONDS,
  DURATION_MICROSECONDS,
  DURATION_NANOSECONDS,
  DICTIONARY32,
  STRING,
  LIST,
  DECIMAL32,
  DECIMAL64,
  DECIMAL128,
  STRUCT,
  NUM_TYPE_IDS
};

class data_type {
 public:
  constexpr data_type() = default;
  explicit constexpr data_type(type_id id) : _id{id} {}
  explicit constexpr data_type(type_id id, int32_t scale) : _id{id}, _scale{scale} {}
  [[nodiscard]] constexpr type_id id() const noexcept { return _id; }
  [[nodiscard]] constexpr int32_t scale() const noexcept { return _scale; }

 private:
  type_id _id{type_id::EMPTY};
  int32_t _scale{};
};
constexpr bool operator==(data_type const& a, data_type const& b) { return a.id() == b.id() && a.scale() == b.scale(); }
constexpr bool operator!=(data_type const& a, data_type const& b) { return !(a == b); }

// size in bytes of one element of a fixed-width type, 0 otherwise
constexpr std::size_t size_of(data_type t)
{
  switch (t.id()) {
    case type_id::INT8:
    case type_id::UINT8:
    case type_id::BOOL8: return 1;
    case type_id::INT16:
    case type_id::UINT16: return 2;
    case type_id::INT32:
    case type_id::UINT32:
    case type_id::FLOAT32:
    case type_id::TIMESTAMP_DAYS:
    case type_id::DURATION_DAYS:
    case type_id::DECIMAL32: return 4;
    case type_id::INT64:
    case type_id::UINT64:
    case type_id::FLOAT64:
    case type_id::TIMESTAMP_SECONDS:
    case type_id::TIMESTAMP_MILLISECONDS:
    case type_id::TIMESTAMP_MICROSECONDS:
    case type_id::TIMESTAMP_NANOSECONDS:
    case type_id::DURATION_SECONDS:
    case type_id::DURATION_MILLISECONDS:
    case type_id::DURATION_MICROSECONDS:
    case type_id::DURATION_NANOSECONDS:
    case type_id::DECIMAL64: return 8;
    default: return 0;
  }
}
constexpr bool is_fixed_width(data_type t) { return size_of(t) != 0; }
constexpr bool is_floating_point(data_type t) { return t.id() == type_id::FLOAT32 || t.id() == type_id::FLOAT64; }
// (the run-time forms of cudf/utilities/traits.hpp:173-219: numeric = integers, floating point, bool; index type = integral, not bool)
constexpr bool is_index_type(data_type t)
{
  return static_cast<int>(t.id()) >= static_cast<int>(type_id::INT8) && static_cast<int>(t.id()) <= static_cast<int>(type_id::UINT64);
}
constexpr bool is_numeric(data_type t) { return is_index_type(t) || is_floating_point(t) || t.id() == type_id::BOOL8; }
constexpr bool is_nested(data_type t) { return t.id() == type_id::LIST || t.id() == type_id::STRUCT; }

template <typename T>
constexpr type_id type_to_id();
#define CUDF_AMD_TYPE_MAP(T, ID) \
  template <>                    \
  constexpr type_id type_to_id<T>() { return type_id::ID; }
CUDF_AMD_TYPE_MAP(int8_t, INT8)
CUDF_AMD_TYPE_MAP(int16_t, INT16)
CUDF_AMD_TYPE_MAP(int32_t, INT32)
CUDF_AMD_TYPE_MAP(int64_t, INT64)
CUDF_AMD_TYPE_MAP(uint8_t, UINT8)
CUDF_AMD_TYPE_MAP(uint16_t, UINT16)
CUDF_AMD_TYPE_MAP(uint32_t, UINT32)
CUDF_AMD_TYPE_MAP(uint64_t, UINT64)
CUDF_AMD_TYPE_MAP(float, FLOAT32)
CUDF_AMD_TYPE_MAP(double, FLOAT64)
CUDF_AMD_TYPE_MAP(bool, BOOL8)
#undef CUDF_AMD_TYPE_MAP

}  // namespace cudf
