// cudf/interop.hpp -- Arrow C Device Data Interface in / out for the columns of the hot path
// (reference: cpp/include/cudf/interop.hpp:112-184,477-606,838-885; impl cpp/src/interop/
// to_arrow_schema.cpp, to_arrow_device.cu:480-560, from_arrow_device.cu:380-480).
//
// The column layout IS Arrow's (data buffer + LSB-first validity bitmap), so both directions are
// zero-copy: to_arrow_device moves (or views) the device buffers into an ArrowDeviceArray whose
// release callback frees them; from_arrow_device returns views over the producer's buffers.
// device_type is ARROW_DEVICE_ROCM (the reference writes ARROW_DEVICE_CUDA), sync_event points at a
// hipEvent_t recorded on the producing stream.  Supported here: the fixed-width numeric types of
// the hot path (INT8..UINT64, FLOAT32/64); BOOL8 (Arrow packs booleans into bits), strings, nested
// types, decimals, timestamps are outside this tier and throw cudf::data_type_error.
#pragma once
#include <cudf/column/column.hpp>
#include <cudf/column/column_view.hpp>
#include <cudf/table/table.hpp>
#include <cudf/table/table_view.hpp>
#include <cudf/utilities/default_stream.hpp>
#include <cudf/utilities/memory_resource.hpp>

#include <cstdint>
#include <memory>
#include <optional>
#include <span>
#include <string>
#include <vector>

// ---- the Arrow C data / device data interface (ABI-stable struct definitions from the Arrow
// specification; guarded by the specification's own macros so Arrow's headers can coexist)
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4
extern "C" {
struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};
}
#endif
#ifndef ARROW_C_DEVICE_DATA_INTERFACE
#define ARROW_C_DEVICE_DATA_INTERFACE
typedef int32_t ArrowDeviceType;
#define ARROW_DEVICE_CPU 1
#define ARROW_DEVICE_CUDA 2
#define ARROW_DEVICE_CUDA_HOST 3
#define ARROW_DEVICE_ROCM 10
#define ARROW_DEVICE_ROCM_HOST 11
#define ARROW_DEVICE_CUDA_MANAGED 13
extern "C" {
struct ArrowDeviceArray {
  struct ArrowArray array;
  int64_t device_id;
  ArrowDeviceType device_type;
  void* sync_event;
  int64_t reserved[3];
};
}
#endif

namespace cudf {

// names for the schema (interop.hpp:112-129)
struct column_metadata {
  std::string name;
  std::string timezone;
  std::optional<int32_t> precision;
  std::vector<column_metadata> children_meta;
  column_metadata(std::string _name) : name(std::move(_name)) {}
  column_metadata() = default;
};

using unique_schema_t       = std::unique_ptr<ArrowSchema, void (*)(ArrowSchema*)>;
using unique_device_array_t = std::unique_ptr<ArrowDeviceArray, void (*)(ArrowDeviceArray*)>;
using owned_columns_t       = std::vector<std::unique_ptr<cudf::column>>;

template <typename ViewType>
struct custom_view_deleter {
  explicit custom_view_deleter(owned_columns_t&& owned) : owned_mem_{std::move(owned)} {}
  void operator()(ViewType* ptr) const { delete ptr; }
  owned_columns_t owned_mem_;  // columns materialised during the import (none for the types supported here)
};
using unique_table_view_t  = std::unique_ptr<cudf::table_view, custom_view_deleter<cudf::table_view>>;
using unique_column_view_t = std::unique_ptr<cudf::column_view, custom_view_deleter<cudf::column_view>>;

// "+s" struct schema with one child per column; a child is flagged nullable when its column is
unique_schema_t to_arrow_schema(table_view const& input, std::span<column_metadata const> metadata);

// Owning exports: the table's / column's device buffers move into the result and are freed by its release
// callback.  A table becomes a struct array with one child per column.
unique_device_array_t to_arrow_device(table&& tbl, rmm::cuda_stream_view stream = cudf::get_default_stream(),
                                      rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());
unique_device_array_t to_arrow_device(column&& col, rmm::cuda_stream_view stream = cudf::get_default_stream(),
                                      rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());
// Non-owning exports: the result views the input's buffers; the caller keeps them alive (interop.hpp:565-606)
unique_device_array_t to_arrow_device(table_view const& tbl, rmm::cuda_stream_view stream = cudf::get_default_stream(),
                                      rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());
unique_device_array_t to_arrow_device(column_view const& col, rmm::cuda_stream_view stream = cudf::get_default_stream(),
                                      rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

// Zero-copy imports: views over the producer's device buffers, valid while `input` is not released.
// `stream` waits on input->sync_event when there is one.  throws std::invalid_argument for NULL inputs or
// memory that is not device accessible, cudf::data_type_error for a non-struct schema (table form) or
// an unsupported element type, std::overflow_error for more than 2^31-1 rows.
unique_table_view_t from_arrow_device(ArrowSchema const* schema, ArrowDeviceArray const* input,
                                      rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                                      rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());
unique_column_view_t from_arrow_device_column(ArrowSchema const* schema, ArrowDeviceArray const* input,
                                              rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                                              rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref());

}  // namespace cudf
