// Arrow C Device Data Interface for fixed-width columns (see include/cudf/interop.hpp).
// reference: cpp/src/interop/to_arrow_schema.cpp, to_arrow_device.cu:480-560, from_arrow_device.cu:380-480.
#include "common.hpp"

#include <cudf/interop.hpp>
#include <cudf/null_mask.hpp>

#include <hip/hip_runtime_api.h>

#include <cstring>
#include <limits>

namespace cudf {

namespace {

char const* arrow_format(data_type t)
{
  switch (t.id()) {
    case type_id::INT8: return "c";
    case type_id::UINT8: return "C";
    case type_id::INT16: return "s";
    case type_id::UINT16: return "S";
    case type_id::INT32: return "i";
    case type_id::UINT32: return "I";
    case type_id::INT64: return "l";
    case type_id::UINT64: return "L";
    case type_id::FLOAT32: return "f";
    case type_id::FLOAT64: return "g";
    default: throw cudf::data_type_error{"Arrow interop on this path covers the fixed-width numeric types only"};
  }
}

data_type cudf_type(char const* f)
{
  CUDF_EXPECTS(f != nullptr && f[0] != 0, "ArrowSchema has no format string", std::invalid_argument);
  if (f[1] == 0) {
    switch (f[0]) {
      case 'c': return data_type{type_id::INT8};
      case 'C': return data_type{type_id::UINT8};
      case 's': return data_type{type_id::INT16};
      case 'S': return data_type{type_id::UINT16};
      case 'i': return data_type{type_id::INT32};
      case 'I': return data_type{type_id::UINT32};
      case 'l': return data_type{type_id::INT64};
      case 'L': return data_type{type_id::UINT64};
      case 'f': return data_type{type_id::FLOAT32};
      case 'g': return data_type{type_id::FLOAT64};
      default: break;
    }
  }
  if (std::strcmp(f, "+s") == 0) return data_type{type_id::STRUCT};
  throw cudf::data_type_error{std::string{"Arrow format '"} + f + "' is outside the fixed-width numeric types of this path"};
}

// ------------------------------------------------------------------ schema
struct schema_private {
  std::string name;
  std::vector<ArrowSchema> child_storage;
  std::vector<ArrowSchema*> child_ptrs;
};

void release_schema(ArrowSchema* s)
{
  if (s == nullptr || s->release == nullptr) return;
  for (int64_t i = 0; i < s->n_children; ++i)
    if (s->children[i]->release) s->children[i]->release(s->children[i]);
  delete static_cast<schema_private*>(s->private_data);
  s->release = nullptr;
}

void fill_schema(ArrowSchema* s, char const* format, std::string name, int64_t flags, int64_t nchildren)
{
  auto* p = new schema_private;
  p->name = std::move(name);
  p->child_storage.resize(nchildren);
  for (auto& c : p->child_storage) p->child_ptrs.push_back(&c);
  std::memset(s, 0, sizeof(*s));
  s->format       = format;
  s->name         = p->name.c_str();
  s->flags        = flags;
  s->n_children   = nchildren;
  s->children     = nchildren ? p->child_ptrs.data() : nullptr;
  s->release      = release_schema;
  s->private_data = p;
}

// ------------------------------------------------------------------ arrays
struct array_private {
  std::vector<void const*> buffers;
  std::vector<ArrowArray> child_storage;
  std::vector<ArrowArray*> child_ptrs;
  std::unique_ptr<rmm::device_buffer> data, mask;  // owning exports
};

void release_array(ArrowArray* a)
{
  if (a == nullptr || a->release == nullptr) return;
  for (int64_t i = 0; i < a->n_children; ++i)
    if (a->children[i]->release) a->children[i]->release(a->children[i]);
  delete static_cast<array_private*>(a->private_data);
  a->release = nullptr;
}

// one fixed-width array: buffers = {validity or NULL, data}
void fill_leaf(ArrowArray* a, size_type length, size_type null_count, int64_t offset, void const* mask, void const* data,
               std::unique_ptr<rmm::device_buffer> own_data, std::unique_ptr<rmm::device_buffer> own_mask)
{
  auto* p     = new array_private;
  p->buffers  = {mask, data};
  p->data     = std::move(own_data);
  p->mask     = std::move(own_mask);
  std::memset(a, 0, sizeof(*a));
  a->length       = length;
  a->null_count   = null_count;
  a->offset       = offset;
  a->n_buffers    = 2;
  a->buffers      = p->buffers.data();
  a->release      = release_array;
  a->private_data = p;
}

void fill_struct(ArrowArray* a, size_type length, int64_t nchildren)
{
  auto* p    = new array_private;
  p->buffers = {nullptr};
  p->child_storage.resize(nchildren);
  for (auto& c : p->child_storage) {
    std::memset(&c, 0, sizeof(c));
    p->child_ptrs.push_back(&c);
  }
  std::memset(a, 0, sizeof(*a));
  a->length       = length;
  a->n_buffers    = 1;
  a->buffers      = p->buffers.data();
  a->n_children   = nchildren;
  a->children     = nchildren ? p->child_ptrs.data() : nullptr;
  a->release      = release_array;
  a->private_data = p;
}

struct device_private {
  hipEvent_t event{};
};

void delete_device_array(ArrowDeviceArray* d)
{
  if (d == nullptr) return;
  if (d->array.release) d->array.release(&d->array);
  if (d->sync_event) {
    auto* ev = static_cast<hipEvent_t*>(d->sync_event);
    (void)hipEventDestroy(*ev);
    delete ev;
  }
  delete d;
}

unique_device_array_t finish(std::unique_ptr<ArrowDeviceArray> d, rmm::cuda_stream_view stream)
{
  int dev = 0;
  CUDF_CUDA_TRY(hipGetDevice(&dev));
  d->device_id   = dev;
  d->device_type = ARROW_DEVICE_ROCM;
  auto* ev       = new hipEvent_t;
  CUDF_CUDA_TRY(hipEventCreateWithFlags(ev, hipEventDisableTiming));
  CUDF_CUDA_TRY(hipEventRecord(*ev, stream.value()));  // consumers order themselves after the producing stream
  d->sync_event = ev;
  return unique_device_array_t{d.release(), delete_device_array};
}

void export_owned(ArrowArray* a, column&& col)
{
  (void)arrow_format(col.type());  // type check
  auto const n     = col.size();
  auto const nulls = col.null_count();
  auto contents    = col.release();
  void const* data = contents.data->data();
  void const* mask = (nulls > 0 && contents.null_mask && contents.null_mask->size() > 0) ? contents.null_mask->data() : nullptr;
  fill_leaf(a, n, nulls, 0, mask, data, std::move(contents.data), std::move(contents.null_mask));
}

void export_view(ArrowArray* a, column_view const& col)
{
  (void)arrow_format(col.type());
  fill_leaf(a, col.size(), col.null_count(), col.offset(), col.has_nulls() ? col.null_mask() : nullptr, col.head<void>(), nullptr,
            nullptr);
}

// ------------------------------------------------------------------ import
void check_device(ArrowDeviceArray const* input)
{
  CUDF_EXPECTS(input->device_type == ARROW_DEVICE_ROCM || input->device_type == ARROW_DEVICE_ROCM_HOST,
               "ArrowDeviceArray memory must be accessible to the ROCm device", std::invalid_argument);
}

column_view import_leaf(ArrowSchema const* schema, ArrowArray const* a, rmm::cuda_stream_view stream)
{
  auto const type = cudf_type(schema->format);
  CUDF_EXPECTS(type.id() != type_id::STRUCT, "nested types are outside this path", cudf::data_type_error);
  CUDF_EXPECTS(a->length <= static_cast<int64_t>(std::numeric_limits<size_type>::max()),
               "Number of rows exceeds cuDF's maximum supported row count (cudf::size_type).", std::overflow_error);
  CUDF_EXPECTS(a->n_buffers == 2, "fixed-width Arrow arrays carry two buffers", std::invalid_argument);
  auto const n      = static_cast<size_type>(a->length);
  auto const offset = static_cast<size_type>(a->offset);
  auto const* mask  = static_cast<bitmask_type const*>(a->buffers[0]);
  size_type nulls   = 0;
  if (mask != nullptr) {
    // Arrow allows null_count == -1 (not computed): count the unset bits of [offset, offset + n)
    nulls = a->null_count >= 0 ? static_cast<size_type>(a->null_count) : cudf::null_count(mask, offset, offset + n, stream);
  }
  return column_view{type, n, a->buffers[1], mask, nulls, offset};
}

}  // namespace

unique_schema_t to_arrow_schema(table_view const& input, std::span<column_metadata const> metadata)
{
  CUDF_EXPECTS(metadata.size() == static_cast<std::size_t>(input.num_columns()),
               "columns' metadata should be equal to the number of columns in table", std::invalid_argument);
  auto s = std::make_unique<ArrowSchema>();
  fill_schema(s.get(), "+s", "", 0, input.num_columns());
  for (size_type i = 0; i < input.num_columns(); ++i) {
    auto const& c = input.column(i);
    fill_schema(s->children[i], arrow_format(c.type()), metadata[i].name, c.has_nulls() ? ARROW_FLAG_NULLABLE : 0, 0);
  }
  return unique_schema_t{s.release(), [](ArrowSchema* p) {
                           if (p == nullptr) return;
                           if (p->release) p->release(p);
                           delete p;
                         }};
}

unique_device_array_t to_arrow_device(table&& tbl, rmm::cuda_stream_view stream, rmm::device_async_resource_ref)
{
  auto d     = std::make_unique<ArrowDeviceArray>();
  std::memset(d.get(), 0, sizeof(*d));
  auto const rows = tbl.num_rows();
  auto cols       = tbl.release();
  fill_struct(&d->array, rows, static_cast<int64_t>(cols.size()));
  for (std::size_t i = 0; i < cols.size(); ++i) export_owned(d->array.children[i], std::move(*cols[i]));
  return finish(std::move(d), stream);
}

unique_device_array_t to_arrow_device(column&& col, rmm::cuda_stream_view stream, rmm::device_async_resource_ref)
{
  auto d = std::make_unique<ArrowDeviceArray>();
  std::memset(d.get(), 0, sizeof(*d));
  export_owned(&d->array, std::move(col));
  return finish(std::move(d), stream);
}

unique_device_array_t to_arrow_device(table_view const& tbl, rmm::cuda_stream_view stream, rmm::device_async_resource_ref)
{
  auto d = std::make_unique<ArrowDeviceArray>();
  std::memset(d.get(), 0, sizeof(*d));
  fill_struct(&d->array, tbl.num_rows(), tbl.num_columns());
  for (size_type i = 0; i < tbl.num_columns(); ++i) export_view(d->array.children[i], tbl.column(i));
  return finish(std::move(d), stream);
}

unique_device_array_t to_arrow_device(column_view const& col, rmm::cuda_stream_view stream, rmm::device_async_resource_ref)
{
  auto d = std::make_unique<ArrowDeviceArray>();
  std::memset(d.get(), 0, sizeof(*d));
  export_view(&d->array, col);
  return finish(std::move(d), stream);
}

unique_table_view_t from_arrow_device(ArrowSchema const* schema, ArrowDeviceArray const* input, rmm::cuda_stream_view stream,
                                      rmm::device_async_resource_ref)
{
  CUDF_EXPECTS(schema != nullptr && input != nullptr, "input ArrowSchema and ArrowDeviceArray must not be NULL",
               std::invalid_argument);
  check_device(input);
  if (input->sync_event != nullptr)
    CUDF_CUDA_TRY(hipStreamWaitEvent(stream.value(), *static_cast<hipEvent_t*>(input->sync_event), 0));
  CUDF_EXPECTS(cudf_type(schema->format).id() == type_id::STRUCT, "Must pass a struct to `from_arrow_device`",
               cudf::data_type_error);
  CUDF_EXPECTS(schema->n_children == input->array.n_children, "schema and array disagree on the number of children",
               std::invalid_argument);
  std::vector<column_view> cols;
  for (int64_t i = 0; i < input->array.n_children; ++i) {
    auto v = import_leaf(schema->children[i], input->array.children[i], stream);
    // rows of the parent struct: its offset / length select a window of every child
    if (input->array.offset != 0 || input->array.length != input->array.children[i]->length)
      v = column_view{v.type(), static_cast<size_type>(input->array.length), v.head<void>(), v.null_mask(),
                      v.nullable() ? cudf::null_count(v.null_mask(), v.offset() + static_cast<size_type>(input->array.offset),
                                                      v.offset() + static_cast<size_type>(input->array.offset + input->array.length), stream)
                                   : 0,
                      v.offset() + static_cast<size_type>(input->array.offset)};
    cols.push_back(v);
  }
  return unique_table_view_t{new table_view{cols}, custom_view_deleter<table_view>{owned_columns_t{}}};
}

unique_column_view_t from_arrow_device_column(ArrowSchema const* schema, ArrowDeviceArray const* input,
                                              rmm::cuda_stream_view stream, rmm::device_async_resource_ref)
{
  CUDF_EXPECTS(schema != nullptr && input != nullptr, "input ArrowSchema and ArrowDeviceArray must not be NULL",
               std::invalid_argument);
  check_device(input);
  if (input->sync_event != nullptr)
    CUDF_CUDA_TRY(hipStreamWaitEvent(stream.value(), *static_cast<hipEvent_t*>(input->sync_event), 0));
  auto v = import_leaf(schema, &input->array, stream);
  return unique_column_view_t{new column_view{v}, custom_view_deleter<column_view>{owned_columns_t{}}};
}

}  // namespace cudf
