// cudf::hash_join and the free-function joins over the C ABI.
// reference: cpp/src/join/hash_join/hash_join.cu:32-198, cpp/src/join/join.cu:27-124,
// cpp/src/join/join_utils.cu:45-157.
#include "common.hpp"

#include <cudf/join/hash_join.hpp>
#include <cudf/join/join.hpp>

#include <vector>

namespace cudf {
namespace detail {

namespace {
using map_ptr = std::unique_ptr<rmm::device_uvector<size_type>>;

join_result empty_result(rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  return {std::make_unique<rmm::device_uvector<size_type>>(0, stream, mr),
          std::make_unique<rmm::device_uvector<size_type>>(0, stream, mr)};
}

// row indices of the null rows of a nullable column (rare path: host round trip)
std::vector<size_type> null_rows(column_view const& c, rmm::cuda_stream_view stream)
{
  std::vector<size_type> out;
  if (!c.has_nulls()) return out;
  auto const words = num_bitmask_words(c.offset() + c.size());
  std::vector<bitmask_type> h(words);
  CUDF_CUDA_TRY(hipMemcpyAsync(h.data(), c.null_mask(), words * sizeof(bitmask_type), hipMemcpyDeviceToHost, stream.value()));
  stream.synchronize();
  for (size_type i = 0; i < c.size(); ++i) {
    auto const s = c.offset() + i;
    if (!((h[s / 32] >> (s % 32)) & 1u)) out.push_back(i);
  }
  return out;
}
}  // namespace

class hash_join_impl {
 public:
  hash_join_impl(table_view const& build, nullable_join has_nulls, null_equality compare_nulls, double load_factor,
                 rmm::cuda_stream_view stream)
    : _build{build}, _has_nulls{has_nulls == nullable_join::YES}, _nulls_equal{compare_nulls == null_equality::EQUAL},
      _load_factor{load_factor}
  {
    CUDF_EXPECTS(0 != build.num_columns(), "Hash join build table is empty", std::invalid_argument);
    CUDF_EXPECTS(load_factor > 0 && load_factor <= 1,
                 "Invalid load factor: must be greater than 0 and less than or equal to 1.", std::invalid_argument);
    CUDF_EXPECTS(build.num_columns() == 1, "multi-column join keys are not supported on this path yet");
    auto const& key = build.column(0);
    _key_size       = static_cast<int>(size_of(key.type()));
    CUDF_EXPECTS(_key_size == 4 || _key_size == 8, "join key must be a 4- or 8-byte fixed-width column", cudf::data_type_error);
    if (build.num_rows() == 0) return;
    _table_bytes = gx_join_table_bytes(_key_size, key.size(), load_factor);
    _table       = rmm::device_buffer{_table_bytes, stream};
    rmm::device_buffer holder;
    auto const* mask = key.has_nulls() ? rebased_mask(key, holder, stream) : nullptr;
    gx_check(gx_join_build(_key_size, row0(key), mask, key.size(), _table.data(), _table_bytes, load_factor, gxs(stream)),
             "hash_join build");
    _build_nulls = null_rows(key, stream);
    stream.synchronize();
  }

  void check_probe(table_view const& probe) const
  {
    CUDF_EXPECTS(probe.num_columns() == _build.num_columns(), "Mismatch in number of columns to be joined on",
                 std::invalid_argument);
    CUDF_EXPECTS(probe.column(0).type() == _build.column(0).type(), "Mismatch in joining column data types",
                 cudf::data_type_error);
    CUDF_EXPECTS(_has_nulls || !cudf::has_nulls(probe), "Probe table has nulls while build table was not hashed with null check.",
                 std::invalid_argument);
  }

  // pairs from the kernel probe (+ null x null cross product when nulls compare equal)
  join_result probe_join(table_view const& probe, bool left_outer, std::optional<std::size_t> output_size,
                         rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr) const
  {
    check_probe(probe);
    auto const& pk = probe.column(0);
    rmm::device_buffer holder;
    auto const* pmask = pk.has_nulls() ? rebased_mask(pk, holder, stream) : nullptr;
    std::vector<size_type> pnulls = _nulls_equal ? null_rows(pk, stream) : std::vector<size_type>{};
    std::size_t const cross = _nulls_equal ? pnulls.size() * _build_nulls.size() : 0;
    // with nulls EQUAL a null probe row that has a null partner must not also emit (i, NoMatch)
    bool const suppress_null_nomatch = left_outer && cross > 0;

    std::size_t capacity = output_size.value_or(static_cast<std::size_t>(pk.size())) + cross;
    rmm::device_buffer cursor{sizeof(int64_t), stream};
    map_ptr l, r;
    int64_t total = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
      l = std::make_unique<rmm::device_uvector<size_type>>(capacity, stream, mr);
      r = std::make_unique<rmm::device_uvector<size_type>>(capacity, stream, mr);
      CUDF_CUDA_TRY(hipMemsetAsync(cursor.data(), 0, sizeof(int64_t), stream.value()));
      gx_check(gx_join_probe(_key_size, row0(pk), pmask, pk.size(), _table.data(), _table_bytes, left_outer ? 1 : 0,
                             l->data(), r->data(), static_cast<int64_t>(capacity), static_cast<int64_t*>(cursor.data()),
                             gxs(stream)),
               "hash_join probe");
      total = read_i64(static_cast<int64_t const*>(cursor.data()), stream);
      if (static_cast<std::size_t>(total) + cross <= capacity) break;
      capacity = static_cast<std::size_t>(total) + cross;  // duplicate build keys: exact size now known
    }
    std::size_t n = static_cast<std::size_t>(total);
    if (cross > 0) {
      std::vector<size_type> hl, hr;
      hl.reserve(cross);
      hr.reserve(cross);
      for (auto p : pnulls)
        for (auto b : _build_nulls) {
          hl.push_back(p);
          hr.push_back(b);
        }
      CUDF_CUDA_TRY(hipMemcpyAsync(l->data() + n, hl.data(), cross * sizeof(size_type), hipMemcpyHostToDevice, stream.value()));
      CUDF_CUDA_TRY(hipMemcpyAsync(r->data() + n, hr.data(), cross * sizeof(size_type), hipMemcpyHostToDevice, stream.value()));
      stream.synchronize();
      n += cross;
    }
    (void)suppress_null_nomatch;  // the kernel skips null probe rows only for inner joins; see left_join below
    l->shrink(n);
    r->shrink(n);
    return {std::move(l), std::move(r)};
  }

  join_result inner_join(table_view const& probe, std::optional<std::size_t> output_size, rmm::cuda_stream_view stream,
                         rmm::device_async_resource_ref mr) const
  {
    if (probe.num_rows() == 0 || _build.num_rows() == 0) {  // trivial joins (hash_join.cu:32-45)
      check_probe(probe);
      return empty_result(stream, mr);
    }
    return probe_join(probe, false, output_size, stream, mr);
  }

  join_result left_join(table_view const& probe, std::optional<std::size_t> output_size, rmm::cuda_stream_view stream,
                        rmm::device_async_resource_ref mr) const
  {
    check_probe(probe);
    auto const n = probe.num_rows();
    if (n == 0) return empty_result(stream, mr);
    if (_build.num_rows() == 0) {  // (iota, JoinNoMatch...) -- join_utils.cu:45-60
      auto l = std::make_unique<rmm::device_uvector<size_type>>(n, stream, mr);
      auto r = std::make_unique<rmm::device_uvector<size_type>>(n, stream, mr);
      gx_check(gx_sequence_i32(l->data(), n, 0, gxs(stream)), "sequence");
      std::vector<size_type> nm(n, JoinNoMatch);
      CUDF_CUDA_TRY(hipMemcpyAsync(r->data(), nm.data(), n * sizeof(size_type), hipMemcpyHostToDevice, stream.value()));
      stream.synchronize();
      return {std::move(l), std::move(r)};
    }
    CUDF_EXPECTS(!(_nulls_equal && probe.column(0).has_nulls() && !_build_nulls.empty()),
                 "left/full join with null keys on both sides under null_equality::EQUAL is not supported on this path yet");
    return probe_join(probe, true, output_size, stream, mr);
  }

  join_result full_join(table_view const& probe, std::optional<std::size_t> output_size, rmm::cuda_stream_view stream,
                        rmm::device_async_resource_ref mr) const
  {
    auto [l, r]   = left_join(probe, output_size, stream, cudf::get_current_device_resource_ref());
    auto const nb = _build.num_rows();
    if (nb == 0) return {std::move(l), std::move(r)};
    std::size_t const n0 = l->size();
    auto ol = std::make_unique<rmm::device_uvector<size_type>>(n0 + nb, stream, mr);
    auto orr = std::make_unique<rmm::device_uvector<size_type>>(n0 + nb, stream, mr);
    if (n0) {
      CUDF_CUDA_TRY(hipMemcpyAsync(ol->data(), l->data(), n0 * sizeof(size_type), hipMemcpyDeviceToDevice, stream.value()));
      CUDF_CUDA_TRY(hipMemcpyAsync(orr->data(), r->data(), n0 * sizeof(size_type), hipMemcpyDeviceToDevice, stream.value()));
    }
    rmm::device_buffer cursor{sizeof(int64_t), stream};
    int64_t const start = static_cast<int64_t>(n0);
    CUDF_CUDA_TRY(hipMemcpyAsync(cursor.data(), &start, sizeof(int64_t), hipMemcpyHostToDevice, stream.value()));
    stream.synchronize();
    run_with_scratch(
      [&](void* t, std::size_t* b) {
        return gx_join_complement(r->data(), static_cast<int64_t>(n0), nb, ol->data(), orr->data(),
                                  static_cast<int64_t>(n0 + nb), static_cast<int64_t*>(cursor.data()), t, b, gxs(stream));
      },
      "full_join complement", stream);
    auto const total = read_i64(static_cast<int64_t const*>(cursor.data()), stream);
    ol->shrink(static_cast<std::size_t>(total));
    orr->shrink(static_cast<std::size_t>(total));
    return {std::move(ol), std::move(orr)};
  }

  std::size_t inner_join_size(table_view const& probe, rmm::cuda_stream_view stream) const
  {
    check_probe(probe);
    if (probe.num_rows() == 0 || _build.num_rows() == 0) return 0;
    auto const& pk = probe.column(0);
    rmm::device_buffer holder;
    auto const* pmask = pk.has_nulls() ? rebased_mask(pk, holder, stream) : nullptr;
    rmm::device_buffer cnt{sizeof(int64_t), stream};
    gx_check(gx_join_count(_key_size, row0(pk), pmask, pk.size(), _table.data(), _table_bytes,
                           static_cast<int64_t*>(cnt.data()), gxs(stream)),
             "hash_join count");
    auto total = static_cast<std::size_t>(read_i64(static_cast<int64_t const*>(cnt.data()), stream));
    if (_nulls_equal && pk.has_nulls()) total += static_cast<std::size_t>(pk.null_count()) * _build_nulls.size();
    return total;
  }

  [[nodiscard]] size_type build_rows() const { return _build.num_rows(); }

 private:
  table_view _build;
  bool _has_nulls;
  bool _nulls_equal;
  double _load_factor;
  int _key_size{0};
  std::size_t _table_bytes{0};
  rmm::device_buffer _table{};
  std::vector<size_type> _build_nulls{};
};

}  // namespace detail

hash_join::~hash_join() = default;

hash_join::hash_join(table_view const& build, null_equality compare_nulls, rmm::cuda_stream_view stream)
  : hash_join(build, nullable_join::YES, compare_nulls, 0.5, stream)
{
}

hash_join::hash_join(table_view const& build, nullable_join has_nulls, null_equality compare_nulls, double load_factor,
                     rmm::cuda_stream_view stream)
  : _impl{std::make_unique<detail::hash_join_impl const>(build, has_nulls, compare_nulls, load_factor, stream)}
{
}

join_result hash_join::inner_join(table_view const& probe, std::optional<std::size_t> output_size,
                                  rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr) const
{
  return _impl->inner_join(probe, output_size, stream, mr);
}
join_result hash_join::left_join(table_view const& probe, std::optional<std::size_t> output_size,
                                 rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr) const
{
  return _impl->left_join(probe, output_size, stream, mr);
}
join_result hash_join::full_join(table_view const& probe, std::optional<std::size_t> output_size,
                                 rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr) const
{
  return _impl->full_join(probe, output_size, stream, mr);
}
std::size_t hash_join::inner_join_size(table_view const& probe, rmm::cuda_stream_view stream) const
{
  return _impl->inner_join_size(probe, stream);
}
std::size_t hash_join::left_join_size(table_view const& probe, rmm::cuda_stream_view stream) const
{
  return _impl->left_join(probe, {}, stream, cudf::get_current_device_resource_ref()).first->size();
}
std::size_t hash_join::full_join_size(table_view const& probe, rmm::cuda_stream_view stream,
                                      rmm::device_async_resource_ref mr) const
{
  return _impl->full_join(probe, {}, stream, mr).first->size();
}

join_result inner_join(table_view const& left_keys, table_view const& right_keys, null_equality compare_nulls,
                       rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  // build on the smaller side, swap the pair back (join.cu:49-59)
  if (right_keys.num_rows() > left_keys.num_rows()) {
    hash_join hj{left_keys, nullable_join::YES, compare_nulls, 0.5, stream};
    auto res = hj.inner_join(right_keys, {}, stream, mr);
    return {std::move(res.second), std::move(res.first)};
  }
  hash_join hj{right_keys, nullable_join::YES, compare_nulls, 0.5, stream};
  return hj.inner_join(left_keys, {}, stream, mr);
}

join_result left_join(table_view const& left_keys, table_view const& right_keys, null_equality compare_nulls,
                      rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  hash_join hj{right_keys, nullable_join::YES, compare_nulls, 0.5, stream};
  return hj.left_join(left_keys, {}, stream, mr);
}

join_result full_join(table_view const& left_keys, table_view const& right_keys, null_equality compare_nulls,
                      rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  hash_join hj{right_keys, nullable_join::YES, compare_nulls, 0.5, stream};
  return hj.full_join(left_keys, {}, stream, mr);
}

}  // namespace cudf
