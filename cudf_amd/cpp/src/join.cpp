// cudf::hash_join and the free-function joins over the C ABI.
// reference: cpp/src/join/hash_join/hash_join.cu:32-198, cpp/src/join/join.cu:27-124,
// cpp/src/join/join_utils.cu:45-157.
#include "common.hpp"
#include "row_encoding.hpp"

#include <cudf/join/distinct_hash_join.hpp>
#include <cudf/join/filtered_join.hpp>
#include <cudf/join/hash_join.hpp>
#include <cudf/join/join.hpp>
#include <cudf/null_mask.hpp>

#include <algorithm>
#include <mutex>
#include <stdexcept>
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

// device int64 cursor <- v (the full 64 bits: the pair total of a partitioned join may exceed size_type), with no host
// source to outlive the call: each 32-bit half is written by the sequence kernel from a by-value argument
void set_cursor_async(rmm::device_buffer& cursor, std::size_t v, rmm::cuda_stream_view stream)
{
  CUDF_CUDA_TRY(hipMemsetAsync(cursor.data(), 0, sizeof(int64_t), stream.value()));
  auto* w = static_cast<int32_t*>(cursor.data());
  auto const lo = static_cast<int32_t>(static_cast<uint32_t>(v & 0xFFFFFFFFull));
  auto const hi = static_cast<int32_t>(static_cast<uint32_t>(v >> 32));
  if (lo != 0) detail::gx_check(gx_sequence_i32(w, 1, lo, detail::gxs(stream)), "cursor");
  if (hi != 0) detail::gx_check(gx_sequence_i32(w + 1, 1, hi, detail::gxs(stream)), "cursor");
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
  // allow_hash: rows wider than 8 bytes may be keyed by their 64-bit hash (row_encoder); pair-producing probes then
  // certify their output against the key columns, everything else goes through exact()
  hash_join_impl(table_view const& build, nullable_join has_nulls, null_equality compare_nulls, double load_factor,
                 rmm::cuda_stream_view stream, bool allow_hash = true)
    : _build{build}, _has_nulls{has_nulls == nullable_join::YES}, _nulls_equal{compare_nulls == null_equality::EQUAL},
      _load_factor{load_factor}
  {
    CUDF_EXPECTS(0 != build.num_columns(), "Hash join build table is empty", std::invalid_argument);
    CUDF_EXPECTS(load_factor > 0 && load_factor <= 1,
                 "Invalid load factor: must be greater than 0 and less than or equal to 1.", std::invalid_argument);
    // one fixed-width key per row: the column itself (4/8-byte integers), else the row encoding
    auto const& c0    = build.column(0);
    bool const direct = build.num_columns() == 1 && (size_of(c0.type()) == 4 || size_of(c0.type()) == 8) &&
                        !is_floating_point(c0.type());
    if (direct) {
      _key = c0;
    } else {
      _enc = std::make_unique<row_encoder>(build, _nulls_equal, stream, allow_hash);
      _key = _enc->build_keys();
    }
    auto const& key = _key;
    _key_size       = static_cast<int>(size_of(key.type()));
    if (build.num_rows() == 0) return;
    _table_bytes = gx_join_table_bytes(_key_size, key.size(), load_factor);
    _table       = rmm::device_buffer{_table_bytes, stream};
    rmm::device_buffer holder;
    auto const* mask = key.has_nulls() ? rebased_mask(key, holder, stream) : nullptr;
    if (mask == nullptr && key.size() >= (1 << 20) && gx_join_partition_bits(_key_size, _table_bytes) > 0) {
      // large build side: rows are partitioned first so that the inserts hit L2-resident sub-tables
      run_with_scratch(
        [&](void* t, std::size_t* b) {
          return gx_join_build_partitioned(_key_size, row0(key), key.size(), _table.data(), _table_bytes, load_factor, t, b,
                                           gxs(stream));
        },
        "hash_join build", stream);
    } else {
      gx_check(gx_join_build(_key_size, row0(key), mask, key.size(), _table.data(), _table_bytes, load_factor, gxs(stream)),
               "hash_join build");
    }
    _build_nulls = null_rows(key, stream);
  }

  [[nodiscard]] bool hashed() const { return _enc && _enc->hashed(); }
  // The exactly-encoded twin of a hashed table, built on first use: the operations that only COUNT matches cannot
  // be certified pair by pair, and a pair-producing probe lands here after a 64-bit collision.
  hash_join_impl const& exact(rmm::cuda_stream_view stream) const
  {
    std::lock_guard<std::mutex> lock(_exact_mutex);
    if (!_exact) {
      _exact = std::make_unique<hash_join_impl const>(_build, _has_nulls ? nullable_join::YES : nullable_join::NO,
                                                      _nulls_equal ? null_equality::EQUAL : null_equality::UNEQUAL, _load_factor,
                                                      stream, false);
      // cold path, once per table: probes are const and may come from other threads on other streams
      // (hash_join.hpp:63-68) -- none of them may see the twin before its build kernels have finished
      stream.synchronize();
    }
    return *_exact;
  }

  // the probe table's key column in the build side's key space (owner keeps an encoded column alive)
  column_view probe_key(table_view const& probe, std::unique_ptr<column>& owner, rmm::cuda_stream_view stream) const
  {
    if (!_enc) return probe.column(0);
    owner = _enc->encode(probe, stream);
    return owner->view();
  }

  void check_probe(table_view const& probe) const
  {
    CUDF_EXPECTS(probe.num_columns() == _build.num_columns(), "Mismatch in number of columns to be joined on",
                 std::invalid_argument);
    for (size_type i = 0; i < probe.num_columns(); ++i)
      CUDF_EXPECTS(probe.column(i).type() == _build.column(i).type(), "Mismatch in joining column data types",
                   cudf::data_type_error);
    CUDF_EXPECTS(_has_nulls || !cudf::has_nulls(probe), "Probe table has nulls while build table was not hashed with null check.",
                 std::invalid_argument);
  }

  // pairs from the kernel probe (+ null x null cross product when nulls compare equal)
  join_result probe_join(table_view const& probe, bool left_outer, std::optional<std::size_t> output_size,
                         rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr) const
  {
    check_probe(probe);
    std::unique_ptr<column> owner;
    auto const pk = probe_key(probe, owner, stream);
    rmm::device_buffer holder;
    auto const* pmask = pk.has_nulls() ? rebased_mask(pk, holder, stream) : nullptr;
    std::vector<size_type> pnulls = _nulls_equal ? null_rows(pk, stream) : std::vector<size_type>{};
    std::size_t const cross = _nulls_equal ? pnulls.size() * _build_nulls.size() : 0;
    // with nulls EQUAL a null probe row that has a null partner must not also emit (i, NoMatch): flag bit 1
    int const outer_flags = left_outer ? (cross > 0 ? 3 : 1) : 0;

    std::size_t capacity = output_size.value_or(static_cast<std::size_t>(pk.size())) + cross;
    rmm::device_buffer cursor{sizeof(int64_t), stream};
    map_ptr l, r;
    int64_t total = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
      l = std::make_unique<rmm::device_uvector<size_type>>(capacity, stream, mr);
      r = std::make_unique<rmm::device_uvector<size_type>>(capacity, stream, mr);
      CUDF_CUDA_TRY(hipMemsetAsync(cursor.data(), 0, sizeof(int64_t), stream.value()));
      if (pmask == nullptr && pk.size() >= (1 << 22) && gx_join_partition_bits(_key_size, _table_bytes) > 0) {
        // large probe against a table far beyond the L2s: partitioned probe (chains run on LDS tags)
        run_with_scratch(
          [&](void* t, std::size_t* b) {
            return gx_join_probe_partitioned(_key_size, row0(pk), pk.size(), _table.data(), _table_bytes, outer_flags & 1,
                                             l->data(), r->data(), static_cast<int64_t>(capacity),
                                             static_cast<int64_t*>(cursor.data()), t, b, gxs(stream));
          },
          "hash_join probe", stream);
      } else {
        gx_check(gx_join_probe(_key_size, row0(pk), pmask, pk.size(), _table.data(), _table_bytes, outer_flags,
                               l->data(), r->data(), static_cast<int64_t>(capacity), static_cast<int64_t*>(cursor.data()),
                               gxs(stream)),
                 "hash_join probe");
      }
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
    l->shrink(n);
    r->shrink(n);
    if (hashed() && n > 0 && count_row_mismatches(probe, _build, l->data(), r->data(), n, stream) != 0)
      return exact(stream).probe_join(probe, left_outer, output_size, stream, mr);  // a 64-bit collision
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
    set_cursor_async(cursor, n0, stream);
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
    if (hashed()) return exact(stream).inner_join_size(probe, stream);
    if (probe.num_rows() == 0 || _build.num_rows() == 0) return 0;
    std::unique_ptr<column> owner;
    auto const pk = probe_key(probe, owner, stream);
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

  // cudf::filtered_join: ascending rows of `probe` with (anti: without) a match
  map_ptr semi_anti(table_view const& probe, bool anti, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr) const
  {
    auto const n = probe.num_rows();
    if (n == 0) return std::make_unique<rmm::device_uvector<size_type>>(0, stream, mr);
    if (_build.num_rows() == 0 || _build.num_columns() == 0) {  // nothing can match
      auto out = std::make_unique<rmm::device_uvector<size_type>>(anti ? n : 0, stream, mr);
      if (anti) gx_check(gx_sequence_i32(out->data(), n, 0, gxs(stream)), "sequence");
      return out;
    }
    check_probe(probe);
    if (hashed()) return exact(stream).semi_anti(probe, anti, stream, mr);  // "has a match" cannot be certified pair by pair
    std::unique_ptr<column> owner;
    auto const pk = probe_key(probe, owner, stream);
    rmm::device_buffer holder;
    auto const* pmask = pk.has_nulls() ? rebased_mask(pk, holder, stream) : nullptr;
    auto out          = std::make_unique<rmm::device_uvector<size_type>>(n, stream, mr);
    rmm::device_buffer cnt{sizeof(int64_t), stream};
    int const null_matches = (_nulls_equal && !_build_nulls.empty()) ? 1 : 0;
    run_with_scratch(
      [&](void* t, std::size_t* b) {
        return gx_join_filter(_key_size, row0(pk), pmask, n, _table.data(), _table_bytes, anti ? 1 : 0, null_matches,
                              out->data(), static_cast<int64_t*>(cnt.data()), t, b, gxs(stream));
      },
      "filtered_join", stream);
    out->shrink(static_cast<std::size_t>(read_i64(static_cast<int64_t const*>(cnt.data()), stream)));
    return out;
  }

  // cudf::distinct_hash_join::left_join: the build row of every probe row, JoinNoMatch where none
  map_ptr lookup(table_view const& probe, rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr) const
  {
    auto const n = probe.num_rows();
    auto out     = std::make_unique<rmm::device_uvector<size_type>>(n, stream, mr);
    if (n == 0) return out;
    if (_build.num_rows() == 0) {
      std::vector<size_type> nm(n, JoinNoMatch);
      CUDF_CUDA_TRY(hipMemcpyAsync(out->data(), nm.data(), n * sizeof(size_type), hipMemcpyHostToDevice, stream.value()));
      stream.synchronize();
      return out;
    }
    check_probe(probe);
    std::unique_ptr<column> owner;
    auto const pk = probe_key(probe, owner, stream);
    rmm::device_buffer holder;
    auto const* pmask = pk.has_nulls() ? rebased_mask(pk, holder, stream) : nullptr;
    gx_check(gx_join_lookup(_key_size, row0(pk), pmask, n, _table.data(), _table_bytes, out->data(), gxs(stream)),
             "distinct_hash_join lookup");
    if (pmask && _nulls_equal && !_build_nulls.empty())  // null == null: the (single) null build row
      gx_check(gx_fill_nulls(4, out->data(), pmask, n, static_cast<uint64_t>(static_cast<uint32_t>(_build_nulls.front())),
                             gxs(stream)),
               "distinct_hash_join null rows");
    if (hashed() && count_row_mismatches(probe, _build, nullptr, out->data(), static_cast<std::size_t>(n), stream) != 0)
      return exact(stream).lookup(probe, stream, mr);  // a 64-bit collision
    return out;
  }

  // matches per probe row (at least `min_count`): *_join_match_context
  std::unique_ptr<rmm::device_uvector<size_type>> match_counts(table_view const& probe, size_type min_count,
                                                               rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr) const
  {
    check_probe(probe);
    if (hashed()) return exact(stream).match_counts(probe, min_count, stream, mr);
    auto const n = probe.num_rows();
    auto counts  = std::make_unique<rmm::device_uvector<size_type>>(n, stream, mr);
    if (n == 0) return counts;
    if (_build.num_rows() == 0) {
      std::vector<size_type> h(n, min_count);
      CUDF_CUDA_TRY(hipMemcpyAsync(counts->data(), h.data(), n * sizeof(size_type), hipMemcpyHostToDevice, stream.value()));
      stream.synchronize();
      return counts;
    }
    std::unique_ptr<column> owner;
    auto const pk = probe_key(probe, owner, stream);
    rmm::device_buffer holder;
    auto const* pmask = pk.has_nulls() ? rebased_mask(pk, holder, stream) : nullptr;
    gx_check(gx_join_count_rows(_key_size, row0(pk), pmask, n, _table.data(), _table_bytes, min_count, counts->data(), gxs(stream)),
             "hash_join match counts");
    if (pmask && _nulls_equal && !_build_nulls.empty()) {  // null == null: a null probe row matches every null build row
      auto const c = std::max<size_type>(min_count, static_cast<size_type>(_build_nulls.size()));
      gx_check(gx_fill_nulls(4, counts->data(), pmask, n, static_cast<uint64_t>(static_cast<uint32_t>(c)), gxs(stream)),
               "hash_join match counts of null rows");
    }
    return counts;
  }

  // one chunk [start, end) of the context's left table; left indices refer to the whole table
  join_result partitioned_join(join_partition_context const& ctx, bool left_outer, rmm::cuda_stream_view stream,
                               rmm::device_async_resource_ref mr) const
  {
    CUDF_EXPECTS(ctx.left_table_context != nullptr, "partitioned join: missing match context", std::invalid_argument);
    auto const& left = ctx.left_table_context->_left_table;
    auto const start = ctx.left_start_idx, end = ctx.left_end_idx;
    CUDF_EXPECTS(0 <= start && start <= end && end <= left.num_rows(), "partitioned join: chunk out of range", std::out_of_range);
    if (start == end) return empty_result(stream, mr);
    std::vector<column_view> cols;
    for (auto const& c : left) {
      size_type nulls = 0;
      if (c.nullable() && c.null_count() > 0) nulls = cudf::null_count(c.null_mask(), c.offset() + start, c.offset() + end, stream);
      cols.emplace_back(c.type(), end - start, c.head<void>(), c.null_mask(), nulls, c.offset() + start);
    }
    table_view chunk{cols};
    join_result res = left_outer ? left_join(chunk, {}, stream, mr)
                                 : ((_build.num_rows() == 0) ? empty_result(stream, mr) : probe_join(chunk, false, {}, stream, mr));
    gx_check(gx_add_i32(res.first->data(), static_cast<int64_t>(res.first->size()), start, gxs(stream)), "partitioned join: re-base");
    return res;
  }

  [[nodiscard]] size_type build_rows() const { return _build.num_rows(); }

 private:
  table_view _build;
  bool _has_nulls;
  bool _nulls_equal;
  double _load_factor;
  std::unique_ptr<row_encoder> _enc{};
  mutable std::mutex _exact_mutex{};
  mutable std::unique_ptr<hash_join_impl const> _exact{};
  column_view _key{};
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

join_match_context hash_join::inner_join_match_context(table_view const& left, rmm::cuda_stream_view stream,
                                                       rmm::device_async_resource_ref mr) const
{
  return join_match_context{left, _impl->match_counts(left, 0, stream, mr)};
}
join_match_context hash_join::left_join_match_context(table_view const& left, rmm::cuda_stream_view stream,
                                                      rmm::device_async_resource_ref mr) const
{
  return join_match_context{left, _impl->match_counts(left, 1, stream, mr)};  // an unmatched row still emits one pair
}
join_match_context hash_join::full_join_match_context(table_view const& left, rmm::cuda_stream_view stream,
                                                      rmm::device_async_resource_ref mr) const
{
  return join_match_context{left, _impl->match_counts(left, 1, stream, mr)};
}
join_result hash_join::partitioned_inner_join(join_partition_context const& context, rmm::cuda_stream_view stream,
                                              rmm::device_async_resource_ref mr) const
{
  return _impl->partitioned_join(context, false, stream, mr);
}
join_result hash_join::partitioned_left_join(join_partition_context const& context, rmm::cuda_stream_view stream,
                                             rmm::device_async_resource_ref mr) const
{
  return _impl->partitioned_join(context, true, stream, mr);
}
join_result hash_join::partitioned_full_join(join_partition_context const& context, rmm::cuda_stream_view stream,
                                             rmm::device_async_resource_ref mr) const
{
  return _impl->partitioned_join(context, true, stream, mr);  // the unmatched right rows come from finalize
}
// hash_join.hpp:420-440 / join_utils.cu:86-157: concatenate the chunks, append (JoinNoMatch, r) for every right row no
// chunk matched
join_result hash_join::finalize_partitioned_full_join(host_span<device_span<size_type const> const> left_partials,
                                                      host_span<device_span<size_type const> const> right_partials,
                                                      size_type left_table_num_rows, size_type right_table_num_rows,
                                                      rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  (void)left_table_num_rows;
  CUDF_EXPECTS(left_partials.size() == right_partials.size(), "finalize_partitioned_full_join: partial counts differ",
               std::invalid_argument);
  std::size_t total = 0;
  for (std::size_t i = 0; i < left_partials.size(); ++i) {
    CUDF_EXPECTS(left_partials[i].size() == right_partials[i].size(), "finalize_partitioned_full_join: partial sizes differ",
                 std::invalid_argument);
    total += left_partials[i].size();
  }
  auto const cap = total + static_cast<std::size_t>(right_table_num_rows);
  auto l = std::make_unique<rmm::device_uvector<size_type>>(cap, stream, mr);
  auto r = std::make_unique<rmm::device_uvector<size_type>>(cap, stream, mr);
  std::size_t at = 0;
  for (std::size_t i = 0; i < left_partials.size(); ++i) {
    auto const n = left_partials[i].size();
    if (n == 0) continue;
    CUDF_CUDA_TRY(hipMemcpyAsync(l->data() + at, left_partials[i].data(), n * sizeof(size_type), hipMemcpyDeviceToDevice, stream.value()));
    CUDF_CUDA_TRY(hipMemcpyAsync(r->data() + at, right_partials[i].data(), n * sizeof(size_type), hipMemcpyDeviceToDevice, stream.value()));
    at += n;
  }
  rmm::device_buffer cursor{sizeof(int64_t), stream};
  detail::set_cursor_async(cursor, total, stream);
  if (right_table_num_rows > 0)
    detail::run_with_scratch(
      [&](void* t, std::size_t* b) {
        return gx_join_complement(r->data(), static_cast<int64_t>(total), right_table_num_rows, l->data(), r->data(),
                                  static_cast<int64_t>(cap), static_cast<int64_t*>(cursor.data()), t, b, detail::gxs(stream));
      },
      "finalize_partitioned_full_join", stream);
  auto const n = detail::read_i64(static_cast<int64_t const*>(cursor.data()), stream);
  l->shrink(static_cast<std::size_t>(n));
  r->shrink(static_cast<std::size_t>(n));
  return {std::move(l), std::move(r)};
}

// ---- cudf::filtered_join (include/cudf/join/filtered_join.hpp:51-144; src/join/filtered_join/filtered_join.cu)
filtered_join::~filtered_join() = default;
filtered_join::filtered_join(table_view const& right, null_equality compare_nulls, rmm::cuda_stream_view stream)
  : filtered_join(right, compare_nulls, 0.5, stream)
{
}
filtered_join::filtered_join(table_view const& right, null_equality compare_nulls, double load_factor,
                             rmm::cuda_stream_view stream)
{
  CUDF_EXPECTS(load_factor > 0 && load_factor <= 1,
               "Invalid load factor: must be greater than 0 and less than or equal to 1.", std::invalid_argument);
  _right_rows = right.num_rows();
  // an empty right table (no columns or no rows) matches nothing (semi_anti_join_tests.cpp:357-421)
  if (right.num_columns() > 0 && right.num_rows() > 0)
    _impl = std::make_unique<detail::hash_join_impl const>(right, nullable_join::YES, compare_nulls, load_factor, stream,
                                                                  false);  // semi / anti results cannot be certified pair by pair: exact keys
}
std::unique_ptr<rmm::device_uvector<size_type>> filtered_join::semi_join(table_view const& left, rmm::cuda_stream_view stream,
                                                                         rmm::device_async_resource_ref mr) const
{
  if (!_impl) return std::make_unique<rmm::device_uvector<size_type>>(0, stream, mr);
  return _impl->semi_anti(left, false, stream, mr);
}
std::unique_ptr<rmm::device_uvector<size_type>> filtered_join::anti_join(table_view const& left, rmm::cuda_stream_view stream,
                                                                         rmm::device_async_resource_ref mr) const
{
  if (!_impl) {  // every left row
    auto const n = left.num_rows();
    auto out     = std::make_unique<rmm::device_uvector<size_type>>(n, stream, mr);
    if (n) detail::gx_check(gx_sequence_i32(out->data(), n, 0, detail::gxs(stream)), "sequence");
    return out;
  }
  return _impl->semi_anti(left, true, stream, mr);
}

// ---- cudf::distinct_hash_join (include/cudf/join/distinct_hash_join.hpp:50-124; src/join/distinct_hash_join.cu)
distinct_hash_join::~distinct_hash_join() = default;
distinct_hash_join::distinct_hash_join(table_view const& right, null_equality compare_nulls, double load_factor,
                                       rmm::cuda_stream_view stream)
{
  CUDF_EXPECTS(load_factor > 0 && load_factor <= 1,
               "Invalid load factor: must be greater than 0 and less than or equal to 1.", std::invalid_argument);
  CUDF_EXPECTS(0 != right.num_columns(), "Hash join build table is empty", std::invalid_argument);
  _impl = std::make_unique<detail::hash_join_impl const>(right, nullable_join::YES, compare_nulls, load_factor, stream);
}
join_result distinct_hash_join::inner_join(table_view const& left, rmm::cuda_stream_view stream,
                                           rmm::device_async_resource_ref mr) const
{
  return _impl->inner_join(left, {}, stream, mr);
}
std::unique_ptr<rmm::device_uvector<size_type>> distinct_hash_join::left_join(table_view const& left,
                                                                              rmm::cuda_stream_view stream,
                                                                              rmm::device_async_resource_ref mr) const
{
  return _impl->lookup(left, stream, mr);
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
