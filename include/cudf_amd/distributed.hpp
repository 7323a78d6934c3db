// cudf_amd/distributed.hpp -- C++ face of the sharded operators (C ABI: cudf_amd/gxd.h; implementation:
// cudf_amd/cpp/src/distributed.cpp in libcudf.so, RCCL over xGMI, one process per GPU).  Takes / returns the cudf types of
// include/cudf: every rank passes its SHARD and receives its shard of the result.  Reference analogues: the shuffle of
// libcudf_streaming (cpp/libcudf_streaming/src/partition_utils.cpp:72-117, partition.cpp:56-80) and the collectives of
// cudf_polars' streaming executor (collectives/sort.py, streaming/join.py:58-135, streaming/groupby.py:411-437).
#pragma once
#include <cudf/column/column.hpp>
#include <cudf/column/column_view.hpp>
#include <cudf/table/table.hpp>
#include <cudf/types.hpp>
#include <cudf/utilities/error.hpp>
#include <cudf_amd/gxd.h>
#include <rmm/device_buffer.hpp>

#include <array>
#include <memory>
#include <utility>
#include <vector>

namespace cudf_amd {
namespace distributed {

using unique_id = std::array<char, 128>;
inline unique_id make_unique_id()
{
  unique_id id{};
  CUDF_EXPECTS(gxd_unique_id(id.data()) == 0, gxd_last_error());
  return id;
}

// an RCCL communicator of `world` ranks on the current device; collective over the ranks (ncclCommInitRank)
class communicator {
 public:
  communicator(unique_id const& id, int world, int rank) { CUDF_EXPECTS(gxd_comm_create(id.data(), world, rank, &_c) == 0, gxd_last_error()); }
  ~communicator() { gxd_comm_destroy(_c); }
  communicator(communicator const&)            = delete;
  communicator& operator=(communicator const&) = delete;
  // `world` LOGICAL ranks on the current device (gxd_comm_create_loopback): device-to-device copies stand in for the links.  Each
  // communicator must be driven by a host thread of its own on a NON-BLOCKING stream of its own (the operators are collective).
  static std::vector<std::unique_ptr<communicator>> loopback(int world)
  {
    std::vector<gxd_comm*> raw(static_cast<std::size_t>(world > 0 ? world : 0), nullptr);
    CUDF_EXPECTS(world > 0 && gxd_comm_create_loopback(world, raw.data()) == 0, gxd_last_error());
    std::vector<std::unique_ptr<communicator>> out;
    for (auto* r : raw) out.emplace_back(std::unique_ptr<communicator>(new communicator(r)));
    return out;
  }
  [[nodiscard]] int rank() const { return gxd_comm_rank(_c); }
  [[nodiscard]] int world() const { return gxd_comm_world(_c); }
  [[nodiscard]] gxd_comm* get() const { return _c; }

 private:
  explicit communicator(gxd_comm* c) : _c{c} {}
  gxd_comm* _c{nullptr};
};

namespace detail {
// results are allocated by the C side through this callback: device_buffers from the caller's memory resource
struct result_alloc {
  rmm::cuda_stream_view stream;
  rmm::device_async_resource_ref mr;
  std::vector<rmm::device_buffer> bufs;
  static void* fn(std::size_t bytes, void* ctx)
  {
    auto* self = static_cast<result_alloc*>(ctx);
    try {
      self->bufs.emplace_back(bytes, self->stream, self->mr);
    } catch (...) {
      return nullptr;
    }
    return self->bufs.back().data();
  }
  rmm::device_buffer take(void const* p)
  {
    for (auto& b : bufs)
      if (b.data() == p) return std::move(b);
    return rmm::device_buffer{0, stream, mr};
  }
};
inline gx_stream_t gxs(rmm::cuda_stream_view s) { return reinterpret_cast<gx_stream_t>(s.value()); }
inline void check(int rc) { CUDF_EXPECTS(rc == 0, gxd_last_error()); }
}  // namespace detail

// Global sort of one fixed-width column without nulls: rank r returns the r-th range, sorted (cudf::sort, sharded).
inline std::unique_ptr<cudf::column> sort(cudf::column_view const& keys, communicator& comm, bool force_exchange = false,
                                          rmm::cuda_stream_view stream      = cudf::get_default_stream(),
                                          rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref())
{
  CUDF_EXPECTS(!keys.has_nulls(), "distributed::sort: columns with nulls are not supported on this path");
  detail::result_alloc ra{stream, mr, {}};
  void* out  = nullptr;
  int64_t n  = 0;
  auto const* p = static_cast<char const*>(keys.head<void>()) + static_cast<std::size_t>(keys.offset()) * cudf::size_of(keys.type());
  detail::check(gxd_sort(comm.get(), static_cast<int>(keys.type().id()), p, keys.size(), 0, force_exchange ? 1 : 0, &detail::result_alloc::fn, &ra,
                         &out, &n, detail::gxs(stream)));
  CUDF_EXPECTS(n <= static_cast<int64_t>(std::numeric_limits<cudf::size_type>::max()), "distributed::sort: shard exceeds size_type",
               std::overflow_error);
  return std::make_unique<cudf::column>(keys.type(), static_cast<cudf::size_type>(n), ra.take(out), rmm::device_buffer{}, 0);
}

// cudf::hash_join over sharded tables: the build side is exchanged and hashed once, every probe moves only the probe side.
// Pairs come back as INT64 columns of GLOBAL row ids (first row of the owning rank's shard + local row).
class hash_join {
 public:
  hash_join(cudf::column_view const& build_keys, communicator& comm, bool force_exchange = false,
            rmm::cuda_stream_view stream = cudf::get_default_stream())
  {
    CUDF_EXPECTS(!build_keys.has_nulls(), "distributed::hash_join: keys with nulls are not supported on this path");
    _type         = build_keys.type();
    auto const* p = static_cast<char const*>(build_keys.head<void>()) + static_cast<std::size_t>(build_keys.offset()) * cudf::size_of(_type);
    detail::check(gxd_join_build(comm.get(), static_cast<int>(_type.id()), p, build_keys.size(), force_exchange ? 1 : 0, detail::gxs(stream), &_j));
  }
  ~hash_join() { gxd_join_destroy(_j); }
  hash_join(hash_join const&)            = delete;
  hash_join& operator=(hash_join const&) = delete;

  [[nodiscard]] std::pair<std::unique_ptr<cudf::column>, std::unique_ptr<cudf::column>> inner_join(
    cudf::column_view const& probe_keys, rmm::cuda_stream_view stream = cudf::get_default_stream(),
    rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref()) const
  {
    CUDF_EXPECTS(probe_keys.type() == _type, "Mismatch in joining column data types", cudf::data_type_error);
    CUDF_EXPECTS(!probe_keys.has_nulls(), "distributed::hash_join: keys with nulls are not supported on this path");
    detail::result_alloc ra{stream, mr, {}};
    int64_t *l = nullptr, *r = nullptr, n = 0;
    auto const* p = static_cast<char const*>(probe_keys.head<void>()) + static_cast<std::size_t>(probe_keys.offset()) * cudf::size_of(_type);
    detail::check(gxd_join_probe(_j, p, probe_keys.size(), 0, &detail::result_alloc::fn, &ra, &l, &r, &n, detail::gxs(stream)));
    CUDF_EXPECTS(n <= static_cast<int64_t>(std::numeric_limits<cudf::size_type>::max()), "distributed::hash_join: shard of the result exceeds size_type",
                 std::overflow_error);
    auto const t = cudf::data_type{cudf::type_id::INT64};
    return {std::make_unique<cudf::column>(t, static_cast<cudf::size_type>(n), ra.take(l), rmm::device_buffer{}, 0),
            std::make_unique<cudf::column>(t, static_cast<cudf::size_type>(n), ra.take(r), rmm::device_buffer{}, 0)};
  }

 private:
  gxd_join* _j{nullptr};
  cudf::data_type _type{cudf::type_id::EMPTY};
};

// groupby(keys).agg(sum, count) over all shards; {keys ascending, sums (INT64 / FLOAT64), counts (INT64)} of the groups this rank owns
inline std::unique_ptr<cudf::table> groupby_sum_count(cudf::column_view const& keys, cudf::column_view const& values, communicator& comm,
                                                      bool force_exchange = false, rmm::cuda_stream_view stream = cudf::get_default_stream(),
                                                      rmm::device_async_resource_ref mr = cudf::get_current_device_resource_ref())
{
  CUDF_EXPECTS(keys.size() == values.size(), "Size mismatch between request values and groupby keys.", std::invalid_argument);
  CUDF_EXPECTS(!keys.has_nulls() && !values.has_nulls(), "distributed::groupby_sum_count: nulls are not supported on this path");
  detail::result_alloc ra{stream, mr, {}};
  void *ok = nullptr, *os = nullptr;
  int64_t* oc = nullptr;
  int64_t g   = 0;
  auto at = [](cudf::column_view const& c) { return static_cast<char const*>(c.head<void>()) + static_cast<std::size_t>(c.offset()) * cudf::size_of(c.type()); };
  detail::check(gxd_groupby_sum_count(comm.get(), static_cast<int>(keys.type().id()), at(keys), static_cast<int>(values.type().id()), at(values),
                                      keys.size(), 0, force_exchange ? 1 : 0, &detail::result_alloc::fn, &ra, &ok, &os, &oc, &g, detail::gxs(stream)));
  auto const sum_t = cudf::data_type{cudf::is_floating_point(values.type()) ? cudf::type_id::FLOAT64 : cudf::type_id::INT64};
  std::vector<std::unique_ptr<cudf::column>> cols;
  cols.emplace_back(std::make_unique<cudf::column>(keys.type(), static_cast<cudf::size_type>(g), ra.take(ok), rmm::device_buffer{}, 0));
  cols.emplace_back(std::make_unique<cudf::column>(sum_t, static_cast<cudf::size_type>(g), ra.take(os), rmm::device_buffer{}, 0));
  cols.emplace_back(std::make_unique<cudf::column>(cudf::data_type{cudf::type_id::INT64}, static_cast<cudf::size_type>(g), ra.take(oc), rmm::device_buffer{}, 0));
  return std::make_unique<cudf::table>(std::move(cols));
}

}  // namespace distributed
}  // namespace cudf_amd
