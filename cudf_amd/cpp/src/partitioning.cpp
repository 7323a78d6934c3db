// cudf::hash_partition / cudf::partition over the C ABI (gx_murmur3_32 -> gx_hash_partition_map -> gx_gather).
// reference: cpp/src/partitioning/partitioning.cu:53-92 (modulo / bitwise partitioners), 568-745 (hash_partition_table),
// 875-921 (empty results, partition by map), 923-972 (front ends); contract pinned by
// cpp/tests/partitioning/hash_partition_test.cpp:49-141,190-209 and partition_test.cpp.
#include "common.hpp"

#include <cudf/column/column_factories.hpp>
#include <cudf/copying.hpp>
#include <cudf/hashing.hpp>
#include <cudf/partitioning.hpp>

namespace cudf {
namespace {

// cudf::empty_like(table_view) (cpp/src/copying/copy.cpp): the columns' types, zero rows
std::unique_ptr<table> empty_like_table(table_view const& t)
{
  std::vector<std::unique_ptr<column>> cols;
  cols.reserve(t.num_columns());
  for (auto const& c : t) cols.emplace_back(make_empty_column(c.type()));
  return std::make_unique<table>(std::move(cols));
}

// rows regrouped by value[i] % num_partitions through a stable map; ALWAYS num_partitions + 1 offsets, last = rows
// (partitioning.cu:684-688: "Add the total row count as the last offset")
std::pair<std::unique_ptr<table>, std::vector<size_type>> regroup(table_view const& input, uint32_t const* value, int num_partitions,
                                                                  rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  auto const n = input.num_rows();
  rmm::device_uvector<int32_t> map(n, stream), offs(num_partitions + 1, stream);
  detail::run_with_scratch(
    [&](void* t, std::size_t* b) {
      return gx_hash_partition_map(value, n, num_partitions, map.data(), offs.data(), t, b, detail::gxs(stream));
    },
    "hash_partition", stream);
  std::vector<size_type> offsets(static_cast<std::size_t>(num_partitions) + 1);
  CUDF_CUDA_TRY(hipMemcpyAsync(offsets.data(), offs.data(), offsets.size() * sizeof(size_type), hipMemcpyDeviceToHost, stream.value()));
  column_view mapv{data_type{type_id::INT32}, n, map.data(), nullptr, 0};
  auto out = gather(input, mapv, out_of_bounds_policy::DONT_CHECK, stream, mr);
  stream.synchronize();  // the offsets are a host vector
  offsets.back() = n;
  return {std::move(out), std::move(offsets)};
}

}  // namespace

std::pair<std::unique_ptr<table>, std::vector<size_type>> hash_partition(table_view const& input, table_view const& keys, int num_partitions,
                                                                         hash_id hash_function, uint32_t seed, rmm::cuda_stream_view stream,
                                                                         rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(keys.num_columns() == 0 || input.num_rows() == keys.num_rows(),
               "Input table and key table must have same number of rows, or key table should have no columns.", std::invalid_argument);
  CUDF_EXPECTS(hash_function == hash_id::HASH_MURMUR3 || hash_function == hash_id::HASH_IDENTITY, "Unsupported hash function in hash_partition");
  // no partitions, no rows or nothing to hash: an empty table and num_partitions + 1 zeros (partitioning.cu:883-886)
  if (num_partitions <= 0 || input.num_rows() == 0 || keys.num_columns() == 0)
    return {empty_like_table(input), std::vector<size_type>(static_cast<std::size_t>(std::max(num_partitions, 0)) + 1, 0)};
  if (hash_function == hash_id::HASH_IDENTITY) {
    // IdentityHash<T> = static_cast<uint32_t>(element) for every numeric T (partitioning.cu:852-872, 885-889), rows folded like any
    // row hash: the first column's element hash, later columns through hash_combine, a null element = UINT32_MAX
    // (row_operator/hashing.cuh:118-134).  The seed is not used (IdentityHash(uint32_t) {}).  What the reference's own tests use it
    // for: a key column of externally computed row hashes (hash_partition_test.cpp:411-415).
    for (auto const& c : keys) CUDF_EXPECTS(is_numeric(c.type()), "IdentityHash does not support this data type");
    rmm::device_uvector<uint32_t> h(input.num_rows(), stream);
    for (size_type k = 0; k < keys.num_columns(); ++k) {
      auto const& c = keys.column(k);
      rmm::device_buffer holder;
      auto const* mask = c.has_nulls() ? detail::rebased_mask(c, holder, stream) : nullptr;
      detail::gx_check(gx_identity_hash_32(detail::gx_type(c.type()), detail::row0(c), mask, c.size(), k > 0 ? 1 : 0, h.data(), detail::gxs(stream)),
                       "gx_identity_hash_32");
    }
    return regroup(input, h.data(), num_partitions, stream, mr);
  }
  auto h = hashing::murmurhash3_x86_32(keys, seed, stream);
  return regroup(input, h->view().head<uint32_t>(), num_partitions, stream, mr);
}

std::pair<std::unique_ptr<table>, std::vector<size_type>> hash_partition(table_view const& input, std::vector<size_type> const& columns_to_hash,
                                                                         int num_partitions, hash_id hash_function, uint32_t seed,
                                                                         rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  // (select() throws std::out_of_range for an invalid index: partitioning.hpp:91)
  return hash_partition(input, input.select(columns_to_hash), num_partitions, hash_function, seed, stream, mr);
}

std::pair<std::unique_ptr<table>, std::vector<size_type>> partition(table_view const& t, column_view const& partition_map, size_type num_partitions,
                                                                    rmm::cuda_stream_view stream, rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(t.num_rows() == partition_map.size(), "Size mismatch between table and partition map.");
  CUDF_EXPECTS(!partition_map.has_nulls(), "Unexpected null values in partition_map.");
  if (num_partitions <= 0 || t.num_rows() == 0)
    return {empty_like_table(t), std::vector<size_type>(static_cast<std::size_t>(std::max(num_partitions, 0)) + 1, 0)};
  // any integral map type but bool (partitioning.cu:780-842: is_index_type, else "Unexpected, non-integral partition map.")
  CUDF_EXPECTS(is_index_type(partition_map.type()), "Unexpected, non-integral partition map.");
  // map values lie in [0, num_partitions) (partitioning.hpp:52-55), so value % num_partitions is the value: the hash-partition
  // kernels regroup by the map's low 32 bits
  if (partition_map.type().id() == type_id::INT32 || partition_map.type().id() == type_id::UINT32)
    return regroup(t, static_cast<uint32_t const*>(detail::row0(partition_map)), num_partitions, stream, mr);
  rmm::device_uvector<uint32_t> ids(t.num_rows(), stream);
  detail::gx_check(gx_identity_hash_32(detail::gx_type(partition_map.type()), detail::row0(partition_map), nullptr, partition_map.size(), 0, ids.data(),
                                       detail::gxs(stream)),
                   "gx_identity_hash_32");
  return regroup(t, ids.data(), num_partitions, stream, mr);
}

// cudf::round_robin_partition (src/partitioning/round_robin.cu:150-275): the reference computes the gather map in closed form; here
// the partition ids (start_partition + i, taken modulo num_partitions by the regrouping kernels) go through the same stable regrouping
// as cudf::partition -- rows of a partition in input order, as dealing cards leaves them
std::pair<std::unique_ptr<table>, std::vector<size_type>> round_robin_partition(table_view const& input, size_type num_partitions,
                                                                               size_type start_partition, rmm::cuda_stream_view stream,
                                                                               rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(num_partitions > 0, "Incorrect number of partitions. Must be greater than 0.");
  CUDF_EXPECTS(start_partition < num_partitions, "Incorrect start_partition index. Must be less than number of partitions.");
  CUDF_EXPECTS(start_partition >= 0, "Incorrect start_partition index. Must be positive.");
  auto const n = input.num_rows();
  if (n == 0) return {empty_like_table(input), std::vector<size_type>(static_cast<std::size_t>(num_partitions) + 1, 0)};
  if (input.num_columns() == 0) {  // no column to move: the offsets alone (round_robin_test.cpp:32-42)
    std::vector<size_type> offs(static_cast<std::size_t>(num_partitions) + 1, 0);
    for (size_type p = 0; p < num_partitions; ++p) {
      // rows i with (start + i) % P == p: i = (p - start) mod P, then every P-th
      size_type const first = ((p - start_partition) % num_partitions + num_partitions) % num_partitions;
      offs[p + 1]           = offs[p] + (first < n ? (n - first + num_partitions - 1) / num_partitions : 0);
    }
    return {std::make_unique<table>(input, stream, mr), std::move(offs)};
  }
  rmm::device_uvector<int32_t> ids(n, stream);
  detail::gx_check(gx_sequence_i32(ids.data(), n, start_partition, detail::gxs(stream)), "round_robin_partition");
  // (start_partition + i < 2^32: as uint32 the sum is exact even where the int32 wrapped)
  return regroup(input, reinterpret_cast<uint32_t const*>(ids.data()), num_partitions, stream, mr);
}

}  // namespace cudf
