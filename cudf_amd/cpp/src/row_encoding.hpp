// row_encoding.hpp -- multi-column keys -> one fixed-width key per row (host side of gx_pack_keys /
// gx_dense_rank / gx_join_lookup).  The reference compares whole rows inside its hash tables
// (cpp/include/cudf/detail/row_operator/primitive_row_operators.cuh:207-274, equality.cuh); here the
// single-key kernels stay as they are and the rows are encoded first, cf. the reference's own
// cudf::key_remapping (cpp/include/cudf/join/key_remapping.hpp).
#pragma once
#include "common.hpp"

#include <cudf/table/table.hpp>
#include <cudf/table/table_view.hpp>

#include <memory>
#include <vector>

namespace cudf {
namespace detail {

// UINT64 column: the columns' values concatenated (widths must sum to <= 8 bytes; floats normalised
// so that -0.0 == +0.0 and NaN == NaN).  The result carries no validity.
std::unique_ptr<column> pack_columns(std::vector<column_view> const& cols, rmm::cuda_stream_view stream);

// UINT64 column: a 64-bit hash of every row of the columns (any widths; floats normalised as in pack_columns).  Equal
// rows hash equal; a result obtained through the hashes is certified by count_row_mismatches == 0.
std::unique_ptr<column> hash_columns(std::vector<column_view> const& cols, rmm::cuda_stream_view stream);
// number of pairs (lidx[i], ridx[i]) -- nullptr = row i itself, a negative index = no row -- whose rows differ
int64_t count_row_mismatches(table_view const& left, table_view const& right, size_type const* lidx, size_type const* ridx,
                             std::size_t npairs, rmm::cuda_stream_view stream);

struct dense_rank_result {
  std::unique_ptr<column> ids;  // INT32, one per row, no validity (null == null has its own, last, id)
  std::unique_ptr<column> rep;  // INT32, one per id: the smallest row with that id
  size_type num_ids{0};
};
dense_rank_result dense_rank(column_view const& col, rmm::cuda_stream_view stream);

// Dense ids of the rows of ONE table (groupby keys): rows holding a null in any column are marked
// null in `ids` (null_policy::EXCLUDE drops them); rep[id] = first row of the id.
// Exact, one radix sort per key column (gx_dense_rank): the fallback of row_keys below.
dense_rank_result dense_row_ids(table_view const& keys, rmm::cuda_stream_view stream);

// ONE 8-byte key per row of a groupby key table, without sorting: the packed column values when their widths sum to
// <= 8 bytes (exact), else a 64-bit row hash; rows holding a null key get a null key (null_policy::EXCLUDE drops them).
// The single-key hash groupby runs on view(); key_columns() turns the distinct keys of its result back into key
// columns (same order) and -- for hashed keys -- certifies that no two different rows shared a hash: nullptr means a
// 64-bit collision, and the caller re-runs through dense_row_ids.
class row_keys {
 public:
  row_keys(table_view const& keys, rmm::cuda_stream_view stream);
  [[nodiscard]] column_view view() const { return _col->view(); }
  [[nodiscard]] bool exact() const { return _exact; }
  [[nodiscard]] std::unique_ptr<table> key_columns(column_view const& distinct, rmm::cuda_stream_view stream,
                                                   rmm::device_async_resource_ref mr) const;

 private:
  table_view _keys;
  std::vector<column_view> _bare;
  bool _exact{false};
  std::unique_ptr<column> _col;
};

// Encoder for hash_join: learns the id space from the BUILD table once, then maps any probe table
// with the same schema into it.  Keys of rows that cannot equal any build row come out as values
// no build key has (or null, when nulls compare unequal), so the single-key join does the rest.
class row_encoder {
 public:
  // allow_hash: rows wider than 8 bytes are keyed by a 64-bit row hash (one pass) when the null semantics permit; the
  // caller must then certify every result with count_row_mismatches and re-run with allow_hash = false on a collision
  row_encoder(table_view const& build, bool nulls_equal, rmm::cuda_stream_view stream, bool allow_hash = false);
  [[nodiscard]] bool hashed() const { return _hashed; }
  ~row_encoder();
  [[nodiscard]] column_view build_keys() const { return _build_keys->view(); }
  [[nodiscard]] std::unique_ptr<column> encode(table_view const& probe, rmm::cuda_stream_view stream) const;

 private:
  struct dictionary {  // distinct packed values of one level -> their dense id
    rmm::device_buffer table;
    std::size_t table_bytes{0};
    size_type null_id{-1};  // id of "null" at this level, -1 if the build side had none
    size_type num_ids{0};
  };
  dictionary make_dictionary(column_view const& packed, dense_rank_result const& r, size_type nulls,
                             rmm::cuda_stream_view stream) const;
  std::unique_ptr<column> lookup(dictionary const& d, column_view const& packed, bitmask_type const* valid,
                                 rmm::cuda_stream_view stream) const;

  bool _nulls_equal;
  bool _pack_only{false};  // widths sum to <= 8 bytes and the build side has no nulls
  bool _hashed{false};     // keys are 64-bit row hashes
  std::vector<data_type> _types;
  std::vector<dictionary> _col_dict;   // one per column
  std::vector<dictionary> _pair_dict;  // one per inner pair level (columns - 2)
  std::unique_ptr<column> _build_keys;
};

}  // namespace detail
}  // namespace cudf
