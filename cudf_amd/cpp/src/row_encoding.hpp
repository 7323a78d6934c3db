// row_encoding.hpp -- multi-column keys -> one fixed-width key per row (host side of gx_pack_keys /
// gx_dense_rank / gx_join_lookup).  The reference compares whole rows inside its hash tables
// (cpp/include/cudf/detail/row_operator/primitive_row_operators.cuh:207-274, equality.cuh); here the
// single-key kernels stay as they are and the rows are encoded first, cf. the reference's own
// cudf::key_remapping (cpp/include/cudf/join/key_remapping.hpp).
#pragma once
#include "common.hpp"

#include <cudf/table/table_view.hpp>

#include <memory>
#include <vector>

namespace cudf {
namespace detail {

// UINT64 column: the columns' values concatenated (widths must sum to <= 8 bytes; floats normalised
// so that -0.0 == +0.0 and NaN == NaN).  The result carries no validity.
std::unique_ptr<column> pack_columns(std::vector<column_view> const& cols, rmm::cuda_stream_view stream);

struct dense_rank_result {
  std::unique_ptr<column> ids;  // INT32, one per row, no validity (null == null has its own, last, id)
  std::unique_ptr<column> rep;  // INT32, one per id: the smallest row with that id
  size_type num_ids{0};
};
dense_rank_result dense_rank(column_view const& col, rmm::cuda_stream_view stream);

// Dense ids of the rows of ONE table (groupby keys): rows holding a null in any column are marked
// null in `ids` (null_policy::EXCLUDE drops them); rep[id] = first row of the id.
dense_rank_result dense_row_ids(table_view const& keys, rmm::cuda_stream_view stream);

// Encoder for hash_join: learns the id space from the BUILD table once, then maps any probe table
// with the same schema into it.  Keys of rows that cannot equal any build row come out as values
// no build key has (or null, when nulls compare unequal), so the single-key join does the rest.
class row_encoder {
 public:
  row_encoder(table_view const& build, bool nulls_equal, rmm::cuda_stream_view stream);
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
  std::vector<data_type> _types;
  std::vector<dictionary> _col_dict;   // one per column
  std::vector<dictionary> _pair_dict;  // one per inner pair level (columns - 2)
  std::unique_ptr<column> _build_keys;
};

}  // namespace detail
}  // namespace cudf
