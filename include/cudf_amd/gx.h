/*
 * gx.h -- the C ABI of libcudf_amd (MI355X / gfx950 kernel layer).
 *
 * This is the drop-in boundary for the cudf hot path: plain pointers and sizes, no C++ types,
 * no torch types.  Everything above it (include/cudf/ C++ API mirror, the Python ctypes
 * wrapper) only allocates, validates and throws; everything below it is hand-written HIP.
 *
 * Conventions (all entry points):
 *   - return 0 on success, a positive hipError_t value on a runtime failure, a negative
 *     GX_E* code on misuse;
 *   - every pointer is a DEVICE pointer unless the parameter is documented "host";
 *   - work is enqueued on `stream` and is asynchronous w.r.t. the host unless the entry point
 *     returns a host value (documented per function);
 *   - entry points never allocate: scratch comes from the caller through the cub-style query
 *     convention (tmp == NULL  ->  *tmp_bytes is set to the required size and nothing runs),
 *     mirroring the reference's call sites, e.g. cpp/src/sort/sort_radix.cu:67-71;
 *   - row counts are int64_t here; the cudf::size_type (int32) limit is enforced one layer up
 *     (cpp/include/cudf/types.hpp:76).
 *
 * Each declaration cites the reference interface it replaces (paths relative to
 * /root/reference/cpp).
 */
#ifndef CUDF_AMD_GX_H
#define CUDF_AMD_GX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* gx_stream_t; /* == hipStream_t */

/* element types: numeric values match cudf::type_id (include/cudf/types.hpp:184-216) */
enum gx_dtype {
  GX_INT8    = 1,
  GX_INT16   = 2,
  GX_INT32   = 3,
  GX_INT64   = 4,
  GX_UINT8   = 5,
  GX_UINT16  = 6,
  GX_UINT32  = 7,
  GX_UINT64  = 8,
  GX_FLOAT32 = 9,
  GX_FLOAT64 = 10,
  GX_BOOL8   = 11
};

enum gx_error {
  GX_SUCCESS       = 0,
  GX_EINVAL        = -1, /* bad argument (null pointer, negative size, aliasing) */
  GX_EDTYPE        = -2, /* dtype not supported by this entry point */
  GX_ETMP          = -3, /* scratch buffer too small */
  GX_EINTERNAL     = -4, /* device-side protocol failure reported by gx_sort_status */
  GX_EOVERFLOW     = -5  /* output does not fit the caller's capacity */
};

/* reduce / scan / groupby operator codes: match cudf::aggregation::Kind for the kinds we
 * implement (include/cudf/aggregation.hpp:84-130) */
enum gx_op {
  GX_OP_SUM         = 0,
  GX_OP_PRODUCT     = 1,
  GX_OP_MIN         = 2,
  GX_OP_MAX         = 3,
  GX_OP_COUNT_VALID = 4,
  GX_OP_COUNT_ALL   = 5,
  GX_OP_COUNT_NONZERO = 6, /* gx_reduce only: valid elements != 0 (NaN counts); cudf::reduce ANY / ALL (reductions/any.cu, all.cu) */
  GX_OP_MEAN        = 10
};

const char* gx_version(void);
/* size in bytes of one element of `dtype`, 0 if unknown */
int gx_dtype_size(int dtype);

/* ------------------------------------------------------------------------------------------
 * Radix sort.
 * Replaces cub::DeviceRadixSort::SortKeys[Descending] as called at src/sort/sort_radix.cu:69-76
 * (and the float path :80-119): stable LSD radix sort of a fixed-width column without nulls,
 * begin_bit = 0, end_bit = 8*sizeof(T).  Floats: -0.0 == +0.0 (input order kept), NaNs after
 * +Inf in input order (descending: NaNs first, reverse input order -- the composite-key rule of
 * sort_radix.cu:36-45).  `in` and `out` must not alias.
 * ------------------------------------------------------------------------------------------ */
int gx_sort_keys(int dtype, const void* in, void* out, int64_t n, int descending,
                 void* tmp, size_t* tmp_bytes, gx_stream_t stream);

/* Replaces cub::DeviceRadixSort::SortPairs[Descending] as called at
 * src/sort/sorted_order_radix.cu:83-94,125-135.  vals_in == NULL means iota (what
 * thrust::sequence writes at :70-73).  keys_out may be NULL (sorted keys are then kept only in
 * scratch, as the reference discards them: :67). */
int gx_sort_pairs(int key_dtype, const void* keys_in, void* keys_out, const int32_t* vals_in,
                  int32_t* vals_out, int64_t n, int descending, void* tmp, size_t* tmp_bytes,
                  gx_stream_t stream);

/* cudf::sorted_order / stable_sorted_order of one column (include/cudf/sorting.hpp:44-64;
 * src/sort/sort_column.cu:22-44, stable_sort_column.cu:22-46, sort_column_impl.cuh:35-57).
 * valid == NULL: radix path above.  Otherwise `valid` is an Arrow validity bitmap (LSB-first,
 * 1 = valid, bit 0 = row 0): null rows are placed first when (nulls_before XOR descending) --
 * the flip of sort_column_impl.cuh:42-45 -- in input order, valid rows are radix sorted (stable,
 * NaN greater than every number and equivalent to each other in BOTH directions: the comparator
 * path, include/cudf/detail/row_operator/common_utils.cuh:157-169).  null_count is the host-side
 * null count the column_view carries (include/cudf/column/column_view.hpp null_count()). */
int gx_sorted_order(int dtype, const void* keys, const uint32_t* valid, int64_t n, int64_t null_count,
                    int descending, int nulls_before, int32_t* out_indices, void* tmp, size_t* tmp_bytes,
                    gx_stream_t stream);

/* cudf::sorted_order / stable_sorted_order of a TABLE of 1 <= ncols <= 8 numeric columns without nulls: replaces thrust::sort /
 * thrust::stable_sort of the row indices under the lexicographic row comparator (src/sort/sort_impl.cuh:61-93; src/sort/sort.cu:22-50).
 * cols[c] / dtypes[c] / descending[c] (NULL = all ascending) are HOST arrays of ncols entries; the columns are device buffers of n rows.
 * Stable (ties of the whole tuple in row order); NaN equivalent to each other and greater than every number in both directions,
 * -0.0 == +0.0 (include/cudf/detail/row_operator/common_utils.cuh:157-169).  One keys-only word sort on a nested rank of the tuple plus
 * a pass over the runs of equal ranks (gx_order.hip); gx_sort_status(tmp) reports the word sort's status. */
int gx_sorted_order_table(int ncols, const int* dtypes, const void* const* cols, const int* descending, int64_t n, int32_t* out_indices,
                          void* tmp, size_t* tmp_bytes, gx_stream_t stream);

/* Copies the device-side status word of the last sort that used `tmp` to *status_host (0 = ok).  Synchronises `stream`.
 * 5: a look-back wait made no progress for 30 s of wall-clock time (gx_sort_set_spin_limit_ms) and was abandoned -- the output is
 * NOT sorted (every write stayed inside it); 3: a bookkeeping mismatch of the hybrid path, the LSD passes produced the output. */
int gx_sort_status(const void* tmp, int* status_host, gx_stream_t stream);
/* The same copy, queued on `stream` and NOT waited for: *status_host_pinned (host memory the device can write, e.g.
 * hipHostMalloc) holds the status once everything queued on `stream` so far has run.  The asynchronous form the
 * C++ surface uses so that cudf::sort returns without a host round trip (reference: sort.cu:52-89 is asynchronous). */
int gx_sort_status_async(const void* tmp, int* status_host_pinned, gx_stream_t stream);
/* What an abandoned look-back wait does, for the sorts the CALLING THREAD issues from now on: 0 (default) = __builtin_trap -- loud, fatal
 * for the process' HIP context: right for a caller that never reads the status word; 1 = the recoverable form: status 5, every write
 * inside the output, the caller MUST read the word (gx_sort_status / gx_sort_status_async) before it trusts the result. */
void gx_sort_set_fault_mode(int recoverable);



/* ------------------------------------------------------------------------------------------
 * The two halves of the sort, for a SHARDED sort whose exchange sits between the sort's own two partition levels (gxd_sort;
 * the reference's shape is sample -> boundaries -> shuffle -> local sort: python/cudf_polars/cudf_polars/streaming/
 * actor_graph/collectives/sort.py, with cub::DeviceRadixSort as the local sort, cpp/src/sort/sort_radix.cu:52-161).
 * INT64 / UINT64 / INT32 / UINT32 keys, ascending, keys only.  One scratch blob serves all calls of a sort: query its size with
 * gx_sortx_sample(tmp = NULL); n = this rank's rows, recv_rows_max = the most rows it is prepared to receive.
 *   gx_sortx_sample   varying-bit masks of a sample of the shard            -> gx_sortx_masks reads {OR, NOR} (sortable form)
 *   gx_sortx_level0   level 0 (256 bins on the 8 bits below the highest bit that varies in masks2 = the OR over ALL ranks of
 *                     what gx_sortx_masks returned) into padded (bin, input range) slots of the level-0 buffer
 *   gx_sortx_tables   cur0[r * 256 + bin] = keys in slot (range r, bin), slot0[...] = its first key in the level-0 buffer (slots
 *                     are laid out bin-major: the 8 slots of a bin are neighbours, bins ascend), slot_total = rows in use,
 *                     state = 3 when level 0 succeeded (anything else: take another path)
 *   gx_sortx_level0_buffer   the level-0 buffer: own_rows keys of this rank's slots, the rest (up to total_rows) is the
 *                     receive area the caller fills with the spans its peers send
 *   gx_sortx_finish   level 1 + cell sort over `nreg` regions {first key in the level-0 buffer, keys, level-0 bucket}, given in
 *                     bucket order; n = their total; `out` receives the n sorted keys
 *   gx_sortx_status   1 when the sorted output is valid (0: a device-side check failed -- take another path)
 * masks, tables, status synchronise the stream (they return host values). */
int gx_sortx_sample(int dtype, const void* keys, int64_t n, int64_t recv_rows_max, void* tmp, size_t* tmp_bytes, gx_stream_t stream);
int gx_sortx_masks(const void* tmp, uint64_t* masks2_host, gx_stream_t stream);
int gx_sortx_level0(int dtype, const void* keys, int64_t n, int64_t recv_rows_max, const uint64_t* masks2_host, void* tmp, gx_stream_t stream);
int gx_sortx_tables(const void* tmp, uint32_t* cur0_host, uint32_t* slot0_host, uint32_t* slot_total_host, int32_t* state_host, gx_stream_t stream);
void* gx_sortx_level0_buffer(int dtype, void* tmp, int64_t n, int64_t recv_rows_max, int64_t* own_rows, int64_t* total_rows);
int gx_sortx_finish(int dtype, int64_t n_send, int64_t recv_rows_max, int64_t n, const uint64_t* masks2_host, const uint32_t* reg_start_host,
                    const uint32_t* reg_count_host, const uint32_t* reg_bucket_host, int nreg, void* out, void* tmp, gx_stream_t stream);
int gx_sortx_status(const void* tmp, int32_t* ok_host, gx_stream_t stream);

/* info8_host (host, 8 x int32) = {hybrid attempted, hybrid used, d1, shift2, bits2, LDS passes,
 * largest cell, active LSD passes (-1 when the hybrid path produced the output)} of the last sort
 * that used `tmp`.  Synchronises `stream`. */
int gx_sort_info(const void* tmp, int32_t* info8_host, gx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Gather.  Replaces cudf::detail::gather for fixed-width columns
 * (include/cudf/detail/gather.cuh:108-131 data, :506-577 validity): out[i] = in[map[i]].
 * elem_size in {1,2,4,8}.  map entries must be in [0, src_rows) (bounds_policy::DONT_CHECK);
 * entries equal to INT32_MIN (JoinNoMatch) produce a null/zero row when nullify_oob != 0.
 * src_valid/out_valid may be NULL (no validity handled).
 * ------------------------------------------------------------------------------------------ */
int gx_gather(int elem_size, const void* src, const uint32_t* src_valid, int64_t src_rows,
              const int32_t* map, int64_t n, int nullify_oob, void* out, uint32_t* out_valid,
              gx_stream_t stream);
/* Sharded joins: out[j] = rows[idx[j]] + seg_bases[s], s = the segment of the receive buffer position idx[j] falls
 * into (segment k holds seg_counts[k] entries; nseg <= 16 ranks; counts and bases are HOST arrays).  Turns the
 * (int32 local row) a rank received next to each key into the global int64 row id of a join pair in one gather
 * (the per-rank split of cudf_polars' shuffle carries the same information as a partition id). */
int gx_gather_global_rows(const int32_t* rows, int64_t nrows, const int32_t* idx, int64_t n, int nseg,
                          const int64_t* seg_counts_host, const int64_t* seg_bases_host, int64_t* out, gx_stream_t stream);

/* The same with the segment table in DEVICE memory and any number of segments (the chunked exchange of gxd.h leaves
 * chunks x ranks segments): segtab_dev = [nseg + 1] int64 segment starts followed by [nseg] int64 bases. */
int gx_gather_global_rows_dev(const int32_t* rows, int64_t nrows, const int32_t* idx, int64_t n, int nseg, const int64_t* segtab_dev,
                              int64_t* out, gx_stream_t stream);
/* The sharded join's pairs without a gather: a row travels as enc = (source rank << shift) | row (the payload of the *_pl entry
 * points below) and is decoded in one streaming pass: out[j] = bases_dev[src] + row, plus chunk(j) * chunk_rows_dev[src] when
 * the row was counted inside its sender's CHUNK -- chunk(j) = the c with snap_dev[c] <= j < snap_dev[c + 1], snap_dev = the
 * pair positions at which the probe of each received chunk started (nchunks + 1 device int64; nchunks <= 1024). */
int gx_decode_global_rows(const int32_t* enc, int64_t n, int shift, const int64_t* bases_dev, const int64_t* chunk_rows_dev,
                          const int64_t* snap_dev, int nchunks, int64_t* out, gx_stream_t stream);
/* out[i] = in[i] as int64 (row indices / counts leaving the int32 world of cudf::size_type) */
int gx_widen_i32_i64(const int32_t* in, int64_t n, int64_t* out, gx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Validity bitmaps.  Replace src/bitmask/null_mask.cu:152 (set), :339-409 (count),
 * include/cudf/detail/null_mask.cuh:68 (bitmask_and).
 * ------------------------------------------------------------------------------------------ */
int gx_bitmask_set(uint32_t* mask, int64_t begin_bit, int64_t end_bit, int valid, gx_stream_t stream);
/* *count_dev (device int64) = number of set bits in [begin_bit, end_bit) */
int gx_bitmask_count(const uint32_t* mask, int64_t begin_bit, int64_t end_bit, int64_t* count_dev,
                     gx_stream_t stream);
/* dst bits [dst_begin_bit, +nbits) = src bits [src_begin_bit, +nbits); src == NULL writes 1s.  The
 * building block of cudf::copy_bitmask (src/bitmask/null_mask.cu:357-390) and concatenate_masks
 * (src/copying/concatenate.cu:111-170); bits of dst outside the range are preserved. */
int gx_bitmask_copy(uint32_t* dst, int64_t dst_begin_bit, const uint32_t* src, int64_t src_begin_bit,
                    int64_t nbits, gx_stream_t stream);
/* out = AND of `nmasks` bitmaps (host array of device pointers; NULL entries = all valid),
 * nbits bits each; *count_dev (optional) = set bits of the result */
int gx_bitmask_and(const uint32_t* const* masks_host, int nmasks, int64_t nbits, uint32_t* out,
                   int64_t* count_dev, gx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Hashing / partitioning.
 * gx_murmur3_32: cudf::hashing::detail::MurmurHash3_x86_32<T>
 * (include/cudf/hashing/detail/murmurhash3_x86_32.cuh:22-67), null -> UINT32_MAX, and the
 * column fold of primitive_row_operators.cuh:247-268: combine == 0 writes out[i] = hash(x[i]),
 * combine != 0 writes out[i] = hash_combine(out[i], hash(x[i])) (hashing.hpp:83-86).
 * ------------------------------------------------------------------------------------------ */
int gx_murmur3_32(int dtype, const void* in, const uint32_t* valid, int64_t n, uint32_t seed,
                  int combine, uint32_t* out, gx_stream_t stream);

/* IdentityHash<T> of cudf::hash_partition(..., hash_id::HASH_IDENTITY) (src/partitioning/partitioning.cu:852-872): the element
 * cast to uint32 (static_cast<uint32_t>(key); floating point truncated toward zero, NaN / negative -> 0, >= 2^32 -> UINT32_MAX as
 * the device conversion does), null -> UINT32_MAX, the same column fold as gx_murmur3_32 (combine).  Also what turns a
 * cudf::partition map of any integral type (partitioning.cu:780-842, is_index_type) into the uint32 ids gx_hash_partition_map takes. */
int gx_identity_hash_32(int dtype, const void* in, const uint32_t* valid, int64_t n, int combine,
                        uint32_t* out, gx_stream_t stream);

/* cudf::hash_partition (include/cudf/partitioning.hpp:103-110; src/partitioning/partitioning.cu:
 * 53-92,120-360,568-660) reduced to its index form: from row hashes, a stable gather map that
 * groups rows by partition (hash % num_partitions) plus num_partitions+1 int32 offsets.
 * The caller gathers every column through the map (gx_gather). */
int gx_hash_partition_map(const uint32_t* row_hash, int64_t n, int num_partitions,
                          int32_t* out_map, int32_t* out_offsets, void* tmp, size_t* tmp_bytes,
                          gx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Row-key encoding for multi-column join / groupby keys.  The reference hashes and compares whole
 * rows inside its tables (include/cudf/detail/row_operator/primitive_row_operators.cuh:95-163,
 * 207-274); here a row is encoded into ONE fixed-width key first (cf. the reference's own
 * include/cudf/join/key_remapping.hpp) and the single-key kernels below do the rest.
 *
 * gx_pack_keys: out[i] = the ncols (<= 8) column values of row i concatenated into a uint64 (column 0
 *   most significant; widths must sum to <= 8 bytes, else GX_EINVAL).  FLOAT32/64 columns are
 *   normalised (-0.0 -> +0.0, every NaN -> one NaN): the row comparator's equality classes
 *   (detail/row_operator/common_utils.cuh:215-220).  `cols` / `dtypes` are HOST arrays.
 * gx_dense_rank: out_ids[i] in [0, G) with equal values sharing an id (null == null has its own id,
 *   floats compared as above), ids ascending with the value, nulls last; out_rep[g] (optional, n
 *   entries) = the smallest row of id g; *out_ngroups_dev = G.  sorted_order + adjacent difference +
 *   scan + scatter.  cub-style scratch query.
 * ------------------------------------------------------------------------------------------ */
int gx_pack_keys(int ncols, const void* const* cols, const int* dtypes, int64_t n, uint64_t* out,
                 gx_stream_t stream);
/* gx_unpack_keys: the inverse of gx_pack_keys -- out_cols[k][i] = column k of packed[i] (float columns come back
 * normalised: +0.0 for either zero, one NaN).  Turns the distinct packed keys of a groupby back into key columns. */
int gx_unpack_keys(int ncols, void* const* out_cols, const int* dtypes, int64_t n, const uint64_t* packed,
                   gx_stream_t stream);
/* Rows wider than 8 bytes: hash-and-verify.  gx_hash_rows64: out[i] = a 64-bit hash of the ncols (<= 8) column
 *   values of row i (floats normalised as in gx_pack_keys; equal rows hash equal); the single-key join / groupby
 *   kernels then run on the hashes, one pass over the key columns instead of one radix sort per column -- the
 *   reference hashes the row once too (detail/row_operator/primitive_row_operators.cuh:247-268).
 * gx_rows_mismatch_count: *mismatch_dev = the number of pairs (lidx[i], ridx[i]) -- NULL index array = row i
 *   itself, a negative index = no row, skipped -- whose rows differ in some column (the row equality the reference
 *   evaluates inside its probe, primitive_row_operators.cuh:95-163).  0 certifies a result obtained through the
 *   hashes as exact; otherwise (a 64-bit collision) the caller re-runs through gx_dense_rank. */
int gx_hash_rows64(int ncols, const void* const* cols, const int* dtypes, int64_t n, uint64_t seed, uint64_t* out,
                   gx_stream_t stream);
int gx_rows_mismatch_count(int ncols, const void* const* lcols, const void* const* rcols, const int* dtypes,
                           const int32_t* lidx, const int32_t* ridx, int64_t npairs, int64_t* mismatch_dev,
                           gx_stream_t stream);
int gx_dense_rank(int dtype, const void* keys, const uint32_t* valid, int64_t n, int64_t null_count,
                  int32_t* out_ids, int32_t* out_rep, int64_t* out_ngroups_dev, void* tmp,
                  size_t* tmp_bytes, gx_stream_t stream);
/* data[i] = value_bits (low elem_size bytes) for every row whose validity bit is 0 -- cudf::replace_nulls
 * with a scalar (src/replace/nulls.cu), in place; valid == NULL: no-op. */
int gx_fill_nulls(int elem_size, void* data, const uint32_t* valid, int64_t n, uint64_t value_bits,
                  gx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Hash join (single fixed-width key column of 4 or 8 bytes; multi-column keys are packed or
 * hashed by the layer above -- see include/cudf/join/).
 * Replaces cudf::detail::hash_join build (src/join/hash_join/hash_join.cu:62-99,112-148 ->
 * cuco::static_multiset::insert) and probe (src/join/hash_join/retrieve_impl.cuh:28-113,
 * size_impl.cuh:26-62 -> cuco count / retrieve).
 *
 * The table is an open-addressing multiset of {key, row} slots followed by one 4-bit tag per slot
 * (0 = empty, else hash bits; the partitioned probe walks chains on the tags in LDS), sized by
 * gx_join_table_bytes = 256 + slots * (slot bytes + 1/2);
 * build rows whose validity bit is 0 are skipped (null_equality::UNEQUAL semantics of
 * hash_join.cu:77-84; for EQUAL the caller maps nulls to a reserved key -- see DESIGN.md).
 * ------------------------------------------------------------------------------------------ */
size_t gx_join_table_bytes(int key_size, int64_t build_rows, double load_factor);
int gx_join_build(int key_size, const void* build_keys, const uint32_t* build_valid,
                  int64_t build_rows, void* table, size_t table_bytes, double load_factor,
                  gx_stream_t stream);
/* Number of matches: *count_dev (device int64).  probe_valid NULL = all valid. */
int gx_join_count(int key_size, const void* probe_keys, const uint32_t* probe_valid,
                  int64_t probe_rows, const void* table, size_t table_bytes, int64_t* count_dev,
                  gx_stream_t stream);
/* Emits pairs (probe_idx, build_idx) in unspecified order (include/cudf/join/join.hpp:131-134)
 * through a device cursor *cursor_dev (must be zeroed by the caller); pairs beyond `capacity`
 * are counted but not written (caller compares the cursor with capacity).
 * left_outer bit 0: probe rows without a match emit (probe_idx, INT32_MIN) -- cudf::left_join
 * (src/join/join.cu:62-85).  left_outer bit 1 (with bit 0): NULL probe rows emit nothing -- under
 * null_equality::EQUAL with null build rows present their partners are those rows, which the caller appends
 * (hash_join.cu:77-84). */
int gx_join_probe(int key_size, const void* probe_keys, const uint32_t* probe_valid,
                  int64_t probe_rows, const void* table, size_t table_bytes, int left_outer,
                  int32_t* out_probe_idx, int32_t* out_build_idx, int64_t capacity,
                  int64_t* cursor_dev, gx_stream_t stream);

/* Partitioned form of gx_join_probe for large probes (no probe nulls) against tables far beyond the
 * L2s: the probe rows are radix-partitioned on the table's top hash bits (one streaming pass), then
 * probed partition by partition with XCD-affine scheduling so that each ~2 MiB sub-table is L2
 * resident while it is probed.  Same outputs and cursor convention as gx_join_probe (pair order
 * unspecified).  cub-style scratch query.  GX_EINVAL when the table is too small to partition
 * (gx_join_partition_bits() == 0): use gx_join_probe. */
int gx_join_probe_partitioned(int key_size, const void* probe_keys, int64_t probe_rows, const void* table,
                              size_t table_bytes, int left_outer, int32_t* out_probe_idx,
                              int32_t* out_build_idx, int64_t capacity, int64_t* cursor_dev, void* tmp,
                              size_t* tmp_bytes, gx_stream_t stream);
/* Partitioned form of gx_join_build for large build sides without nulls: the rows are radix-
 * partitioned on the table's top hash bits, then inserted partition by partition so that the CAS
 * and slot writes of concurrently running workgroups fall into one L2-resident ~2 MiB sub-table.
 * Produces the same table as gx_join_build (slot positions may differ among equal hashes).
 * cub-style scratch query.  GX_EINVAL when the table is too small to partition. */
int gx_join_build_partitioned(int key_size, const void* build_keys, int64_t build_rows, void* table,
                              size_t table_bytes, double load_factor, void* tmp, size_t* tmp_bytes,
                              gx_stream_t stream);
/* Matches per probe row: counts[i] = max(min_count, number of build rows whose key equals probe key i); null probe
 * rows count 0.  The per-row form of gx_join_count behind cudf::hash_join::*_join_match_context
 * (cpp/include/cudf/join/hash_join.hpp:259-340; cpp/src/join/hash_join/size_impl.cuh:26-62). */
int gx_join_count_rows(int key_size, const void* probe_keys, const uint32_t* probe_valid, int64_t probe_rows,
                       const void* table, size_t table_bytes, int32_t min_count, int32_t* counts, gx_stream_t stream);
/* data[i] += value wherever data[i] != INT32_MIN (JoinNoMatch): re-bases the probe indices of a chunk of a
 * partitioned join (hash_join.hpp:352-440) onto the whole left table. */
int gx_add_i32(int32_t* data, int64_t n, int32_t value, gx_stream_t stream);
/* One-pass partition of rows into `nparts` <= 16 contiguous groups (histogram + LDS-ranked scatter, the partition pass
 * of the partitioned join): out_keys = the keys grouped by destination (order inside a group unspecified), out_rows
 * (optional) = their row indices, offsets_dev[nparts + 1] = group starts (device int64).  What a rank runs before the
 * all-to-all of the distributed operators (cudf::hash_partition, cpp/src/partitioning/partitioning.cu:568-660, feeding
 * the shuffle of cpp/libcudf_streaming/src/partition_utils.cpp:72-117):
 *   mode 0  destination = the top 32 bits of a multiplicative hash independent of the join table's slot bits, scaled into
 *           [0, nparts) by a multiply-shift (any nparts; for a power of two these are the hash's top bits);
 *           key_dtype any 4- or 8-byte type (bit patterns are hashed);
 *   mode 1  destination = number of splitters <= key in cudf sort order; splitters_host = nparts - 1 ascending keys
 *           of key_dtype in HOST memory (INT32/UINT32/FLOAT32/INT64/UINT64/FLOAT64).
 * cub-style scratch query. */
int gx_partition_rows(int key_dtype, const void* keys, int64_t n, int mode, int nparts, const void* splitters_host,
                      void* out_keys, int32_t* out_rows, int64_t* offsets_dev, void* tmp, size_t* tmp_bytes,
                      gx_stream_t stream);
/* The same pass over a CHUNK of a shard: out_rows[i] = row_base + (index inside `keys`), so that chunks of one column can be
 * partitioned (and exchanged) one after the other while the row ids stay those of the whole shard. */
int gx_partition_rows_at(int key_dtype, const void* keys, int64_t n, int32_t row_base, int mode, int nparts, const void* splitters_host,
                         void* out_keys, int32_t* out_rows, int64_t* offsets_dev, void* tmp, size_t* tmp_bytes,
                         gx_stream_t stream);
/* The SPECULATIVE form (cap_rows > 0): no histogram pass.  Group g owns the fixed slot [g, g + 1) * cap_rows of out_keys /
 * out_rows (which hold nparts * cap_rows elements); offsets_dev[g] receives the ROWS of group g (not a start), offsets_dev[nparts]
 * is non-zero when some group outgrew its slot -- its surplus rows were dropped and the caller must partition this input again
 * with the exact form.  What the sharded operators run per chunk before an exchange: one read of the keys instead of two. */
int gx_partition_rows_spec_at(int key_dtype, const void* keys, int64_t n, int32_t row_base, int mode, int nparts, const void* splitters_host,
                              int64_t cap_rows, void* out_keys, int32_t* out_rows, int64_t* offsets_dev, void* tmp, size_t* tmp_bytes,
                              gx_stream_t stream);
/* gx_join_probe_partitioned for a chunk of the probe side: the probe indices written are row_base + (index inside probe_keys);
 * pairs are appended at *cursor_dev (which the caller zeroes once, before the first chunk). */
int gx_join_probe_partitioned_at(int key_size, const void* probe_keys, int64_t probe_rows, int32_t row_base, const void* table,
                                 size_t table_bytes, int left_outer, int32_t* out_probe_idx, int32_t* out_build_idx,
                                 int64_t capacity, int64_t* cursor_dev, void* tmp, size_t* tmp_bytes, gx_stream_t stream);
/* PAYLOAD forms: a build / probe row is represented by payload[i] (int32 >= 0) instead of its row number i -- what the table
 * slots store and what the pair arrays receive.  The sharded join puts encoded global rows there (gx_decode_global_rows). */
int gx_join_build_pl(int key_size, const void* build_keys, const int32_t* payload, const uint32_t* build_valid, int64_t build_rows,
                     void* table, size_t table_bytes, double load_factor, gx_stream_t stream);
int gx_join_build_partitioned_pl(int key_size, const void* build_keys, const int32_t* payload, int64_t build_rows, void* table,
                                 size_t table_bytes, double load_factor, void* tmp, size_t* tmp_bytes, gx_stream_t stream);
int gx_join_probe_partitioned_pl(int key_size, const void* probe_keys, const int32_t* payload, int64_t probe_rows, int32_t row_base,
                                 const void* table, size_t table_bytes, int left_outer, int32_t* out_probe_idx, int32_t* out_build_idx,
                                 int64_t capacity, int64_t* cursor_dev, void* tmp, size_t* tmp_bytes, gx_stream_t stream);
/* log2 of the number of partitions the partitioned probe uses for this table; 0 = not partitionable */
int gx_join_partition_bits(int key_size, size_t table_bytes);

/* out_build_idx[i] = the first build row whose key equals probe key i, or INT32_MIN (JoinNoMatch) --
 * the left join against DISTINCT build keys, in probe order and without an output reservation:
 * cudf::distinct_hash_join::left_join (include/cudf/join/distinct_hash_join.hpp:111-116,
 * src/join/distinct_hash_join.cu).  Null probe rows (validity bit 0) get JoinNoMatch. */
int gx_join_lookup(int key_size, const void* probe_keys, const uint32_t* probe_valid, int64_t probe_rows,
                   const void* table, size_t table_bytes, int32_t* out_build_idx, gx_stream_t stream);
/* Left semi (anti == 0) / left anti (anti != 0) join: the ASCENDING list of probe rows that have
 * (have no) match in the table, *count_dev = its length -- the contains map + stable copy_if of
 * cudf::filtered_join::semi_join / anti_join (src/join/filtered_join/filtered_join.cu:124-156).
 * Null probe rows count as matching iff null_matches != 0 (null_equality::EQUAL and the build side
 * holds a null).  out_probe_idx needs room for probe_rows entries.  cub-style scratch query. */
int gx_join_filter(int key_size, const void* probe_keys, const uint32_t* probe_valid, int64_t probe_rows,
                   const void* table, size_t table_bytes, int anti, int null_matches,
                   int32_t* out_probe_idx, int64_t* count_dev, void* tmp, size_t* tmp_bytes,
                   gx_stream_t stream);

/* cudf::full_join's complement step (src/join/join_utils.cu:86-157): appends (JoinNoMatch, r) for
 * every build row r in [0, build_rows) that does not occur in build_idx[0..n) to the pair arrays,
 * starting at *cursor_dev (device int64: pairs already present, updated to the new total; pairs
 * beyond `capacity` are counted but not written).  tmp: cub-style query. */
int gx_join_complement(const int32_t* build_idx, int64_t n, int64_t build_rows, int32_t* out_probe_idx,
                       int32_t* out_build_idx, int64_t capacity, int64_t* cursor_dev, void* tmp,
                       size_t* tmp_bytes, gx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Groupby, hash path (single int32/int64 key column; values int32/int64/float32/float64).
 * Replaces src/groupby/hash/compute_groupby.cu:50-155 + compute_global_memory_aggs.cuh:123-157:
 * distinct keys (unspecified order) and per-group SUM / COUNT_VALID / COUNT_ALL / MIN / MAX.
 * float sums are accumulated in 128-bit fixed point (exact, order independent, bit-reproducible)
 * and rounded once: see DESIGN.md "groupby".  Outputs have capacity `max_groups`; *ngroups_dev
 * (device int64) receives the group count.  Rows with key validity 0 are dropped
 * (null_policy::EXCLUDE, compute_groupby.cu:62-66).
 * ------------------------------------------------------------------------------------------ */
int gx_groupby_sum_count(int key_dtype, const void* keys, const uint32_t* keys_valid,
                         int val_dtype, const void* vals, const uint32_t* vals_valid, int64_t n,
                         int64_t max_groups, void* out_keys, void* out_sum /* f64 or i64 */,
                         int32_t* out_count_valid, int32_t* out_count_all, int64_t* ngroups_dev,
                         void* tmp, size_t* tmp_bytes, gx_stream_t stream);

/* Groupby SUM + COUNT on SEVERAL key columns, rows compared inside the table (the reference hashes a row once and
 * compares rows in its probe: include/cudf/detail/row_operator/primitive_row_operators.cuh:95-163, 247-268).
 * key_cols: nkeys (2..4) HOST array of device pointers to columns of 8-byte words (int64 / uint64 columns as they are;
 * narrower integer columns widened by the caller); out_key_cols: nkeys device columns of capacity max_groups that
 * receive the groups' key words.  No nulls (nullable inputs take gx_hash_rows64 + gx_groupby_sum_count).  Groups come
 * out in unspecified order.  *ngroups_dev: the group count; -1: more than max_groups groups (retry with a larger bound);
 * -2: keys skewed beyond the partition slots / more groups per partition than an LDS table holds -- the caller takes the
 * single-key path over row hashes instead (nothing usable was written). */
int gx_groupby_sum_count_wide(int nkeys, const void* const* key_cols, int val_dtype, const void* vals, int64_t n,
                              int64_t max_groups, void* const* out_key_cols, void* out_sum /* f64 or i64 */,
                              int32_t* out_count, int64_t* ngroups_dev, void* tmp, size_t* tmp_bytes, gx_stream_t stream);

/* Groupby MIN / MAX of one value column (src/groupby/hash/global_memory_aggregator.cuh:18-238):
 * same conventions as gx_groupby_sum_count; out_min / out_max have the VALUE dtype (either may be
 * NULL); a group without a valid value has count 0 and an unspecified min/max (the caller nulls it).
 * Floats: -0.0 == +0.0, NaN greater than every number (the row comparator's order). */
int gx_groupby_min_max(int key_dtype, const void* keys, const uint32_t* keys_valid, int val_dtype,
                       const void* vals, const uint32_t* vals_valid, int64_t n, int64_t max_groups,
                       void* out_keys, void* out_min, void* out_max, int32_t* out_count_valid,
                       int64_t* ngroups_dev, void* tmp, size_t* tmp_bytes, gx_stream_t stream);

/* Compound groupby aggregations on top of gx_groupby_sum_count / gx_groupby_min_max.
 * gx_square: out[i] = in[i]^2 in the SUM accumulator type of `dtype` (integers -> int64 wrapping mod
 *   2^64, FLOAT32/64 keep their type): the values of the reference's single-pass SUM_OF_SQUARES
 *   (include/cudf/detail/aggregation/aggregation.hpp:944-954), summed by gx_groupby_sum_count.
 * gx_var_from_sums: mode 0 M2 = sum_sqr - sum^2/count (valid for every group), 1 VARIANCE = M2/(count-ddof),
 *   2 STD = sqrt(VARIANCE); null where count == 0 or count - ddof <= 0
 *   (src/groupby/common/m2_var_std.cu:44-61,153-190; hash_compound_agg_finalizer.cu:135-186).
 *   sum_dtype GX_INT64 / GX_FLOAT64 / GX_FLOAT32; mask_out holds ceil(n/32) words; *null_count_dev = nulls.
 * gx_groupby_arg_select: ARGMIN / ARGMAX (global_memory_aggregator.cuh: ARGMIN/ARGMAX updates): out_rows[g] =
 *   the smallest row i with group_of_row[i] == g, a valid value and vals[i] == target[g] (target = the
 *   group's MIN or MAX; floats NaN == NaN); 0x7F7F7F7F where the group has no valid value. */
int gx_square(int dtype, const void* in, int64_t n, void* out, gx_stream_t stream);
int gx_var_from_sums(int sum_dtype, const void* sum_sqr, const void* sum, const int32_t* count, int64_t n,
                     int ddof, int mode, double* out, uint32_t* mask_out, int64_t* null_count_dev,
                     gx_stream_t stream);
int gx_groupby_arg_select(int val_dtype, const void* vals, const uint32_t* vals_valid,
                          const int32_t* group_of_row, int64_t n, const void* target, int64_t num_groups,
                          int32_t* out_rows, gx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Sort-based groupby building blocks: rows that are (or have been brought, through `order`) in key order.
 * Replace compute_group_offsets / label_segments (cpp/src/groupby/sort/sort_helper.cu:151-214),
 * thrust::reduce_by_key of the sort-path aggregations (sort/group_single_pass_reduction_util.cuh:133-200,
 * sort/group_count.cu:25-89), segmented_shift (groupby.cu:306-346), group_replace_nulls
 * (sort/group_replace_nulls.cu) and the rank scans of cpp/src/sort/rank.cu:60-330.
 *
 * gx_group_heads: heads[i] = 1 iff row order[i] differs from row order[i - 1] in this column (order NULL =
 *   identity; heads[0] = 1; null == null, NaN == NaN, -0.0 == +0.0); combine != 0 ORs into `heads`, so a key
 *   TABLE is one call per column.
 * gx_group_offsets: labels[i] = group of sorted row i, offsets[g] = first sorted row of group g,
 *   offsets[ngroups] = n (capacity n + 1), sizes (optional, capacity n) = rows per group; *ngroups_dev.
 * gx_segmented_reduce: one value per group at out[label] -- op GX_OP_SUM / PRODUCT (integers -> INT64, floats
 *   keep their type; float SUM in double-double: <= 1 ulp, order-independent), MIN / MAX (input type),
 *   COUNT_VALID (out NULL); out_count_valid (optional) = valid values per group.  `vals` are in sorted order.
 * gx_segmented_shift: out[i] = in[i - offset] when that row is in the same group, else the fill value
 *   (fill_valid == 0: null); out_valid (optional) holds ceil(n/64)*2 words.
 * gx_segmented_fill_nulls: replace_policy PRECEDING (backward == 0) / FOLLOWING: a null takes the nearest valid
 *   value of its group before / after it, or stays null.
 * gx_rank_from_groups: scatter ranks to out[order[i]]: method 0 FIRST, 1 AVERAGE, 2 MIN, 3 MAX, 4 DENSE
 *   (rank_method, cpp/include/cudf/aggregation.hpp:91-98); scale > 0 = percentage (rank / scale, or
 *   (rank - 1) / (scale - 1) when one_normalized); exactly one of out_i32 / out_f64.
 * ------------------------------------------------------------------------------------------ */
int gx_group_heads(int dtype, const void* col, const uint32_t* valid, const int32_t* order, int64_t n, int combine,
                   uint8_t* heads, gx_stream_t stream);
int gx_group_offsets(const uint8_t* heads, int64_t n, int32_t* labels, int32_t* offsets, int32_t* sizes,
                     int64_t* ngroups_dev, void* tmp, size_t* tmp_bytes, gx_stream_t stream);
int gx_segmented_reduce(int val_dtype, const void* vals, const uint32_t* vals_valid, const uint8_t* heads,
                        const int32_t* labels, int64_t n, int op, void* out, int32_t* out_count_valid, void* tmp,
                        size_t* tmp_bytes, gx_stream_t stream);
int gx_segmented_shift(int elem_size, const void* in, const uint32_t* in_valid, const int32_t* labels, int64_t n,
                       int64_t offset, uint64_t fill_bits, int fill_valid, void* out, uint32_t* out_valid,
                       gx_stream_t stream);
int gx_segmented_fill_nulls(int elem_size, const void* in, const uint32_t* in_valid, const uint8_t* heads, int64_t n,
                            int backward, void* out, uint32_t* out_valid, void* tmp, size_t* tmp_bytes,
                            gx_stream_t stream);
int gx_rank_from_groups(const int32_t* order, const int32_t* labels, const int32_t* offsets, int64_t n, int method,
                        double scale, int one_normalized, int32_t* out_i32, double* out_f64, gx_stream_t stream);
/* Segment id of every row for cudf::segmented_sorted_order (cpp/src/sort/segmented_sort_impl.cuh:178-203): rows of
 * segment [offsets[j], offsets[j+1]) get offsets[j+1]; rows outside every segment get unique ascending ids. */
int gx_segment_ids(const int32_t* offsets, int64_t num_offsets, int64_t num_rows, int32_t* ids, gx_stream_t stream);


/* Result finalizers of cudf::groupby::aggregate: (a) validity bitmap of SUM / MEAN results from
 * COUNT_VALID -- a group without a valid value is null (src/groupby/hash/output_utils.cu:68-70);
 * *null_count_dev (device int64) receives the number of null groups; mask_out needs
 * (n + 31) / 32 words.  (b) MEAN = SUM / COUNT_VALID computed in double
 * (src/groupby/hash/hash_compound_agg_finalizer.cu:92-133); sum_dtype in {INT64, FLOAT64, FLOAT32}. */
int gx_valid_from_counts(const int32_t* counts, int64_t n, uint32_t* mask_out, int64_t* null_count_dev,
                         gx_stream_t stream);
int gx_mean_from_sum(int sum_dtype, const void* sum, const int32_t* count, int64_t n, double* out,
                     gx_stream_t stream);

/* Segmented inclusive scan over sorted group labels: replaces thrust::inclusive_scan_by_key at
 * src/groupby/sort/group_scan_util.cuh:109-130 (groupby::scan SUM/MIN/MAX).  keys are the
 * (already sorted) key column; out[i] = op over the rows of the same key run up to i.
 * Null values (vals_valid bit 0) contribute the identity. Integer SUM accumulates in int64.
 * op GX_OP_COUNT_VALID / GX_OP_COUNT_ALL (sort/group_count_scan.cu:24-62): out is INT32, out[i] = valid rows (all
 * rows) of the run up to and including i; `vals` is not read and may be NULL. */
int gx_segmented_scan(int key_dtype, const void* sorted_keys, int val_dtype, const void* vals,
                      const uint32_t* vals_valid, int64_t n, int op, void* out, void* tmp,
                      size_t* tmp_bytes, gx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Column reduce / scan.
 * gx_reduce replaces cub::DeviceReduce::Reduce at include/cudf/reduction/detail/reduction.cuh:
 * 46-83 (cudf::reduce SUM/MIN/MAX/PRODUCT; nulls skipped).  Result written to *out_dev in
 * `out_dtype` (GX_INT64, GX_UINT64 or GX_FLOAT64 for SUM/PRODUCT; in_dtype for MIN/MAX; an integer type for
 * GX_OP_COUNT_NONZERO, the number of valid elements != 0 -- ANY = count > 0, ALL = count == valid count);
 * *valid_count_dev (device int64, optional) = number of valid inputs.
 * gx_scan replaces thrust::inclusive_scan / exclusive_scan at
 * src/reductions/scan/scan_inclusive.cu:76-89 and scan_exclusive.cu; output dtype == input
 * dtype (ints wrap); nulls contribute the identity (null_policy handling of the mask is done by
 * the caller: scan_inclusive.cu:198-216).
 * ------------------------------------------------------------------------------------------ */
int gx_reduce(int in_dtype, const void* in, const uint32_t* valid, int64_t n, int op,
              int out_dtype, void* out_dev, int64_t* valid_count_dev, void* tmp, size_t* tmp_bytes,
              gx_stream_t stream);
int gx_scan(int dtype, const void* in, const uint32_t* valid, int64_t n, int op, int inclusive,
            void* out, void* tmp, size_t* tmp_bytes, gx_stream_t stream);

/* index of the first 0 bit in [0,nbits) or nbits: *pos_dev (device int64).  Used for the
 * null_policy::INCLUDE scan mask (scan_inclusive.cu:44-53 thrust::find_if_not). */
int gx_bitmask_first_unset(const uint32_t* mask, int64_t nbits, int64_t* pos_dev, gx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Synthetic data (bench / tests): counter-based generator so that 1e9-row inputs never cross
 * PCIe.  out[i] = splitmix64(seed + i) mapped to the dtype; `lo`,`hi` bound integer outputs
 * when hi > lo (uniform in [lo, hi)), floats are uniform in [0,1).
 * ------------------------------------------------------------------------------------------ */
int gx_fill_random(int dtype, void* out, int64_t n, uint64_t seed, int64_t lo, int64_t hi,
                   gx_stream_t stream);
/* data[i] = mix64(data[i]) in place, mix64 = the splitmix64 FINALIZER (a bijection of the 64-bit integers): ids j in [0, m)
 * become m distinct keys that look random in every bit -- the join benchmarks' key sets (distinct random build keys,
 * a probe side that hits them with a chosen probability; cpp/benchmarks/join/generate_input_tables.cu:24-103). */
int gx_mix64_inplace(uint64_t* data, int64_t n, gx_stream_t stream);
/* device-to-device copy as a KERNEL on `stream` (16-byte lanes): what a rank does with the part of an exchange that stays
 * on it.  hipMemcpyAsync picks the SDMA engines when other queues are busy -- 32 GB/s for an intra-device copy on this
 * part, measured (profiles/r3_xp_distributed_single_rank.txt: 252 ms for 8 GB in chunks, 4 ms as one kernel). */
int gx_copy_bytes(const void* src, void* dst, size_t bytes, gx_stream_t stream);
/* iota: out[i] = start + i (int32) */
int gx_sequence_i32(int32_t* out, int64_t n, int32_t start, gx_stream_t stream);
/* order-independent 64-bit checksum of a column (sum and xor of splitmix64(element)) and a
 * sortedness violation count in the column's cudf order: res_dev[0]=sum, [1]=xor, [2]=violations */
int gx_checksum(int dtype, const void* in, int64_t n, int descending, uint64_t* res_dev,
                gx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CUDF_AMD_GX_H */
