/*
 * gxd.h -- C ABI of the SHARDED operators (one process per GPU, RCCL over xGMI): distributed cudf::sort,
 * cudf::hash_join (build once / probe many) and groupby SUM + COUNT.  SURVEY.md 8(e); BASELINE config 5.
 *
 * What it replaces in the reference: the shuffle of libcudf_streaming (cpp/libcudf_streaming/src/partition_utils.cpp:
 * 72-117 hash / range partition -> rapidsmpf shuffler -> partition.cpp:56-80 unpack) and the collectives of
 * cudf_polars' streaming executor (python/cudf_polars/cudf_polars/streaming/actor_graph/collectives/sort.py,
 * streaming/join.py:58-135, streaming/groupby.py:411-437).  Here each operator is ONE call per rank that
 *   - partitions the shard in CHUNKS with the HIP kernels of gx.h (gx_partition_rows_at: range split for the sort,
 *     hash split for join / groupby), all chunks enqueued up front on the caller's stream,
 *   - exchanges chunk k on a second stream while chunk k+1 is still being partitioned: the count matrix of a chunk
 *     travels by ncclAllGather, its rows as ONE grouped ncclSend / ncclRecv per peer (the full xGMI mesh: 7 links busy),
 *     and the host only ever waits for a chunk's counts while the GPU works on the next chunk,
 *   - runs the local operator on what arrived (join: chunk by chunk, overlapping the exchange of the later chunks).
 * No all-reduce appears on the data path.  Results stay sharded.
 *
 * Conventions: as gx.h (0 = success, negative GX_E*, positive hipError_t; RCCL errors are reported as GX_EINTERNAL with
 * the text in gxd_last_error()).  Pointers are DEVICE pointers unless named *_host.  RESULT buffers are obtained through
 * the caller's allocator callback (their size is known only after the exchange); everything else lives in a grow-only
 * arena owned by the communicator object and is reused from call to call (persistent send / receive buffers).
 */
#ifndef CUDF_AMD_GXD_H
#define CUDF_AMD_GXD_H

#include <cudf_amd/gx.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gxd_comm gxd_comm; /* an RCCL communicator + exchange stream + persistent buffers */
typedef struct gxd_join gxd_join; /* the local hash table of a sharded build side */

/* device memory for a result: `bytes` > 0; return NULL on failure.  The caller frees it its own way. */
typedef void* (*gxd_alloc_fn)(size_t bytes, void* ctx);

/* 128-byte id for gxd_comm_create (ncclGetUniqueId): made by ONE rank, handed to the others out of band */
int gxd_unique_id(void* id128_host);
/* ncclCommInitRank on the CURRENT device (collective over all `world` ranks) */
int gxd_comm_create(const void* id128_host, int world, int rank, gxd_comm** out);
/* `world` LOGICAL ranks on the CURRENT device: out[0 .. world) receive communicators that share an in-process loopback
 * fabric -- device-to-device copies stand in for the xGMI links, the operators above run unchanged ("validate
 * the partition/exchange logic with N logical ranks on one device").  Every communicator must be driven by a host thread of
 * its own (the operators are collective: rank r's call returns only when its peers have made theirs) on a NON-BLOCKING stream
 * of its own.  A rank that fails breaks the fabric: its peers' calls return GX_EINTERNAL instead of waiting for it.
 * world <= 16 here and in gxd_comm_create; any world size (hash destinations are a multiply-shift of the hash, not a mask). */
int gxd_comm_create_loopback(int world, gxd_comm** out);
int gxd_comm_destroy(gxd_comm* comm);
/* From ANOTHER host thread (a watchdog): make the operator call that is blocked inside a collective of `comm` -- a peer died,
 * or never made its call -- return an error, and every later call fail at once.  RCCL: ncclCommAbort (the communicator is
 * unusable afterwards; gxd_comm_destroy still releases the buffers); loopback fabric: the fabric is broken for all its ranks.
 * The reference's analogue: rapidsmpf's communicator shutdown behind libcudf_streaming's shuffle (partition_utils.cpp:72-117). */
int gxd_comm_abort(gxd_comm* comm);
int gxd_comm_rank(const gxd_comm* comm);
int gxd_comm_world(const gxd_comm* comm);
const char* gxd_last_error(void);
/* TEST HOOK: scales the slot capacity of the speculative partition passes (0 = default margin of 4/3).  A scale below 1 makes
 * every chunk overflow its slots, so the exact re-partition path runs even on a single rank. */
void gxd_test_set_slot_scale(double scale);
/* TEST HOOK: narrows the row field of the (source rank << s) | row codes to `bits` (0 = the default s = 31 - ceil(log2(world))), so
 * that shards of a few million rows already exceed it: build shards then take the documented positions + gather fallback and
 * probe shards are cut into more chunks. */
void gxd_test_set_row_bits(int bits);
/* TEST HOOK: which path gxd_sort takes.  0 (default) = the exchange between the sort's two partition levels for INT32 / UINT32 /
 * INT64 / UINT64 keys from 2^25 rows per rank on average, the sample-sort path otherwise; 1 = always the sample-sort path;
 * 2 = the fused path from 2^21 rows per rank (so that tests reach it with small shards). */
void gxd_test_set_sort_mode(int mode);
/* milliseconds the last operator call of this communicator spent in: [0] partition kernels, [1] host waits for counts,
 * [2] whole call (host clock, the stream is synchronised at the end of every operator) */
int gxd_last_timing(const gxd_comm* comm, double* ms3_host);

/* Global sort of the concatenation of all ranks' shards: rank r receives the r-th range, sorted; the concatenation of the
 * results in rank order is sorted.  Integer keys, large shards: the exchange sits BETWEEN the sort's two partition levels --
 * every rank runs level 0 (256 bins on digit positions all ranks agree on), whole bins are dealt to ranks by the all-gathered
 * histogram, ONE span of the level-0 buffer travels per peer, the receiver runs level 1 + the cell sort (gx.h gx_sortx_*).
 * Otherwise (floats, small shards, a bin too heavy for one rank, a failed device-side check): sample sort -- strided sample ->
 * all-gather -> common splitters -> ONE range-partition pass per chunk -> exchange -> ONE local sort.
 * dtype: a 4- or 8-byte numeric gx_dtype.  chunks <= 0: default (8; sample-sort path only).
 * force_exchange != 0: take the exchange path even for world == 1 (measurements / tests).
 * *out_keys (alloc'ed, *out_n elements). */
int gxd_sort(gxd_comm* comm, int dtype, const void* keys, int64_t n, int chunks, int force_exchange, gxd_alloc_fn alloc,
             void* alloc_ctx, void** out_keys, int64_t* out_n, gx_stream_t stream);

/* cudf::hash_join over sharded tables.  build: hash-partition this rank's build keys (key + int32 row code = 12 B/row on
 * the wire), exchange once, build the local table over what arrived.  probe: per chunk partition -> exchange -> local
 * partitioned probe; every pair comes back as (global probe row, global build row), global row = (first row of the owning
 * rank's shard) + local row.  The int32 that travels with a key is (source rank << s) | row, s = 31 - ceil(log2(world)):
 * it rides through the receiver's partition pass and hash table as the payload and is turned into the global row by one
 * streaming pass over the pair arrays (gx_decode_global_rows) -- no gather through the received row maps.  Probe rows are
 * counted inside the sender's chunk (chunks are cut to fit the row field); a build shard of more than 2^s rows falls back
 * to positions + gx_gather_global_rows_dev.  key_dtype: an 8- or 4-byte numeric type (bit patterns are compared).  No nulls. */
int gxd_join_build(gxd_comm* comm, int key_dtype, const void* build_keys, int64_t n, int force_exchange, gx_stream_t stream,
                   gxd_join** out);
int gxd_join_probe(gxd_join* table, const void* probe_keys, int64_t n, int chunks, gxd_alloc_fn alloc, void* alloc_ctx,
                   int64_t** out_probe_rows, int64_t** out_build_rows, int64_t* out_pairs, gx_stream_t stream);
int gxd_join_destroy(gxd_join* table);

/* groupby(keys).agg(sum, count) over all shards: local aggregate (gx_groupby_sum_count) -> hash-partition the partial
 * (key, sum, count) rows -> exchange -> merge.  Every group ends on exactly one rank, keys ascending.
 * key_dtype INT32 / INT64; val_dtype INT32 / INT64 (sums INT64) or FLOAT32 / FLOAT64 (sums FLOAT64). */
int gxd_groupby_sum_count(gxd_comm* comm, int key_dtype, const void* keys, int val_dtype, const void* vals, int64_t n,
                          int64_t max_groups, int force_exchange, gxd_alloc_fn alloc, void* alloc_ctx, void** out_keys,
                          void** out_sums, int64_t** out_counts, int64_t* out_groups, gx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CUDF_AMD_GXD_H */
