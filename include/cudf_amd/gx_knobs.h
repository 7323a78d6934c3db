/*
 * gx_knobs.h -- tuning knobs and measurement hooks of libcudf_amd.  NOT part of the drop-in boundary (gx.h).  They exist for
 * A/B measurements (bench.py, scripts/) and for tests that must force a code path (fallback algorithms, speculative passes at
 * small sizes); a production caller never calls them.
 * SCOPE: every knob and profile hook is state of the CALLING THREAD (thread_local): it changes what the gx_* entry points do
 * when THAT thread calls them and nothing else, so concurrent callers -- hash_join probes are documented as callable in
 * parallel (cpp/include/cudf/join/hash_join.hpp:63-68) -- never see another thread's experiment.  Where a comment below still
 * says "process-wide" read "per calling thread".  The one exception is gx_groupby_set_partition_bits (it mirrors its value
 * into device memory): process-wide, set it only while no groupby is running.
 * CONSEQUENCE for the multi-rank test drivers (cudf_amd/gxd.py run_ranks, communicator::loopback in C++): every logical rank runs
 * on a host thread of its own, so a knob set on the test's main thread is NOT seen by the rank threads -- set it inside the
 * per-rank function (the gxd_test_set_* hooks of gxd.h are process-wide for that reason).
 */
#ifndef CUDF_AMD_GX_KNOBS_H
#define CUDF_AMD_GX_KNOBS_H

#include <cudf_amd/gx.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Tuning / A-B knob (process-wide): 0 = onesweep (decoupled look-back, 8-tile look-back window,
 * default), 1 = three-kernel passes (tile histogram + scan + scatter; no inter-workgroup
 * communication), 2 = onesweep with a one-tile-per-hop look-back (the textbook form; slower on
 * this chip, kept for A/B measurements). */
void gx_sort_set_algorithm(int algo);

/* Measurement hooks (bench.py's roofline leg): when enabled, every sort records HIP events on
 * the caller's stream around the histogram launch and around each pass's launch(es);
 * gx_sort_profile_read waits for the last sort and returns the durations in milliseconds
 * (pass_ms has room for 8 entries; skipped passes report the few microseconds of their early
 * exit).  enable == 2: only the two events around the first partition level are recorded (an event
 * between two kernels costs the stream ~15 us; a sort records 23): gx_sort_profile_read then reports no
 * pass, gx_sort_profile_read_hybrid the first interval and zeros. */
int gx_sort_profile(int enable);

int gx_sort_profile_read(float* hist_ms, float* pass_ms, int* npass);
/* Event slot (0..63, per calling thread) the following sorts record into and gx_sort_profile_read* read from: K timed calls take K
 * slots and are read after the last one, so that no read-back sits between two timed calls. */
int gx_sort_profile_slot(int slot);

/* durations of the hybrid path's kernels of the last profiled sort, in milliseconds:
 * ms4 = {level-0 partition pass, level-1 partition pass, cell plan (one block), LDS local sort}.
 * GX_EINVAL when the last sort did not enqueue the hybrid path. */
int gx_sort_profile_read_hybrid(float* ms4);

/* Hybrid MSD path (64-bit keys, n >= 2^22): an up-front pass finds the varying bits and histograms the
 * level-0 digit (the 8 bits below the highest varying bit), two partition passes (8 + up to 9 bits), then one
 * kernel that sorts every cell of <= 8192 / 16384 keys on its remaining bits inside LDS (64 B/row of HBM
 * traffic instead of 136).  Enabled by default; the device falls back to the LSD passes by itself when a cell
 * does not fit (skewed keys).  0 disables it (A/B measurements). */
void gx_sort_set_hybrid(int enable);

/* Bits of the hybrid path's local sort.  MEASUREMENT ONLY (the output is not sorted under these): 4 = skip the per-wave
 * sub-bucket sorts, 8 = no LDS atomics in the sub-bucket split (positions instead of ranks), 16 = sorting networks instead
 * of the counting split.  A/B knob (the output IS sorted): 32 = every 8192-key cell takes k_local_sort's sub-bucket path,
 * i.e. k_local_place (counting placement + per-thread window networks) is switched off.  0 = production. */
void gx_sort_set_experiment(int bits);
/* A/B knob (process-wide): workgroups of k_local_place.  0 (default) = one per cell; otherwise that many workgroups walk the
 * cells with a stride (persistent form). */
void gx_sort_set_place_grid(int workgroups);
/* EXPERIMENT (process-wide, default 0): gx_sorted_order of an int32 / uint32 column without nulls, n >= 2^25, as a keys-only sort
 * of the 64-bit words (sortable key << bits(n - 1)) | row on the cursor path (pack, sort, unpack).  Faster for well-spread keys
 * (1e9 rows: 22.2 -> 17.0 ms), much slower for keys with ~1000 rows each (cells overflow: 55-59 ms); see gx_sort.hip. */
void gx_sort_set_order_words(int enable);
/* Cells of the last hybrid sort that used `tmp` which k_local_place found crowded (a bin of the 13-bit counting pass with
 * more than 9 keys: duplicates, clusters) and left to k_local_sort; 0 when every cell was placed, or when k_local_place
 * did not apply (16384-key cells, float keys, fewer than 13 key bits left, knob).  Synchronises `stream`. */
int gx_sort_place_info(const void* tmp, int32_t* todo_cells_host, gx_stream_t stream);

/* Big cells of the last cursor-path sort that used `tmp` (round 4): info3 = {1 when cells that outgrew their slot -- a hot value --
 * were sorted on their own (compacted into X, X sorted by the LSD passes, copied back) while every other cell took the fast
 * path, 0 otherwise; number of such cells; keys in them}.  A column whose big cells hold more than half its keys still falls
 * back to the whole-column LSD passes (mode 0, gx_sort_info's num_active > 0).  Synchronises `stream`. */
int gx_sort_big_info(const void* tmp, int64_t* info3_host, gx_stream_t stream);

/* Cursor path of the hybrid sort (integer 64-bit keys, keys only, n >= 2^25; default on): the digit positions and
 * the slot capacities of the first partition level come from a 1/32 SAMPLE, both partition levels reserve their output
 * runs with one atomic per (tile, bin) instead of a look-back chain, and the first level -- which reads every key anyway --
 * verifies the sample (exact varying-bit masks, slot counts).  On a miss the look-back path sorts the column; the device
 * decides.  enable = 0 switches it off (A/B); margin_sigmas = slack per slot in standard deviations of the estimate
 * (0 = default 8; a negative value makes every slot too small: TEST HOOK for the fallback). */
void gx_sort_set_cursor_path(int enable, float margin_sigmas);
/* A/B knob (per calling thread): 1 (default) = integer keys whose varying bits are their low 15 or fewer (sampled, then verified on
 * every key) are sorted by a histogram + fill (the counting sort of round 5, FastPlan::state 5); 0 = the LSD passes, as before. */
void gx_sort_set_counting(int enable);
/* A/B knob (per calling thread): 1 (default) = a 64-bit integer column whose level-0 buckets the sample shows to be too uneven for
 * two levels of bit digits (bell-shaped, lognormal, Zipf-like, clustered values) is cut on sample-chosen SPLITTERS instead
 * (round 5: k_sp_plan / k_sp_level0, cells cut on a per-bucket warp of the key's position, equality buckets for heavy values); 0 = such a
 * column is declined to the LSD passes, as before. */
void gx_sort_set_splitters(int enable);
/* A/B knob (per calling thread): 1 (default) = FLOAT64 keys-only sorts of >= 2^25 rows take the cursor path on the IEEE total-order
 * flip; every key is checked, and a column with a NaN or a -0.0 -- where an unordered sort and the reference's stable radix sort of
 * (isnan * (idx + 1), value) pairs (cpp/src/sort/sort_radix.cu:36-117) could differ -- is sorted by the stable look-back path
 * instead (decided on the device); 0 = float keys always take the look-back path, as before round 5. */
void gx_sort_set_float_cursor(int enable);
/* A/B knob (per calling thread; round 6): 1 (default) = gx_sorted_order of a 64-bit column without nulls, from 2^25 rows, is a KEYS-ONLY
 * sort of (monotone rank << row bits | row) words + a pass that puts runs of equal ranks right by (key, row) (gx_order.hip): the cost of
 * the keys-only paths on any value distribution; 0 = the round-3 pairs path (look-back levels on key bits; uneven columns -> LSD passes). */
void gx_sort_set_order_map(int enable);
/* Measurement / test hook: state of the last gx_sorted_order that took the word sort with this scratch and row count: info5 = {runs of
 * more than 16 equal ranks (sorted by the long-run pass), long-run list overflow (always 0), row bits, rank bits inside a bucket, buckets
 * whose rank drops key bits}.  Synchronises `stream`. */
int gx_sort_order_map_info(const void* tmp, int64_t n, int32_t* info5_host, gx_stream_t stream);
/* 1 when gx_sorted_order of a column of `dtype` with n rows and no nulls takes the word sort (knob on, 64-bit dtype, 2^25 <= n < 2^31), else 0. */
int gx_order_map_applies(int dtype, int64_t n);
/* (per calling thread) how long a look-back wait may make no progress before it is abandoned and the sort's status word becomes 5
 * (gx_sort_status): milliseconds of wall-clock time, 0 = the default 30 s.  Tests shorten it. */
void gx_sort_set_spin_limit_ms(int ms);
/* TEST HOOK (per calling thread): tile `tile` of every look-back pass of the sorts this thread issues never publishes its granules --
 * the lost-chain fault the guard exists for; -1 = none (default). */
void gx_sort_inject_lost_tile(long long tile);
/* info4 = {splitter mode used, splitters, equality buckets, level-1 bits} of the last sort on this scratch (synchronises). */
int gx_sort_split_info(const void* tmp, int32_t* info4_host, gx_stream_t stream);
/* 0 = not tried, 2 = tried and rejected by the device (the look-back path ran), 3 = the cursor path sorted the column,
 * 4 = the sample showed a key range too narrow for two partition levels (the LSD passes ran, no up-front read of the column).
 * `tmp` is the scratch of that sort call; synchronises the stream. */
int gx_sort_cursor_state(const void* tmp, int32_t* state_host, gx_stream_t stream);

/* A/B knob (process-wide): capacity of a local-sort cell of the hybrid path.  0 = auto (8192-key cells, two
 * workgroups per CU and a 9-bit second partition level, for integer keys-only sorts of up to ~1.02e9 rows;
 * 16384-key cells otherwise), 8192 / 16384 = force where the key kind allows it. */
void gx_sort_set_cell(int keys);

/* A/B knob (process-wide): predecessors a tile of the keys-only hybrid partition passes examines per look-back
 * round (4, 8, 16 = default). */
void gx_sort_set_lookback(int window);

/* Measurement hooks (bench.py's roofline leg), like gx_sort_profile: when enabled, every partitioned probe
 * records HIP events on the caller's stream; gx_join_profile_read waits for the last one and returns
 * ms3 = {partition histogram + offsets, scatter of (key, row) into partitions, probe} in milliseconds. */
int gx_join_profile(int enable);

int gx_join_profile_read(float* ms3);
/* Event slot (0..63, per calling thread), as gx_sort_profile_slot. */
int gx_join_profile_slot(int slot);

/* A/B knob: kernel of the partitioned build.  0 (default) = the window build: the partition's rows regrouped by 2^12-slot window,
 * every window composed in LDS and written once in full lines (no pre-fill of the table, no global atomics); 2 = one workgroup
 * per 2^17-slot sub-table, slots claimed through the 4-bit tags held in LDS and stored one by one; 1 = the round-2 kernel
 * (device-scope CAS on the table's slots, then k_tags over the whole table). */
void gx_join_set_build_kernel(int which);
/* A/B knob (process-wide): rows per workgroup tile of the partition scatter (4096, 8192, 16384; 0 = default:
 * the largest that fits the LDS next to the per-partition counters). */
void gx_join_set_scatter_tile(int rows);

/* A/B knob (per calling thread; round 6; default 133 = bits 0, 2 and 7): bit 0 = the partition pass of the partitioned probe writes 12-byte
 * {key, row} RECORDS (one 96-B run per tile and partition instead of a 64-B key run and a 32-B row run: k_pj2_scatter_rec) and the
 * pipelined probe reads them (8-byte keys, default probe kernel only); bit 1 = with bit 0: 24576-row scatter tiles (12-row runs);
 * bit 2 = the pipelined probe's service wave takes tickets that ARE (region, piece) -- every region is cut into the same number of
 * pieces -- and issues the ticket atomic and the fill-counter read one trip ahead of their use (0: the round-3 chain of four
 * dependent round trips per piece); bit 3 = without bit 0: the windowed scatter writing key / row arrays (measured slower);
 * bits 4-6 = ablations of the probe for measurements (WRONG results: 1 no slot reads, 2 no staging, 3 no tag lookups; with bit 7:
 * 4 no chain walks); bit 7 (with bit 0, tables of <= 2^28 slots) = the probe with two register sets, every load requested a whole trip
 * before its use, rows that need a dependent read deferred to an overflow list (k_pj2_probe_rare); bits 8-15 / 16-23 (measurement):
 * workgroups / 4 of the partition pass / of the probe; bit 24 (measurement): the partition pass bumps its fill counters in region order
 * (fill[partition * 8 + range]: every line of counters shared by the eight XCDs, the layout until round 6: 5.8 ms against 4.7). */
void gx_join_set_experiment(int bits);

/* Tests (per calling thread): rows per workgroup slice of the overflow list of the probe selected by gx_join_set_experiment bit 7
 * (k_pj2_probe_pipe<LONG>: every load a trip ahead of its use; rows that need a dependent read go to the list and are settled by
 * k_pj2_probe_rare).  0 = the default, n / 16 / 256 rows; a full slice makes its workgroup walk such rows in place. */
void gx_join_set_overflow_slice(int rows);

/* A/B knob (per calling thread): 0 = software-pipelined tag probe on LDS-resident 4-bit tags (default), 1 = the round-1 tag probe,
 * 2 / 3 = the L2-resident DIRECT probe of round 5 (k_pj3_probe_direct, 4 / 2 rows per thread): no tags, no LDS tables -- the
 * workgroups of an XCD take the pieces of the XCD's partitions in order, so the 2-MiB sub-table they all probe sits in the
 * XCD's L2 and every row reads its home slot there; 4 / 5 = k_pj4_probe_tags, the LDS-tag probe without the software pipeline and the
 * pair staging (2 rows per thread and two workgroups per CU / 4 rows and one); 6 / 7 = the same with the tag windows read from the L2
 * instead of LDS -- these two ALSO change the partition count of the probe (2^20-slot sub-tables: an eighth of the partitions).
 * All measured at parity or slower in round 5 (profiles/r5_join_probe_ab.txt); values outside 0-7 select 0. */
void gx_join_set_probe_kernel(int which);

/* A/B knob (process-wide): speculative = 1 (default) partitions the probe rows WITHOUT a histogram pass into padded
 * (partition, XCD range) slots with a persistent scatter kernel, falling back on the device to the exact histogram
 * path when a slot overflows (skewed keys); 0 = always the exact path of round 2; 2 = speculative for every row count
 * (tests: by default inputs below 1.7e7 rows take the round-2 path).  early_loads bit 0 (default 0): the
 * pipelined probe requests a piece's rows at the top of a trip instead of at its end; bit 1 set: rows whose chain is not
 * settled by their first candidate slot are parked in a per-wave queue for one trip instead of being finished in place (A/B;
 * default 0: the queue measured slower once the tag window grew to 16 slots). */
void gx_join_set_partition_mode(int speculative, int early_loads);

/* Tuning / A-B knob (process-wide).  algo: 0 = auto (hash-partition rows into 256 LDS-sized
 * partitions and aggregate each in one workgroup's LDS when n >= 2^19, else the global-atomic
 * table), 1 = global-atomic table only, 2 = partitioned for every n > 0.  nsplit: workgroups per
 * partition in the LDS aggregation kernel (1..16). */
void gx_groupby_set_algorithm(int algo, int nsplit);

/* A/B knob (calling thread; round 6): 1 (default) = gx_groupby_sum_count takes DENSE ids by direct address -- where the sampled keys span
 * at most 256 x 7680 values (8-byte values without nulls, integer keys of 4 or 8 bytes, n >= 2^22, 4 max_groups <= n) a partition is an
 * id range, a row travels as its value + a 16-bit remainder (10 B instead of 12) and the aggregate's LDS table is indexed by the
 * remainder; a key outside the planned range falls back to the exact hash sequence on the device.  0 = the hash path of rounds 3 - 5. */
void gx_groupby_set_dense(int on);
/* Which path the last gx_groupby_sum_count that used `tmp` (called with this max_groups) took: info[0] = 1 dense ids by direct address,
 * [1] = 1 a slot overflowed or a key lay outside the planned id range (the exact sequence produced the result), [2] = partition bits of
 * the call's plan (0: process-wide), [3] = ids per partition of the dense path.  Synchronises `stream`. */
int gx_groupby_plan_info(const void* tmp, int64_t max_groups, int32_t* info4_host, gx_stream_t stream);

/* A/B knob (process-wide) of the LDS-partitioned path: 1 (default) = the partition pass runs WITHOUT its histogram into padded
 * (partition, XCD range) slots for n >= 2^22, with the exact histogram path as device-side fallback when a slot overflows
 * (skewed keys); 0 = always the exact path; 2 = speculative for every n (tests). */
void gx_groupby_set_partition_mode(int speculative);

/* A/B knob (process-wide, synchronous): hash partitions of the LDS-partitioned groupby, 2^bits with bits = 8 or 9 fixed, or
 * 0 (default since round 4): 9 bits -- half as many groups per LDS table: keys that hash like random numbers stay below 35 % load
 * at 1e6 groups -- except that the SUM / COUNT path picks 8 bits per call, on the device, when its sample shows DENSE ids (keys
 * below 2 x max_groups, max_groups a real bound): consecutive integers barely collide under the Fibonacci slot hash and 256
 * partitions give the scatter 32-row instead of 16-row runs (9.5 vs 10.1 ms per 1e9 rows; sparse keys 12.7 vs 19.5 ms the other way). */
int gx_groupby_set_partition_bits(int bits);

#ifdef __cplusplus
}
#endif
#endif /* CUDF_AMD_GX_KNOBS_H */
