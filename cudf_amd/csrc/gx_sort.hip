// gx_sort.hip -- LSD radix sort for gfx950 (MI355X), keys-only and key+int32-payload.
//
// Replaces the cub::DeviceRadixSort calls of the reference (cpp/src/sort/sort_radix.cu:69-76,
// cpp/src/sort/sorted_order_radix.cu:83-94).  Design (DESIGN.md "radix sort"):
//   * one up-front histogram kernel computes the 256-bin histogram of EVERY 8-bit digit in a
//     single read of the keys (8 B/row), LDS-staged, wave-uniform fast path for constant digits;
//   * a one-block plan kernel turns the histograms into global bin bases, marks passes whose
//     digit is constant over the whole column as skipped (small-range ints sort in 1-2 passes),
//     and assigns ping-pong buffers so the last active pass lands in the caller's output;
//   * per active pass ONE scatter kernel (16 B/row for int64): a 512-thread workgroup ranks a
//     tile of up to 8192 keys with wave64 ballot match-and-count (stable), reorders the tile in
//     LDS so every bin leaves as one contiguous run, and obtains its global bin offsets by
//     decoupled look-back over 8-byte {flag,epoch,count} granules exchanged with relaxed
//     agent-scope atomics (the per-XCD L2s are not coherent: MI355X_MICROARCH.md);
//     tiles are handed out by an atomic ticket so every predecessor of a running tile is itself
//     running or finished -- no dependence on dispatch order or XCD placement;
//   * algorithm 1 (A/B knob, no inter-workgroup communication): per pass a tile-histogram
//     kernel + device scan + the same scatter kernel reading precomputed offsets.
#include "gx_common.hpp"
#include <cstdlib>
#include <type_traits>
#include <vector>
#include "gx_scan.hpp"

namespace gx {
namespace sort {

constexpr int BINS       = 256;
constexpr int MAX_PASSES = 8;
constexpr int BT         = 512;  // threads per workgroup (8 waves)
constexpr int NW         = BT / GX_WAVE;
constexpr int NRANGE     = 8;    // input ranges == XCDs: each gets its own look-back chain (hybrid level 0)

// Hybrid MSD sort (64-bit keys, n >= 2^22): two MSD partition passes bring every cell (= keys sharing
// all bits above shift2) down to at most one LDS-resident cell, then ONE kernel sorts each cell on all
// remaining bits inside LDS.  HBM traffic: 8 (mask + histogram) + 16 + 16 + 16 = 56 B/row instead of
// 8 + 8 x 16 = 136 B/row for the 8-pass LSD (round 1 also read the bucketed keys once more for a joint
// bucket x digit histogram; level 1 now writes into padded cell slots and needs none).
//   * digits are BIT granular: an up-front pass reduces the OR of the keys and the OR of their complements
//     (a bit varies iff it is set in both) next to a speculative histogram of the top byte; level 0 takes the
//     8 bits below the highest varying bit, so keys confined to a range that is not byte aligned (a rank's
//     shard of a distributed sort, timestamps, n a few per cent above a power of two) stay on this path;
//   * level 1 takes the next `bits2` <= 9 bits.  Keys-only sorts use cells of <= 8192 keys (64 KiB): two
//     local-sort workgroups share a CU, so one cell's loads and stores overlap the other's LDS work;
//     pairs / larger n use 16384-key cells (one workgroup per CU), the round-1 configuration.
// Whether every cell fits is decided ON THE DEVICE (level 1 raises hy.overflow when a cell outgrows its slot,
// k_plan2 checks the cell counts); when one does not (skewed keys) the remaining hybrid kernels turn into
// no-ops and the LSD passes below run instead.
constexpr int CS_MAXBITS = 15;   // counting sort of narrow key ranges (k_cs_*): 32768 bins = 128 KiB of LDS counters, one 1024-thread workgroup per CU
constexpr int NB2MAX     = 1024; // level-1 bins: row stride of the cell tables (cursor path: <= 10 bits)
constexpr int NB9        = 512;  // bins of the 9-bit look-back level-1 pass

struct HybridPlan {
  int32_t attempt;  // k_hy_plan: the hybrid path is being tried
  int32_t ok;       // k_plan2: every cell fits -> LSD passes are skipped, local sort runs
  int32_t shift0;   // level-0 digit: bits [shift0, shift0 + 8) of the sortable key
  int32_t shift2, bits2;  // level-1 digit: bits [shift2, shift2 + bits2)
  int32_t nlocal;         // LDS passes of the local sort's stable fallback
  int32_t lshift[MAX_PASSES], lbits[MAX_PASSES];
  int32_t need_hist;      // the speculative top-byte histogram is not the level-0 digit: k_hy_hist<false> runs
  int32_t cell_max;       // capacity of a local-sort cell (8192 or 16384)
  int32_t overflow;       // level 1: a cell outgrew its slot (skewed keys) -> LSD fallback
  int32_t bad;            // k_plan2: the cell sizes of a bucket do not add up to the bucket (protocol failure)
  uint32_t plan2_done;    // k_plan2: buckets finished (the last one decides)
  unsigned long long or_mask, nor_mask;  // OR of the sortable keys / of their complements
  uint32_t list_tile0[2][NRANGE + 1];  // first global tile of each list
  uint32_t seg_tile0[2][BINS + 1];     // first global tile of each segment (level 0: NRANGE segments)
  uint32_t seg_start[2][BINS], seg_count[2][BINS];
  uint32_t max_cell;
  uint32_t hist0[BINS], gbin0[BINS];   // level-0 digit histogram of the whole column and its exclusive scan
  uint32_t rh0[NRANGE][BINS];          // range-resolved level-0 histogram
  uint32_t rhist[NRANGE][MAX_PASSES][BINS];  // range-resolved byte histograms (k_hist_all, LSD path)
  unsigned long long fold_or;                // signed integer keys: OR of (key XOR its sign extension), see k_plan
  // SIGN FOLD of the level-0 digit (round 4; signed integer keys spread around zero): when every bit from position h up to the
  // sign bit is a copy of the key's sign, the highest varying bit of the sortable key is the (flipped) sign and the 7 bits below
  // it are constant inside either sign class -- level 0 would make TWO buckets.  With `fold` set the level-0 digit is
  // (sign << 7) | bits [h - 7, h): shift0 = h - 7, level 1 and the cell sort take the bits below as before (all of them are
  // below h; the bits in [h, sign) are equal inside a bucket because the sign is part of its digit).
  int32_t fold;
  unsigned long long fold_x;                 // OR of (key ^ sign extension): the sample's (k_hf_sample), then the EXACT one (level 0 / k_hy_hist)
  alignas(128) uint32_t todo_count;          // k_local_place: cells left to k_local_sort (a line of its own: atomics)
  uint32_t todo_pad[31];
  // Round 4, BIG cells of the cursor path: a cell that outgrew its slot (a hot value: 1e6 copies of one key land in ONE cell
  // whatever the digits) costs that cell, not the column.  Its true size is known (the level-1 cursors count every key, also the
  // dropped surplus), so every OTHER cell is sorted as usual; the big cells' keys are fetched again from the level-1 input
  // (only the level-0 buckets that hold one are re-read), compacted into X, X is sorted by the LSD passes -- the same kernels
  // as the whole-column fallback, run on X with a DEVICE-side length -- and copied into the cells' output ranges.
  int32_t big;        // k_plan2: some cell outgrew its slot
  int32_t lsd_mode;   // k_big_plan: 1 = the LSD passes sort X (kbufB, lsd_n keys) instead of the whole column
  uint32_t nbig;      // big cells
  uint32_t big_pad;
  unsigned long long lsd_n;  // keys in X
  uint32_t bigbucket[BINS];  // level-0 buckets that hold a big cell (the rescue pass re-reads only these)
  uint32_t bigcnt[BINS], bigkeys[BINS];    // k_plan2: big cells of bucket b, keys in them
  uint32_t biglist0[BINS], bigx0[BINS];    // k_big_plan: first list entry / first X position of bucket b's big cells
  // Round 4: PER-BUCKET cell slots.  Level 1 used to give every cell a slot of cell_max keys -- 2^17 x 8192 keys for a 1e9-row
  // sort, twice that where the device may take one more level-1 bit or the larger cells: 17.2 of the sort's 34.4 GB of scratch.
  // The plan kernels have the exact level-0 histogram, so bucket b's cells get ccap[b] = its mean cell + 6 sigma + 64 keys (a
  // multiple of 16, at most cell_max) from cbase[b] on: the cell buffer shrinks to n + 640 (848) keys per cell.  ccap[b] == 0:
  // the fixed layout (cell c at c * cell_max).
  uint32_t cbase[BINS], ccap[BINS];
};
__device__ __forceinline__ uint32_t cell_cap(const HybridPlan& hy, uint32_t b) { return hy.ccap[b] ? hy.ccap[b] : (uint32_t)hy.cell_max; }
__device__ __forceinline__ uint32_t cell_slot(const HybridPlan& hy, uint32_t b, uint32_t d2)
{
  return hy.ccap[b] ? hy.cbase[b] + d2 * hy.ccap[b] : ((b << hy.bits2) + d2) * (uint32_t)hy.cell_max;
}
// keys per cell beyond the bucket's mean cell that a slot holds: 6 sigma of a cell of cell_max keys + 64 + rounding (host bound)
static inline size_t cell_slack(int cell_max) { return (size_t)(6.0 * __builtin_sqrt((double)cell_max)) + 64 + 16 + 2; }

// Round 3, the CURSOR path (integer keys, keys only): plan of the speculative passes, see k_hf_scatter
struct FastPlan {
  int32_t state;   // 0 not tried / given up before level 0; 1 planned from the sample; 2 failed the check after level 0
                   // (the look-back path then runs from scratch); 3 verified (the look-back path is skipped);
                   // 4 the sample shows a key range too narrow for two partition levels: straight to the LSD passes
                   // 5 (round 5) the sample shows at most CS_MAXBITS varying low bits: COUNTING SORT (k_cs_*; a key outside that
                   //   range, seen by the count's exact check, turns it into 4)
  int32_t cs_bits;              // counting sort: the low bits that vary (bins = 1 << cs_bits)
  uint32_t cs_fail;             // k_cs_count: a key differs from cs_base above cs_bits (the sample missed it)
  uint32_t cs_groups;           // k_cs_scan: non-empty bins
  unsigned long long cs_base;   // the bits all (sampled) keys share above cs_bits, sortable form
  int32_t fail;    // level 0: a slot outgrew its capacity
  int32_t stride;  // the sample takes every stride-th 64-key chunk
  int32_t hist_ready;  // the first sample kernel's speculative top-byte histogram IS the level-0 digit's (full-range keys)
  int32_t slots_ready; // k_hf_plan stage 1 has sized the level-0 slots (its second launch -- behind the splitter planning -- is then a no-op)
  uint32_t samp[NRANGE][BINS];              // sample histogram of the level-0 digit, per input range
  uint32_t slot0[NRANGE][BINS];             // level-0 output: first key of slot (range, bin) ...
  uint32_t cap0[NRANGE][BINS];              // ... and its capacity
  uint32_t reg_tile0[BINS * NRANGE + 1];    // level 1: first tile of region q = bucket * NRANGE + range
  uint32_t reg_start[BINS * NRANGE], reg_count[BINS * NRANGE];
  alignas(128) uint32_t cur0[NRANGE][BINS];  // level-0 cursors = keys in slot (range, bin); atomics: lines of their own
  // The SHARDED sort (gx_sortx_*: the exchange sits between the two partition levels).  Sender: the digit positions come from
  // masks the caller supplies (the OR over all ranks), so the verdict after level 0 does not compare the local top bit with them.
  // Receiver: level 1 reads an EXTERNAL region table -- regions (bucket, source rank, input range) of the receive area, in
  // bucket order -- instead of reg_* above.
  uint32_t forced_masks;
  // (round 5) the masks that were forced, kept: level 0 reduces this rank's EXACT masks into hy.or_mask / nor_mask, and
  // a key with a bit no rank's SAMPLE saw (one sentinel among ids, a rare odd key among evens) must fail the verdict -- its
  // level-0 digit would be truncated and the receiver's cell sort would skip a byte it believes constant
  unsigned long long forced_or, forced_nor;
  uint32_t ext, nreg;
  uint32_t slot_total;      // sender: rows of the level-0 buffer in use (end of the last slot)
  uint32_t* x_tile0;        // [nreg + 1] first tile of region q (filled by k_hfx_plan)
  const uint32_t* x_start;  // [nreg] first key of region q in the level-1 input
  const uint32_t* x_count;  // [nreg]
  const uint32_t* x_bucket; // [nreg] its level-0 bucket
};

// Every word that workgroups update with atomics lives on its own 128-B line, away from the
// read-only plan: a device-scope atomic drops its line from the L2s, so a counter that shares a
// line with fields every workgroup reads turns those reads into memory-side round trips queued
// behind the atomics.
struct alignas(128) Counter {
  uint32_t v;
  uint32_t pad[31];
};
struct SortCounters {
  Counter tickets[MAX_PASSES];  // LSD passes
  Counter ctr[2][NRANGE];       // hybrid: tile tickets per level and per XCD list
};

// ---------------------------------------------------------------------------------------------------------------------------
// Round 5: SPLITTER mode of the cursor path (VERDICT r4 "missing" 1: a sort whose cost does not depend on the value distribution).
// Two levels of BIT digits cut the key space into equal-width slices, so a smooth but uneven density -- bell-shaped, lognormal,
// Zipf-like, clustered values -- leaves level-0 buckets of 6 - 13 x the mean, their cells overflow and the column fell to 4 - 8 LSD
// passes (24 - 48 ms per 1e9 int64 keys against 11; cub's passes behind cudf::sort do not care: cpp/src/sort/sort_radix.cu:52-161).
// When the sample's level-0 histogram says so (k_hf_plan stage 1), level 0 cuts on <= 255 sample-chosen SPLITTERS instead:
//   * k_sp_plan sorts 16384 sampled keys in one workgroup; every 64th is a splitter; a value that fills two quantiles gets an
//     EQUALITY bucket [v, v + 1) (its keys need no sorting: level 1 copies them straight to the output);
//   * bucket(key) = number of splitters <= key: a 2048-entry LUT over (key - min), cut linearly or logarithmically, gives the first
//     candidate, a short scan of the sorted table the exact bucket;
//   * inside bucket b -- keys in [lo_b, lo_b + w_b), about n / 256 of them, evenly dense when the density is smooth -- cells are EQUAL
//     WIDTH slices: frac = (key - lo_b) / w_b as a 32-bit fixed-point fraction (one 32 x 32 multiply), cell = its top bits2 bits,
//     and the cell sort's 13-bit counting digit the 13 bits below.  Both maps are monotone, which is all the levels need; keys
//     that crowd a cell anyway go through the big-cell path (X + LSD passes) as before.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int SP_NSAMP = 16384;  // sampled keys, sorted by one workgroup (128 KiB of LDS)
constexpr int SP_NLUT  = 4096;  // (2048 until run 19: U[0, 1) doubles -- half the splitters in one binade -- had 4 in the fullest cell: four search steps per key)
constexpr int SP_NPIECE = 16;  // pieces of a bucket's range whose sampled masses shape the bucket's cell map (sp_warp)
constexpr int SP_PSH    = 28;  // piece of a fraction = frac >> SP_PSH
struct SplitPlan {
  int32_t req;             // k_hf_plan stage 1 asks for splitters
  int32_t on;              // k_sp_plan has built the tables: level 0 / level 1 / the cell sort use them
  uint32_t nsp;            // splitters in use; buckets = nsp + 1 <= BINS
  uint32_t neq;            // equality buckets
  unsigned long long kmin; // LUT origin (the sample's smallest key)
  uint32_t lut_steps;            // splitters in the fullest LUT cell of the chosen form
  uint32_t lut_log, lshift;      // LUT form: 0 linear in (key - kmin) >> lshift, 1 logarithmic, 2 (float keys) one linear half per sign (SpLutF)
  unsigned long long f_nmin, f_pmin;  // form 2: the sample's smallest negative / smallest positive key (sortable form) ...
  uint32_t f_nsh, f_psh;              // ... and the shifts that lay each sign's sampled range over SP_NLUT / 2 cells
  unsigned long long tab[BINS];  // sorted splitters (sortable form), padded with ~0
  unsigned long long lo[BINS];   // bucket b maps keys in [lo, lo + w) to [0, 1): its range, clamped to the sample's at either end
  unsigned long long w[BINS];
  uint32_t mlow[BINS];           // frac = x * (2^32 + mlow) >> 31 with x = the normalised (key - lo)
  int32_t nsh[BINS];             // x = rel >> nsh (>= 0) or rel << -nsh: the range's last key becomes a 31-bit number with bit 30 set
  uint32_t eq[BINS];             // the bucket's TRUE range is one value
  uint32_t nc[BINS];             // cells of bucket b (k_hf_plan stage 2, from the EXACT level-0 histogram): about 7400 keys per cell whatever the
                                 // bucket's size -- a 16384-key sample balances the buckets to +- 12 %, and a power-of-two cell count sized for the
                                 // fullest bucket left every cell half empty (the cell sort costs a cell what it costs full: 5.4 against 3.3 ms)
  uint32_t pf[BINS];             // k_hf_plan stage 1: fullest / mean cell of the bucket x 256 as far as the n / 32 sample can tell -- what is left
                                 // of the density's slope INSIDE one piece of the warp (1.01 - 1.05 for smooth densities); cells and their slots
                                 // are sized for it.  (Until run 12 the cells were equal-width slices sized by a peak factor taken from the
                                 // median of ~68 sampled keys: 1.25 on average -- a quarter more cells than needed -- and still 1 - 2.5 % of a
                                 // bell-shaped / float column's keys overfilled the dense-end cells: scripts/xp/xp_split_warp_model.py.)
  uint32_t sub[BINS][SP_NPIECE]; // k_sp_sample: sampled keys per (bucket, sixteenth of the bucket's range)
  uint2 wt[BINS][SP_NPIECE];     // k_hf_plan stage 1: the bucket's WARP -- piece j of the fraction maps linearly onto [x, x + y) with y proportional
                                 // to the piece's sampled mass: a piecewise-linear estimate of the bucket's CDF, so that equal slices of the
                                 // warped fraction (the cells, and the cell sort's counting bins) hold equal numbers of keys
  uint32_t nw[BINS];             // values in the bucket's TRUE range when that is what lo / w describe and it is small (<= 65535), else 0: with
                                 // nw <= cells per bucket every cell holds ONE value -- a NARROW bucket is counted and filled, never sorted
  uint16_t lut[SP_NLUT];         // splitters in LUT cells below c | 0x8000 when cell c holds none (the bucket is then known)
};
struct SpCell {
  unsigned long long lo, w;
  uint32_t mlow;
  int nsh;  // (as planned: x = rel >> nsh, or rel << -nsh; sp_frac uses the one left shift 33 - nsh that does both)
};
__device__ __forceinline__ SpCell sp_cell_of(const SplitPlan& sp, uint32_t b) { return SpCell{sp.lo[b], sp.w[b], sp.mlow[b], sp.nsh[b]}; }
// (bits2 = the level-1 bits the launches are sized for: 1 << bits2 cell slots per bucket)
__device__ __forceinline__ bool sp_narrow(const SplitPlan& sp, uint32_t b, int bits2) { return sp.nw[b] != 0u && sp.nw[b] <= (1u << bits2) && !sp.eq[b]; }
// cell of a key inside its bucket and, below it, the CL2-bit counting digit of the cell sort: both from frac * cells
__device__ __forceinline__ uint32_t sp_cell(uint32_t frac, uint32_t nc) { return __umulhi(frac, nc); }
// (one v_mul_hi_u32 each -- 32-bit multiplies run at a quarter of the VALU rate, and the 64-bit product these are written as costs two:
//  (frac * nc) >> (32 - CL2) = mulhi(frac, nc << CL2) while nc << CL2 < 2^32 -- nc <= 1024, CL2 = 13)
template <int CL2>
__device__ __forceinline__ uint32_t sp_fine(uint32_t frac, uint32_t nc) { return __umulhi(frac, nc << CL2) & ((1u << CL2) - 1u); }
// the bucket's warp (SplitPlan::wt, staged in LDS by the caller): monotone -- piece j ends below x[j] + y[j] <= x[j + 1] -- and the
// IDENTITY for the table {j << 28, 1 << 28} that narrow buckets get (their cell map must stay injective: k_sp_fill inverts it)
__device__ __forceinline__ uint32_t sp_warp(uint32_t frac, const uint2* wt)
{
  const uint2 e = wt[frac >> SP_PSH];
  return e.x + __umulhi(frac << (32 - SP_PSH), e.y);  // = ((frac & (2^28 - 1)) * e.y) >> 28
}
// position of a key inside its bucket's range as a 32-bit fraction; monotone; keys outside the range clamp to its ends
__device__ __forceinline__ uint32_t sp_frac(const SpCell& c, unsigned long long key)
{
  const unsigned long long rel = key > c.lo ? key - c.lo : 0ull;
  if (rel >= c.w) return 0xFFFFFFFFu;
  // x = rel >> nsh (nsh >= 0) or rel << -nsh: the range's last key becomes a 31-bit number with bit 30 set.  Both are
  // (rel << (33 - nsh)) >> 33 -- rel < 2^(31 + nsh), so the left shift loses nothing -- i.e. the high word of ONE shift, halved
  // (computing both shifts and selecting cost three 64-bit instructions per key; nsh = 63 marks a range of one value: x = 0)
  const uint32_t x2 = c.nsh == 63 ? 0u : ((uint32_t)((rel << (33 - c.nsh)) >> 32) & ~1u);  // = x << 1
  return x2 + __umulhi(x2, c.mlow);                                                          // = 2 x + (x * mlow) >> 31
}
__device__ __forceinline__ uint32_t sp_lut_cell(unsigned long long rel, uint32_t lut_log, uint32_t lshift)
{
  if (lut_log) {  // exponent and 6 mantissa bits of rel (64 x 64 = SP_NLUT cells): power-law densities
    if (rel == 0) return 0;
    const int e      = 63 - __builtin_clzll(rel);
    const uint32_t m = e >= 6 ? (uint32_t)(rel >> (e - 6)) & 63u : (uint32_t)(rel << (6 - e)) & 63u;
    return (uint32_t)e * 64u + m;
  }
  const unsigned long long c = rel >> lshift;
  return c < (unsigned long long)(SP_NLUT - 1) ? (uint32_t)c : (uint32_t)(SP_NLUT - 1);
}
// Form 2: TWO linear halves, cut at the widest gap between neighbouring splitters.  Float keys of both signs: the sortable forms of -x
// and +x lie 2^63 apart with nothing between -tiny and +tiny, and one binade is 2^52 of 2^64 -- a linear LUT over the whole range gives
// a binade two cells, the logarithmic one is no better: N(0, 1) doubles had 61 / 79 splitters in the fullest cell and 97 % / 50 % of the
// keys paid a bisection (level 0: 7.3 ms against 5.1 for U[0, 1); 5.3 with this form, run 13).  Integer keys around two far-apart
// centres: a cluster's 125 splitters in one cell of either form (level 0: 6.2 ms).  Each side of the gap gets half the cells, linear
// over its own splitters' range.  (Run 12's four-segment form cost EVERY column 0.7 ms at level 0: eleven 64-bit parameters and a
// 4-way select per key.  This one is two parameters and one compare, behind a block-uniform test of the form.)
struct SpLutF {
  unsigned long long nmin, pmin;  // origin of the lower half (the sample's smallest key) | first key of the upper half (the splitter above the gap)
  uint32_t nsh, psh;
};
__device__ __forceinline__ SpLutF sp_lutf_of(const SplitPlan& sp) { return SpLutF{sp.f_nmin, sp.f_pmin, sp.f_nsh, sp.f_psh}; }
// F2: 1 = the caller knows the form is 2, 0 = knows it is not (k_sp_level0 picks its ranking loop once per tile -- with the test inside
// the loop, run 14, every split-mode column paid 0.1 - 0.15 ms for a form most of them do not use), -1 = look at `form`
template <int KIND, int F2 = -1>
__device__ __forceinline__ uint32_t sp_lut_cell_k(unsigned long long key, unsigned long long kmin, uint32_t form, uint32_t lshift, const SpLutF& f)
{
  if (F2 == 1 || (F2 < 0 && form == 2u)) {
    const bool pos             = key >= f.pmin;
    const unsigned long long o = pos ? f.pmin : f.nmin;
    const unsigned long long c = (key > o ? key - o : 0ull) >> (pos ? f.psh : f.nsh);
    return (pos ? (uint32_t)(SP_NLUT / 2) : 0u) + (c < (unsigned long long)(SP_NLUT / 2 - 1) ? (uint32_t)c : (uint32_t)(SP_NLUT / 2 - 1));
  }
  return sp_lut_cell(key >= kmin ? key - kmin : 0ull, form, lshift);
}
// bucket = number of splitters <= key (exact for every key: the LUT only says where the search starts).  LUT word: bits 0-8 the
// splitters in cells below, bits 9-14 the splitters INSIDE the cell (capped at 63), bit 15 "none inside" (the bucket is then known).
// A few inside: a short scan; many (two far-apart clusters put a whole cluster's splitters into one cell of either LUT form:
// 46 ms for level 0 in the first run): a bisection of that stretch of the sorted table.
template <int KIND, int F2 = -1>
__device__ __forceinline__ uint32_t sp_bucket(const unsigned long long* __restrict__ tab, const uint16_t* __restrict__ lut, unsigned long long key,
                                              uint32_t nsp, unsigned long long kmin, uint32_t lut_log, uint32_t lshift, const SpLutF& lf)
{
  const uint32_t wd            = lut[sp_lut_cell_k<KIND, F2>(key, kmin, lut_log, lshift, lf)];
  uint32_t b                   = wd & 0x1FFu;
  if (wd & 0x8000u) return b;
  const uint32_t inside = (wd >> 9) & 63u;
  if (inside <= 4u) {
    while (b < nsp && tab[b] <= key) ++b;
    return b;
  }
  uint32_t lo = b, hi = inside == 63u ? nsp : b + inside;  // the answer lies in [lo, hi]: first index in [lo, hi) whose splitter is > key
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (tab[mid] <= key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Device-resident plan, first bytes of the caller's scratch.
struct SortPlan {
  uint32_t hist[MAX_PASSES][BINS];  // digit histograms of the whole column
  uint32_t gbin[MAX_PASSES][BINS];  // exclusive scan of hist: global base of each bin
  int32_t pass_skip[MAX_PASSES];
  int32_t pass_src[MAX_PASSES];  // buffer selector: 0 = input, 1 = output (A), 2 = scratch (B)
  int32_t pass_dst[MAX_PASSES];
  int32_t num_active;
  int32_t status;  // 0 ok, 3 hybrid bookkeeping mismatch (the LSD passes then produced the output), 5 (GX_SORT_SPIN_FAULT) a look-back wait
                   // was abandoned (spin_guard): the output is NOT sorted; gx_sort_status reports it
  HybridPlan hy;
  SortCounters cnt;
  FastPlan hf;
  SplitPlan sp;
};

// level-0 digit of a sortable key: ((k >> shift0) & dm) | ((k >> fsh) & fhi), where (dm, fsh, fhi) = (0xFF, 0, 0), or
// (0x7F, width - 8, 0x80) under the sign fold (HybridPlan::fold)
struct Digit0 {
  int shift, fsh;
  uint32_t dm, fhi;
  template <typename KeyT>
  __device__ __forceinline__ uint32_t operator()(KeyT k) const { return ((uint32_t)(k >> shift) & dm) | ((uint32_t)(k >> fsh) & fhi); }
};
__device__ __forceinline__ Digit0 digit0_of(const HybridPlan& hy, int key_bits)
{
  return hy.fold ? Digit0{hy.shift0, key_bits - 8, 0x7Fu, 0x80u} : Digit0{hy.shift0, 0, 0xFFu, 0u};
}
// (sign fold) bits [h, sign) of every key are copies of its sign: h from OR(key ^ sign extension); 0 = do not fold
__device__ __forceinline__ int fold_height(unsigned long long fold_or, unsigned long long V, int key_bits)
{
  if (!((V >> (key_bits - 1)) & 1ull)) return 0;             // the sign does not vary: the plain masks say everything
  const int h = fold_or ? 64 - __builtin_clzll(fold_or) : 0;
  return (h >= 7 && h <= key_bits - 10) ? h : 0;              // >= 9 constant bits between the data and the sign, 7 bits of data for the digit
}

// ------------------------------------------------------------------------------------------
// up-front histogram of every digit
// ------------------------------------------------------------------------------------------
template <typename KeyT, int KIND>
__global__ void __launch_bounds__(BT) k_hist_all(const KeyT* __restrict__ in, int64_t n, KeyT desc_mask,
                                                 SortPlan* plan, int64_t range_rows, const KeyT* __restrict__ inB = nullptr)
{
  // block b histograms rows of input range b % NRANGE (range r = rows [r, r+1) * range_rows); k_plan
  // sums the ranges.  The hybrid's first partition pass runs one look-back chain per range.
  if (plan->hy.ok && !plan->hy.lsd_mode) return;  // the hybrid path sorted the column: no LSD pass will run
  if (plan->hy.lsd_mode) {  // the LSD passes sort X, the compacted big cells of the cursor path (HybridPlan::big)
    in         = inB;
    n          = (int64_t)plan->hy.lsd_n;
    range_rows = div_up(div_up(n, (int64_t)NRANGE), (int64_t)GX_WAVE) * GX_WAVE;
  }
  const int range      = blockIdx.x % NRANGE;
  const int64_t rbegin = (int64_t)range * range_rows < n ? (int64_t)range * range_rows : n;
  const int64_t rend   = (range == NRANGE - 1) ? n : (rbegin + range_rows < n ? rbegin + range_rows : n);
  constexpr int NPASS = sizeof(KeyT);
  __shared__ uint32_t s_hist[NPASS * BINS];
  __shared__ unsigned long long s_fold[NW];
  for (int i = threadIdx.x; i < NPASS * BINS; i += BT) s_hist[i] = 0;
  __syncthreads();
  const unsigned lane  = lane_id();
  KeyT fold            = 0;  // K_SIGNED: bits in which some key differs from its own sign extension
  constexpr int UNROLL = 4;  // independent loads in flight per lane (>= 32 KiB per CU at full occupancy)
  const int64_t stride = (int64_t)(gridDim.x / NRANGE) * BT * UNROLL;
  for (int64_t i0 = rbegin + (int64_t)(blockIdx.x / NRANGE) * BT * UNROLL + threadIdx.x; i0 < rend; i0 += stride) {
    KeyT raw[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = i0 + (int64_t)u * BT;
      raw[u]          = (i < rend) ? in[i] : KeyT(0);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = i0 + (int64_t)u * BT;
      if (i < rend) {
        const KeyT k          = to_sortable<KeyT, KIND>(raw[u], desc_mask);
        if (KIND == K_SIGNED) fold |= (KeyT)(raw[u] ^ (KeyT)(KeyT(0) - (KeyT)(raw[u] >> (8 * sizeof(KeyT) - 1))));
        const uint64_t active = ballot(true);
        const int leader      = __builtin_ctzll(active);
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
          const uint32_t d  = (uint32_t)(k >> (8 * p)) & 0xFFu;
          const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
          const uint64_t same = ballot(d == d0);
          if (same == active) {  // whole wave hits one bin: one add instead of 64 conflicts
            if ((int)lane == leader) atomicAdd(&s_hist[p * BINS + d0], (uint32_t)__builtin_popcountll(active));
          } else if (__builtin_popcountll(same) >= 8) {  // a popular digit (skewed keys: 87 % of a Zipf column share their upper bytes)
            if ((int)lane == leader) atomicAdd(&s_hist[p * BINS + d0], (uint32_t)__builtin_popcountll(same));
            else if (d != d0) atomicAdd(&s_hist[p * BINS + d], 1u);
          } else {
            atomicAdd(&s_hist[p * BINS + d], 1u);
          }
        }
      }
    }
  }
  if (KIND == K_SIGNED) {
    const unsigned long long wf = wave_reduce((unsigned long long)fold, [](unsigned long long x, unsigned long long y) { return x | y; });
    if (lane == 0) s_fold[threadIdx.x / GX_WAVE] = wf;
  }
  __syncthreads();
  if (KIND == K_SIGNED && threadIdx.x == 0) {
    unsigned long long f = 0;
    for (int k = 0; k < NW; ++k) f |= s_fold[k];
    if (f & ~__hip_atomic_load(&plan->hy.fold_or, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&plan->hy.fold_or, f);
  }
  for (int i = threadIdx.x; i < NPASS * BINS; i += BT) {
    const uint32_t c = s_hist[i];
    if (c) atomicAdd(&plan->hy.rhist[range][i / BINS][i % BINS], c);
  }
}

// ------------------------------------------------------------------------------------------
// hybrid path, up-front pass: ONE read of the keys (8 B/row) with one LDS atomic per key.
//   SPEC = true : OR of the sortable keys and OR of their complements (a bit varies over the column iff it is
//                 set in both) + range-resolved histogram of the TOP byte -- the level-0 digit of any column
//                 whose top bit varies (full-range integers, mixed-sign doubles);
//   SPEC = false: range-resolved histogram of the bit-granular level-0 digit k_hy_plan chose; runs only when
//                 that is not the top byte (hy.need_hist), otherwise exits at once.
// Block b histograms rows of input range b % NRANGE (range r = rows [r, r+1) * range_rows): the first
// partition pass runs one look-back chain per range.
// ------------------------------------------------------------------------------------------
template <typename KeyT, int KIND, bool SPEC>
__global__ void __launch_bounds__(BT) k_hy_hist(const KeyT* __restrict__ in, int64_t n, KeyT desc_mask, SortPlan* plan,
                                                int64_t range_rows)
{
  HybridPlan& hy = plan->hy;
  if (plan->hf.state >= 3) return;  // the cursor path has done levels 0 and 1 (3) / has ruled the hybrid path out (4)
  if (!SPEC && !(hy.attempt && hy.need_hist)) return;
  const Digit0 dig     = SPEC ? Digit0{(int)(8 * sizeof(KeyT) - 8), 0, 0xFFu, 0u} : digit0_of(hy, (int)(8 * sizeof(KeyT)));
  const int range      = blockIdx.x % NRANGE;
  const int64_t rbegin = (int64_t)range * range_rows < n ? (int64_t)range * range_rows : n;
  const int64_t rend   = (range == NRANGE - 1) ? n : (rbegin + range_rows < n ? rbegin + range_rows : n);
  __shared__ uint32_t s_hist[BINS];
  __shared__ unsigned long long s_red[2 * NW];
  for (int i = threadIdx.x; i < BINS; i += BT) s_hist[i] = 0;
  __syncthreads();
  const unsigned lane  = lane_id();
  constexpr int UNROLL = 8;  // independent 8-byte loads in flight per lane
  const int64_t stride = (int64_t)(gridDim.x / NRANGE) * BT * UNROLL;
  KeyT vor = 0, vnor = 0, vfold = 0;
  for (int64_t i0 = rbegin + (int64_t)(blockIdx.x / NRANGE) * BT * UNROLL + threadIdx.x; i0 < rend; i0 += stride) {
    KeyT raw[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = i0 + (int64_t)u * BT;
      raw[u]          = (i < rend) ? in[i] : KeyT(0);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = i0 + (int64_t)u * BT;
      if (i < rend) {
        const KeyT k = to_sortable<KeyT, KIND>(raw[u], desc_mask);
        if (SPEC) {
          vor |= k;
          vnor |= (KeyT)~k;
          if (KIND == K_SIGNED) vfold |= (KeyT)(raw[u] ^ (KeyT)(KeyT(0) - (KeyT)(raw[u] >> (8 * sizeof(KeyT) - 1))));
        }
        const uint64_t active = ballot(true);
        const int leader      = __builtin_ctzll(active);
        const uint32_t d      = dig(k);
        const uint32_t d0     = __builtin_amdgcn_readfirstlane(d);
        if (ballot(d == d0) == active) {  // whole wave hits one bin: one add instead of 64 conflicts
          if ((int)lane == leader) atomicAdd(&s_hist[d0], (uint32_t)__builtin_popcountll(active));
        } else {
          atomicAdd(&s_hist[d], 1u);
        }
      }
    }
  }
  if (SPEC) {
    const unsigned long long wo = wave_reduce((unsigned long long)vor, [](unsigned long long x, unsigned long long y) { return x | y; });
    const unsigned long long wn = wave_reduce((unsigned long long)vnor, [](unsigned long long x, unsigned long long y) { return x | y; });
    if (lane == 0) {
      s_red[threadIdx.x / GX_WAVE]      = wo;
      s_red[NW + threadIdx.x / GX_WAVE] = wn;
    }
  }
  __syncthreads();
  if (SPEC && threadIdx.x == 0) {
    unsigned long long o = 0, no = 0;
    for (int k = 0; k < NW; ++k) {
      o |= s_red[k];
      no |= s_red[NW + k];
    }
    atomicOr(&hy.or_mask, o);
    atomicOr(&hy.nor_mask, no);
  }
  if (SPEC && KIND == K_SIGNED) {
    const unsigned long long wf = wave_reduce((unsigned long long)vfold, [](unsigned long long x, unsigned long long y) { return x | y; });
    if (lane == 0 && (wf & ~__hip_atomic_load(&hy.fold_x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) atomicOr(&hy.fold_x, wf);
  }
  for (int i = threadIdx.x; i < BINS; i += BT) {
    const uint32_t c = s_hist[i];
    if (c) atomicAdd(&hy.rh0[range][i], c);
  }
}

// per-bucket cell slots (HybridPlan::ccap / cbase), cursor path: every thread of a BINS-thread block calls it with its bucket's
// key count.  A bucket at the edge of the key range is only partly covered by keys (keys in [0, 1e12): the top bin is 83 % full),
// its cells are as dense as its neighbours', so a bucket takes the LARGEST mean of itself and its two neighbours -- when the
// total fits `budget` keys (the buffer the host made: n + slack per cell + n / 16); otherwise its own mean, which always fits.
__device__ __forceinline__ void plan_cell_slots(HybridPlan& hy, uint32_t bucket_count, int bits2, int cell_max, uint32_t* s_tmp,
                                                unsigned long long budget, uint32_t ncells_given = 0u, uint32_t true_count = 0xFFFFFFFFu)
{
  // ncells_given (splitter mode): the cells this bucket uses (SplitPlan::nc) when that is not all of the 1 << bits2 slots;
  // bucket_count is then the bucket's size x its fullest / mean cell (SplitPlan::pf) and true_count its size: slots sized for the
  // fullest cell need not fit the buffer -- the third choice, slots for the mean cell, always does
  __shared__ uint32_t s_mean[BINS];
  __shared__ unsigned long long s_tmp64[BINS / GX_WAVE + 1];
  (void)s_tmp;
  const int t           = threadIdx.x;
  const uint32_t ncells = ncells_given ? ncells_given : (1u << bits2);
  const uint32_t mean   = bucket_count / ncells + 1u;
  s_mean[t]             = mean;
  __syncthreads();
  uint32_t m3 = mean;
  if (t > 0 && s_mean[t - 1] > m3) m3 = s_mean[t - 1];
  if (t < BINS - 1 && s_mean[t + 1] > m3) m3 = s_mean[t + 1];
  auto cap_of = [&](uint32_t m) {
    uint32_t cap = m + (uint32_t)(6.0f * __builtin_sqrtf((float)m)) + 64u;
    cap          = (cap + 15u) & ~15u;
    return cap > (uint32_t)cell_max ? (uint32_t)cell_max : cap;
  };
  const uint32_t cap_a = cap_of(m3), cap_b = cap_of(mean);
  // (sums in 64 bits: a splitter-mode column of steep buckets can ask for more than 2^32 keys of slots)
  unsigned long long total_a, total_b;
  const unsigned long long base_a = block_exclusive_scan<BINS>((unsigned long long)cap_a * ncells, 0ull, SumOp(), s_tmp64, &total_a);
  const unsigned long long base_b = block_exclusive_scan<BINS>((unsigned long long)cap_b * ncells, 0ull, SumOp(), s_tmp64, &total_b);
  const bool smooth = total_a <= budget;
  uint32_t cap = smooth ? cap_a : cap_b, base = (uint32_t)(smooth ? base_a : base_b);
  if (true_count != 0xFFFFFFFFu) {  // block-uniform (splitter mode)
    const uint32_t cap_c            = cap_of(true_count / ncells + 1u);
    const unsigned long long base_c = block_exclusive_scan<BINS>((unsigned long long)cap_c * ncells, 0ull, SumOp(), s_tmp64, (unsigned long long*)nullptr);
    if (!smooth && total_b > budget) {
      cap  = cap_c;
      base = (uint32_t)base_c;
    }
  }
  hy.ccap[t]  = cap;
  hy.cbase[t] = base;
}

// digits of the local sort's stable LDS passes: the bytes below shift2 that vary somewhere in the column
__device__ __forceinline__ void plan_local_digits(HybridPlan& hy, unsigned long long V, int shift2)
{
  int nl = 0;
  for (int sft = 0; sft < shift2; sft += 8) {
    const int bits = shift2 - sft < 8 ? shift2 - sft : 8;
    if (bits == 8 && ((V >> sft) & 0xFFull) == 0) continue;  // constant byte: nothing to sort on
    hy.lshift[nl] = sft;
    hy.lbits[nl]  = bits;
    ++nl;
  }
  hy.nlocal = nl;
}

// One block of 256 threads.  STAGE 0 (after k_hy_hist<SPEC>): digits from the varying-bit mask.  STAGE 1 (after
// k_hy_hist<!SPEC>): level-0 histogram totals, bin bases per input range, level-0 segment tables.
__global__ void __launch_bounds__(BINS) k_hy_plan(SortPlan* plan, int stage, int key_bits, int64_t n, int bits2, int cell_max,
                                                  int pos_bits, int64_t range_rows, int tile_rows, uint32_t* base1, int cell_alt = 0,
                                                  int signed_keys = 0)
{
  // cell_alt (round 4; sorted_order's pairs): a larger cell capacity (16384) the launches behind are also prepared for.  bits2 comes
  // from n alone; keys whose range is not a power of two have fuller buckets than n / 256, their 8192-key cells overflow and the
  // column fell to the LSD pair passes (91 ms per 1e9 rows, profiles/r4_run8_bench_sorted_order_range1e12.jsonl).  Stage 1 has the
  // exact level-0 histogram: when a bucket's mean cell would not fit it switches the whole sort to the larger cells (one
  // workgroup per CU in the cell sort: slower than 8192-key cells, several times faster than the LSD passes).
  __shared__ uint32_t s_tmp[BINS / GX_WAVE + 1];
  HybridPlan& hy = plan->hy;
  const int t    = threadIdx.x;
  if (plan->hf.state >= 3) return;  // the cursor path owns the plan (3) / has ruled the hybrid path out (4: attempt stays 0)
  if (stage == 0) {
    const unsigned long long V = hy.or_mask & hy.nor_mask;  // bits that differ somewhere in the column
    if (V == 0) {
      if (t == 0) hy.attempt = 0;
      return;
    }
    const int fh     = signed_keys ? fold_height(hy.fold_x, V, key_bits) : 0;  // (k_hy_hist<SPEC> read every key: exact)
    const int top    = fh ? fh : 63 - __builtin_clzll(V);
    const int shift0 = top - 7;
    const int shift2 = shift0 - bits2;
    // the local sort splits a cell on >= 7 further bits in LDS; the packed (key bits, position) words of a pairs
    // sort must fit 64 bits (float keys-only sorts fall back to plain keys + stable LDS passes by themselves)
    const bool ok = shift2 >= 8 && (pos_bits == 0 || shift2 + pos_bits <= 64);
    const bool spec_ok = shift0 == key_bits - 8;  // (never under the sign fold: fh <= key_bits - 10)
    if (!ok) {
      if (t == 0) hy.attempt = 0;
      return;
    }
    if (!spec_ok) {  // the speculative histogram is of the wrong digit: k_hy_hist<false> refills rh0
      for (int r = 0; r < NRANGE; ++r) hy.rh0[r][t] = 0;
    }
    if (t == 0) {
      hy.attempt   = 1;
      hy.shift0    = shift0;
      hy.fold      = fh ? 1 : 0;
      hy.bits2     = bits2;
      hy.shift2    = shift2;
      hy.cell_max  = cell_max;
      hy.need_hist = spec_ok ? 0 : 1;
      plan_local_digits(hy, V, shift2);
    }
    return;
  }
  if (!hy.attempt) return;
  uint32_t c = 0;
  for (int r = 0; r < NRANGE; ++r) c += hy.rh0[r][t];
  const uint32_t exc = block_exclusive_scan<BINS>(c, 0u, SumOp(), s_tmp, (uint32_t*)nullptr);
  hy.hist0[t] = c;
  hy.gbin0[t] = exc;
  if (cell_alt > cell_max) {
    const int full = ((unsigned long long)c >> hy.bits2) > (unsigned long long)(0.97 * (double)cell_max) ? 1 : 0;
    int alt_bits   = 0;
    while ((1 << alt_bits) < cell_alt) ++alt_bits;
    if (__syncthreads_or(full) && t == 0 && (pos_bits == 0 || hy.shift2 + alt_bits <= 64)) hy.cell_max = cell_alt;
  }
  // (the look-back path keeps the fixed cell slots -- ccap stays 0: it has no rescue for a cell that outgrows its slot, and the
  //  per-bucket capacities of the cursor path assume an even density inside a bucket, which float keys do not have)
  __syncthreads();
  // a bucket with more keys than all of its cells hold overflows whatever its keys look like (keys spread around zero: two
  // buckets of n / 2; bell-shaped or Zipf-like values): decline here, before the two partition passes are spent on it
  if (__syncthreads_or((unsigned long long)c > ((unsigned long long)hy.cell_max << hy.bits2) ? 1 : 0)) {
    if (t == 0) hy.attempt = 0;
    return;
  }
  uint32_t run = exc;
  for (int r = 0; r < NRANGE; ++r) {  // level-0 output base of bin t for every input range
    base1[r * NB2MAX + t] = run;
    run += hy.rh0[r][t];
  }
  {  // level 1: bucket t is one segment with its own look-back chain; lists of 32 buckets per XCD
    const uint32_t tiles1 = (c + (uint32_t)tile_rows - 1) / (uint32_t)tile_rows;
    uint32_t total1;
    const uint32_t t0 = block_exclusive_scan<BINS>(tiles1, 0u, SumOp(), s_tmp, &total1);
    hy.seg_start[1][t] = exc;
    hy.seg_count[1][t] = c;
    hy.seg_tile0[1][t] = t0;
    if (t % (BINS / NRANGE) == 0) hy.list_tile0[1][t / (BINS / NRANGE)] = t0;
    if (t == 0) {
      hy.seg_tile0[1][BINS]    = total1;
      hy.list_tile0[1][NRANGE] = total1;
    }
  }
  if (t == 0) {
    uint32_t tiles = 0;
    for (int r = 0; r < NRANGE; ++r) {
      const int64_t b = (int64_t)r * range_rows < n ? (int64_t)r * range_rows : n;
      const int64_t e = (r == NRANGE - 1) ? n : (b + range_rows < n ? b + range_rows : n);
      hy.seg_start[0][r]  = (uint32_t)b;
      hy.seg_count[0][r]  = (uint32_t)(e - b);
      hy.seg_tile0[0][r]  = tiles;
      hy.list_tile0[0][r] = tiles;
      tiles += (uint32_t)((e - b + tile_rows - 1) / tile_rows);
    }
    hy.seg_tile0[0][NRANGE]  = tiles;
    hy.list_tile0[0][NRANGE] = tiles;
  }
}

// one block of 256 threads: plan of the LSD passes
__global__ void __launch_bounds__(BINS) k_plan(SortPlan* plan, int npass, int64_t n, int signed_keys = 0)
{
  // signed_keys (round 4): integers spread around zero -- small signed values, differences, anything in [-a, b) -- have upper bytes
  // that are pure SIGN EXTENSION: 0x00 for the non-negative keys, 0xFF for the negative ones.  Such a byte takes two values, so it
  // is not trivial, yet it carries nothing the top byte does not: when every bit from position h up to the sign bit equals the
  // key's sign (fold_or < 2^h, reduced by k_hist_all over exactly the keys being sorted), the stable passes over the bytes below
  // h order the keys inside either sign class and the pass on the top byte -- also two values -- puts the classes in order.
  // The passes on the bytes in between are skipped: keys in [-1000, 1000) take 3 passes instead of 8 (run 14: 57 ms per 1e9).
  __shared__ uint32_t s_tmp[BINS / GX_WAVE + 1];
  __shared__ int s_skip[MAX_PASSES];
  if (plan->hy.ok && !plan->hy.lsd_mode) return;  // the hybrid path sorted the column (k_plan2 marked every pass as skipped)
  if (plan->hy.lsd_mode) n = (int64_t)plan->hy.lsd_n;
  const int t = threadIdx.x;
  for (int p = 0; p < npass; ++p) {
    uint32_t c = 0;
    for (int r = 0; r < NRANGE; ++r) c += plan->hy.rhist[r][p][t];
    plan->hist[p][t] = c;
    int triv         = __syncthreads_or(c == (uint32_t)n);
    if (signed_keys && p < npass - 1) {
      const unsigned long long f = plan->hy.fold_or;
      const int h                = f ? 64 - __builtin_clzll(f) : 0;  // bits [h, sign) are copies of the sign in every key
      if (8 * p >= h) triv = 1;
    }
    uint32_t exc     = block_exclusive_scan<BINS>(c, 0u, SumOp(), s_tmp, (uint32_t*)nullptr);
    plan->gbin[p][t] = exc;
    if (t == 0) s_skip[p] = triv ? 1 : 0;
  }
  __syncthreads();
  if (t == 0) {
    int k = 0;
    for (int p = 0; p < npass; ++p) k += s_skip[p] ? 0 : 1;
    plan->num_active = k;
    int i = 0, prev_dst = 0;
    for (int p = 0; p < npass; ++p) {
      plan->pass_skip[p] = s_skip[p];
      if (s_skip[p]) {
        plan->pass_src[p] = plan->pass_dst[p] = 0;
        continue;
      }
      ++i;
      const int dst     = ((k - i) % 2 == 0) ? 1 : 2;  // the last active pass writes buffer 1 (output)
      plan->pass_src[p] = (i == 1) ? 0 : prev_dst;
      plan->pass_dst[p] = dst;
      prev_dst          = dst;
    }
  }
}

// ------------------------------------------------------------------------------------------
// per-pass scatter
// ------------------------------------------------------------------------------------------
struct PassArgs {
  void* kbuf[3];
  void* kbufB[3];     // the buffers of the X sort (HybridPlan::lsd_mode): X | work 1 (the sorted X ends here) | work 2
  uint32_t* vbuf[3];  // vbuf[0] may be null: iota
  SortPlan* plan;
  unsigned long long* status;  // [ntiles][256] look-back granules (algorithm 0)
  const uint32_t* tile_off;    // [256][ntiles] absolute offsets (algorithm 1)
  int64_t n;
  int64_t ntiles;
  int pass;
  int order_mode;  // experiment knob for the LBW == 0 kernel: 0 XCD-swizzled blockIdx, 1 plain blockIdx, 2 ticket
  uint64_t desc_mask;
  unsigned long long spin_ticks = 0;  // look-back wait limit in 100 MHz ticks (0: SPIN_SECONDS)
  long long inject_tile         = -1;  // TEST HOOK (gx_sort_inject_lost_tile): this tile never publishes its look-back granules
};

constexpr int PASS_TPB = 8;  // tickets per workgroup of the MULTI form of k_radix_pass
// A look-back wait that makes no progress for SPIN_SECONDS of WALL-CLOCK time (s_memrealtime: the 100 MHz constant clock, read
// every 4096 polls -- a slow or time-sliced predecessor tile is not a fault, however many polls it takes) can only mean a broken
// forward-progress chain.  Until round 5 that ended in __builtin_trap(): fatal for the process' HIP context.  Now the wait is
// ABANDONED: SortPlan::status becomes 5, the waiting bin takes what it has summed so far as its prefix -- never more than the
// true prefix, so every write of the tile stays inside its bin's (or cell's) range; the passes behind it do not run (k_radix_pass
// checks the word first: THEIR bin bases would no longer match what they read); the output is wrong, nothing outside it is
// touched -- and every other waiter that sees the flag gives up at its next check instead of sitting out its own 30 s.  The host
// side reads the word (gx_sort_status) and reports the failure: a wrong order is still never returned as success, and the caller's
// process survives (cudf::sort throws cudf::logic_error; the reference's analogue is a recoverable cudaError from cub,
// utilities/error.hpp:63-86 draws that line).
constexpr uint32_t SPIN_CHECK        = 1u << 12;
constexpr unsigned long long SPIN_SECONDS = 30;
constexpr int32_t SPIN_FAULT = 5;  // SortPlan::status
constexpr unsigned long long SPIN_SOFT_BIT = 1ull << 63;  // in the spin_ticks argument: abandon + report instead of trapping
__device__ __forceinline__ bool spin_guard(uint32_t& spins, unsigned long long& t0, SortPlan* plan, unsigned long long limit_ticks)
{
  if ((++spins & (SPIN_CHECK - 1)) != 0) return false;
  if (__hip_atomic_load(&plan->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == SPIN_FAULT) return true;
  const unsigned long long now = __builtin_amdgcn_s_memrealtime();
  if (t0 == 0) {
    t0 = now;
    return false;
  }
  const unsigned long long limit = limit_ticks & ~SPIN_SOFT_BIT;
  if (now - t0 <= (limit ? limit : SPIN_SECONDS * 100000000ull)) return false;
  __hip_atomic_store(&plan->status, SPIN_FAULT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // Round 6: the recoverable form is OPT-IN per call (gx_sort_set_fault_mode(1): the caller promises to read the
  // status word -- cudf::sort / sorted_order and ops.py do).  Every other caller of the sorts -- hash partitioning, rank, the sharded
  // operators -- never looks at the word, and a wrong order returned as success is worse than a dead process: they keep the trap.
  if (!(limit_ticks & SPIN_SOFT_BIT)) __builtin_trap();
  return true;
}

__device__ __forceinline__ unsigned long long pack_status(unsigned flag, unsigned epoch, uint32_t value)
{
  return ((unsigned long long)flag << 62) | ((unsigned long long)epoch << 32) | value;
}

// LBW: look-back window (0 = no look-back: offsets were precomputed by algorithm 1).  A thread owns
// one bin and inspects LBW predecessor tiles per round with LBW independent loads in flight: a
// status hop costs a memory-side round trip (~1.5-2.5 us under streaming load, the per-XCD L2s do
// not share lines), and a one-tile-per-hop walk settles into a regime where every tile walks
// ~10 predecessors (DESIGN.md "look-back regime"); the window bounds the walk to ~1-2 rounds.
// MULTI: the workgroup handles PASS_TPB consecutive tickets.  Used for the LSD passes that are enqueued
// behind a hybrid attempt: when the attempt succeeds they are no-ops, and the cost of a no-op launch
// is proportional to its grid (51 us for 122k single-tile workgroups at 1e9 keys).
template <typename KeyT, int KIND, bool HAS_VAL, int KPT, int LBW, bool MULTI = false>
__global__ void __launch_bounds__(BT, 4) k_radix_pass(PassArgs a)
{
  constexpr bool LOOKBACK = LBW > 0;
  constexpr int TILE = BT * KPT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  KeyT* s_keys       = reinterpret_cast<KeyT*>(smem);
  uint32_t* s_vals   = reinterpret_cast<uint32_t*>(smem + (size_t)TILE * sizeof(KeyT));
  uint32_t* s_whist  = s_vals + (HAS_VAL ? TILE : 0);  // [NW][256]
  uint32_t* s_gdelta = s_whist + NW * BINS;            // [256]
  uint32_t* s_scan   = s_gdelta + BINS;                // [NW + 1] (padded to 16)
  uint32_t* s_misc   = s_scan + 16;                    // [4]

  SortPlan* plan = a.plan;
  const int pass = a.pass;
  if (plan->pass_skip[pass]) return;  // constant digit: the pass would be the identity
  // after an abandoned look-back wait (spin_guard) no further pass runs: the faulted pass kept its writes inside the bins -- its
  // input was a permutation of the column, its prefixes were too small, never too large -- but its OUTPUT is not one any more (some
  // keys twice, some missing), so the next pass's digit counts would no longer match the up-front histograms its bin bases come
  // from, and its last bins would be written past the end of the buffer
  if (__hip_atomic_load(&plan->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == SPIN_FAULT) return;
  const int src_sel      = plan->pass_src[pass];
  const int dst_sel      = plan->pass_dst[pass];
  const bool mode_b      = plan->hy.lsd_mode != 0;  // X instead of the column, its length on the device
  const int64_t a_n      = mode_b ? (int64_t)plan->hy.lsd_n : a.n;
  const int64_t a_ntiles = mode_b ? div_up(a_n, (int64_t)(BT * KPT)) : a.ntiles;
  const KeyT* kin        = static_cast<const KeyT*>(mode_b ? a.kbufB[src_sel] : a.kbuf[src_sel]);
  KeyT* kout             = static_cast<KeyT*>(mode_b ? a.kbufB[dst_sel] : a.kbuf[dst_sel]);
  const uint32_t* vin    = HAS_VAL ? a.vbuf[src_sel] : nullptr;
  uint32_t* vout         = HAS_VAL ? a.vbuf[dst_sel] : nullptr;
  const KeyT desc_mask   = (KeyT)a.desc_mask;
  const int shift        = pass * 8;
  const unsigned tid     = threadIdx.x;
  const unsigned lane    = lane_id();
  const unsigned w       = tid / GX_WAVE;
  const unsigned epoch   = (unsigned)pass + 1u;

  // X mode: the grid is sized for the column, X is usually a sliver of it -- the surplus workgroups leave before they take a ticket
  // (a ticket is a device-scope atomic on ONE word: 15 000 of them per pass cost more than sorting a small X).  And when X has no
  // more tiles than the grid has workgroups, every workgroup takes ONE ticket: with PASS_TPB tiles handled back to back by the same
  // few workgroups a pass over 43 tiles took 80 us -- eight passes 0.65 ms, for 2e5 keys (profiles/r5_prof_f64_uniform_kernel_stats.txt)
  const bool one_each = MULTI && mode_b && a_ntiles <= (int64_t)gridDim.x;
  if (mode_b && (int64_t)blockIdx.x * ((MULTI && !one_each) ? PASS_TPB : 1) >= a_ntiles) return;
  for (int it = 0; it < (MULTI ? PASS_TPB : 1); ++it) {
  if (one_each && it > 0) return;  // (block-uniform)
  int64_t tile;
  if (LOOKBACK) {
    if (tid == 0) s_misc[0] = atomicAdd(&plan->cnt.tickets[pass].v, 1u);
    __syncthreads();
    tile = s_misc[0];
    if (tile >= a_ntiles) return;  // tickets only grow (MULTI; and the grid is sized for the column when X is sorted)
  } else if (a.order_mode == 2) {
    if (tid == 0) s_misc[0] = atomicAdd(&plan->cnt.tickets[pass].v, 1u);
    __syncthreads();
    tile = s_misc[0];
  } else {
    tile = a.order_mode == 1 ? (int64_t)blockIdx.x : xcd_swizzle(blockIdx.x, gridDim.x);
  }
  const int64_t base = tile * TILE;
  const int nvalid   = (int)((a_n - base < (int64_t)TILE) ? (a_n - base) : (int64_t)TILE);

  // ---- load (wave-striped: wave w owns a contiguous run, lanes consecutive -> 512 B per load)
  KeyT key[KPT];
  uint32_t val[HAS_VAL ? KPT : 1];
  const int wbase = (int)w * (KPT * GX_WAVE) + (int)lane;
  if (nvalid == TILE) {
#pragma unroll
    for (int j = 0; j < KPT; ++j) key[j] = kin[base + wbase + j * GX_WAVE];
    if (HAS_VAL) {
#pragma unroll
      for (int j = 0; j < KPT; ++j)
        val[j] = vin ? vin[base + wbase + j * GX_WAVE] : (uint32_t)(base + wbase + j * GX_WAVE);
    }
  } else {
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int idx = wbase + j * GX_WAVE;
      key[j]        = (idx < nvalid) ? kin[base + idx] : KeyT(0);
      if (HAS_VAL) val[j] = (idx < nvalid) ? (vin ? vin[base + idx] : (uint32_t)(base + idx)) : 0u;
    }
  }

  // ---- per-wave digit counters (each wave zeroes and owns its row: no barrier needed)
  uint32_t* my_hist = s_whist + w * BINS;
#pragma unroll
  for (int k = 0; k < BINS / GX_WAVE; ++k) my_hist[lane + k * GX_WAVE] = 0;

  // ---- stable ranking inside the wave: ballot match on the 8 digit bits, count lower lanes
  uint32_t packed[KPT];  // digit << 16 | rank inside (wave, digit)
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int idx = wbase + j * GX_WAVE;
    uint32_t d    = (uint32_t)(to_sortable<KeyT, KIND>(key[j], desc_mask) >> shift) & 0xFFu;
    if (idx >= nvalid) d = BINS - 1;  // padding sorts last (it is also last in input order)
    uint32_t lower, cnt;
    match_rank8(d, true, ~0ull, lower, cnt);
    const uint32_t prev = my_hist[d];
    if (lower == 0) my_hist[d] = prev + cnt;
    packed[j] = (d << 16) | (prev + lower);
  }
  __syncthreads();

  // ---- per-bin: exclusive prefix over waves, tile total, publish the aggregate early
  uint32_t tile_count = 0;
  if (tid < BINS) {
    uint32_t sum = 0;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) {
      const uint32_t c         = s_whist[w2 * BINS + tid];
      s_whist[w2 * BINS + tid] = sum;
      sum += c;
    }
    tile_count = sum;
  }
  uint32_t pub_count = tile_count;
  if (tid == BINS - 1) pub_count -= (uint32_t)(TILE - nvalid);  // padding is not data
  if (LOOKBACK && tid < BINS) {
    if (tile != a.inject_tile) store_agent_u64(&a.status[tile * BINS + tid], pack_status(tile == 0 ? 2u : 1u, epoch, pub_count));
  }
  const uint32_t bin_start = block_exclusive_scan<BT>(tile_count, 0u, SumOp(), s_scan, (uint32_t*)nullptr);
  if (tid < BINS) {
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) s_whist[w2 * BINS + tid] += bin_start;
  }
  __syncthreads();

  // ---- reorder the tile in LDS (digit-major, stable)
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const uint32_t d   = packed[j] >> 16;
    const uint32_t pos = my_hist[d] + (packed[j] & 0xFFFFu);
    s_keys[pos]        = key[j];
    if (HAS_VAL) s_vals[pos] = val[j];
  }

  // ---- global offset of each bin of this tile
  if (tid < BINS) {
    uint32_t gbase;
    if (LOOKBACK) {
      uint32_t prefix = 0;
      if (tile > 0) {
        int64_t p = tile - 1;
        bool done = false;
        while (!done) {
          constexpr int WN = LBW > 0 ? LBW : 1;
          unsigned long long v[WN];
#pragma unroll
          for (int k = 0; k < LBW; ++k) {
            const int64_t q = p - k;
            v[k]            = (q >= 0) ? load_agent_u64(&a.status[q * BINS + tid]) : pack_status(2u, epoch, 0u);
          }
#pragma unroll
          for (int k = 0; k < LBW; ++k) {
            if (!done) {
              unsigned long long x = v[k];
              uint32_t spins       = 0;
              unsigned long long spin_t0 = 0;
              while ((x >> 62) == 0 || ((unsigned)(x >> 32) & 0xFFu) != epoch) {
                if (spin_guard(spins, spin_t0, plan, a.spin_ticks)) {  // broken forward progress: flagged, never returned as a sorted column
                  x = pack_status(2u, epoch, 0u);
                  break;
                }
                __builtin_amdgcn_s_sleep(2);
                x = load_agent_u64(&a.status[(p - k) * BINS + tid]);
              }
              prefix += (uint32_t)x;
              if ((x >> 62) == 2u) done = true;
            }
          }
          p -= LBW;
        }
        if (tile != a.inject_tile) store_agent_u64(&a.status[tile * BINS + tid], pack_status(2u, epoch, prefix + pub_count));
      }
      gbase = plan->gbin[pass][tid] + prefix;
    } else {
      gbase = a.tile_off[(int64_t)tid * a_ntiles + tile];
    }
    s_gdelta[tid] = gbase - bin_start;
  }
  __syncthreads();

  // ---- write out: consecutive threads -> consecutive addresses inside a bin run
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int i = j * BT + (int)tid;
    if (i < nvalid) {
      const KeyT k       = s_keys[i];
      const uint32_t d   = (uint32_t)(to_sortable<KeyT, KIND>(k, desc_mask) >> shift) & 0xFFu;
      const uint32_t dst = s_gdelta[d] + (uint32_t)i;
      kout[dst]          = k;
      if (HAS_VAL) vout[dst] = s_vals[i];
    }
  }
  if (MULTI) __syncthreads();  // the next tile reuses the LDS
  }  // tiles of this workgroup
}

// algorithm 1: per-tile histogram of the current digit -> tile_hist[bin][tile]
template <typename KeyT, int KIND, int KPT>
__global__ void __launch_bounds__(BT) k_tile_hist(PassArgs a, uint32_t* tile_hist)
{
  constexpr int TILE = BT * KPT;
  __shared__ uint32_t s_hist[BINS];
  SortPlan* plan = a.plan;
  const int pass = a.pass;
  if (plan->pass_skip[pass]) return;
  const KeyT* kin      = static_cast<const KeyT*>(a.kbuf[plan->pass_src[pass]]);
  const KeyT desc_mask = (KeyT)a.desc_mask;
  const int shift      = pass * 8;
  const unsigned tid   = threadIdx.x;
  if (tid < BINS) s_hist[tid] = 0;
  __syncthreads();
  const int64_t tile = xcd_swizzle(blockIdx.x, gridDim.x);
  const int64_t base = tile * TILE;
  const int nvalid   = (int)((a.n - base < (int64_t)TILE) ? (a.n - base) : (int64_t)TILE);
  const unsigned lane = lane_id();
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int i = j * BT + (int)tid;
    if (i < nvalid) {
      const uint32_t d      = (uint32_t)(to_sortable<KeyT, KIND>(kin[base + i], desc_mask) >> shift) & 0xFFu;
      const uint64_t active = ballot(true);
      const uint32_t d0     = __builtin_amdgcn_readfirstlane(d);
      if (ballot(d == d0) == active) {
        if ((int)lane == __builtin_ctzll(active)) atomicAdd(&s_hist[d0], (uint32_t)__builtin_popcountll(active));
      } else {
        atomicAdd(&s_hist[d], 1u);
      }
    }
  }
  __syncthreads();
  if (tid < BINS) tile_hist[(int64_t)tid * a.ntiles + tile] = s_hist[tid];
}

// no active pass (constant column, or n <= 1): the sort is a copy
template <typename KeyT, bool HAS_VAL>
__global__ void __launch_bounds__(256) k_finalize_copy(PassArgs a)
{
  if (a.plan->num_active != 0) return;
  const bool mode_b    = a.plan->hy.lsd_mode != 0;
  const KeyT* kin      = static_cast<const KeyT*>(mode_b ? a.kbufB[0] : a.kbuf[0]);
  KeyT* kout           = static_cast<KeyT*>(mode_b ? a.kbufB[1] : a.kbuf[1]);
  const int64_t a_n    = mode_b ? (int64_t)a.plan->hy.lsd_n : a.n;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a_n; i += stride) {
    kout[i] = kin[i];
    if (HAS_VAL) a.vbuf[1][i] = a.vbuf[0] ? a.vbuf[0][i] : (uint32_t)i;
  }
}

// float descending, radix semantics: the NaN block comes first and must be in REVERSE input
// order (composite key (isnan*(idx+1), f) sorted descending: cpp/src/sort/sort_radix.cu:36-45).
// The stable sort left it in input order; find its length by bisection on the sorted output
// (NaN <=> magnitude bits above the exponent mask) and reverse it in place.
template <typename KeyT, bool HAS_VAL>
__global__ void __launch_bounds__(256) k_reverse_nan_block(KeyT* keys, uint32_t* vals, int64_t n)
{
  constexpr KeyT SIGN = KeyT(1) << (sizeof(KeyT) * 8 - 1);
  constexpr KeyT EXP  = (sizeof(KeyT) == 8) ? KeyT(0x7FF0000000000000ull) : KeyT(0x7F800000u);
  __shared__ long long s_cnt;
  if (threadIdx.x == 0) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = lo + (hi - lo) / 2;
      if ((keys[mid] & KeyT(~SIGN)) > EXP) lo = mid + 1; else hi = mid;
    }
    s_cnt = lo;
  }
  __syncthreads();
  const int64_t cnt    = s_cnt;
  const int64_t half   = cnt / 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += stride) {
    const int64_t j = cnt - 1 - i;
    const KeyT a = keys[i], b = keys[j];
    keys[i] = b;
    keys[j] = a;
    if (HAS_VAL) {
      const uint32_t va = vals[i], vb2 = vals[j];
      vals[i] = vb2;
      vals[j] = va;
    }
  }
}

// ==========================================================================================
// Hybrid MSD sort kernels
// ==========================================================================================
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u; }

struct MsdArgs {
  const void* in;
  void* out;
  const uint32_t* vin;  // payload (row index); NULL at level 0 = iota
  uint32_t* vout;
  SortPlan* plan;
  unsigned long long* status;  // [tiles][256] look-back granules, epoch = level + 1
  const uint32_t* base;        // level 0: [ranges][NB2MAX] output position of each bin of each input range
  uint32_t* cellcount;         // level 1: [256][NB2MAX] keys per cell, written by the last tile of each bucket
  uint32_t cellcap;            // level 1: every cell owns a slot of this many keys in the output buffer
  int64_t n;
  int level;
  int exp;  // experiment bits (A/B knob of the XCD placement; 0 in production)
  uint64_t desc_mask;
  unsigned long long spin_ticks = 0;  // as PassArgs
  long long inject_tile         = -1;
};

// One stable partition pass of the hybrid sort: k_radix_pass's tile body (wave64 ballot ranking,
// LDS reorder, decoupled look-back, coalesced write-out) over SEGMENTS.  A segment is a contiguous
// piece of the input with its own output bases and its own look-back chain: level 0 = the NRANGE
// input ranges (bases from the range-resolved histogram), level 1 = the 256 buckets of level 0
// (each (bucket, digit) cell has its own fixed-capacity slot; the position inside it is the look-back
// prefix).  Chains never cross segments, so tiles are handed out per XCD:
// list x (range x; buckets 32x..32x+31) has its own ticket counter, served first by the workgroups
// running on XCD x (HW_REG_XCC_ID) and by anyone once their own list is drained.  Neighbouring
// tiles thus share an L2, where the partial 128-B lines at the seams of their output runs merge
// (the per-XCD L2s are not coherent; see DESIGN.md "XCD-local write combining").  Correctness
// needs no placement assumption: within a list tickets are handed out in order, so every
// predecessor a tile can wait for is already owned by a running workgroup.
// (Measured and dropped in round 2: persistent workgroups that prefetch the next tile's keys into registers during
// the write-out.  A ticket taken late costs its latency before the write-out barrier (4.98 / 5.20 ms per pass against
// 4.21 / 4.70), a ticket taken early delays the aggregate its successors look back for (5.34 / 6.31 ms):
// profiles/r2_run8_bench_sort_persistent.jsonl, r2_run9_bench_persistent_early_ticket.jsonl.
// Also measured and dropped: issuing the first look-back window's status loads right after publishing the tile's own
// counts, so that they travel during the scan and the LDS reorder (the loads do stay in flight across the barriers:
// no s_waitcnt in between).  4.81 / 5.44 ms per pass against 4.2 / 4.7: words read that early are mostly not
// published yet, so the window is read twice, and the eight extra registers do not come for free
// (profiles/r2_run20_bench_sort_lookback_prefetch.jsonl).)
template <typename KeyT, int KIND, bool HAS_VAL, int KPT, int LBW, int NBL>
__global__ void __launch_bounds__(BT, (KPT <= 8 ? 8 : (KPT <= 12 ? 6 : 4))) k_msd_pass(MsdArgs a)
{
  constexpr int TILE = BT * KPT;
  constexpr int NB   = 1 << NBL;  // bins of this pass: 256 (level 0, level 1 up to 8 bits) or 512 (9-bit level 1)
  static_assert(NB <= BT, "one thread per bin");
  // Ranking inside the tile.  Integer keys: equal keys are indistinguishable, so the partition need not be
  // stable -- one returning LDS atomic per key on an NB-entry counter array replaces the ballot match.
  // Floats keep the stable wave-match ranking (-0.0 == +0.0 must keep input order); pairs too: the row index
  // breaks ties downstream only if cells keep input order.
  constexpr bool STABLE = KIND == K_FLOAT || HAS_VAL;
  constexpr int WROWS   = STABLE ? NW : 2;  // rows of s_whist: per-wave counters, or {counts, bin starts}
  extern __shared__ __attribute__((aligned(16))) char smem[];
  KeyT* s_keys       = reinterpret_cast<KeyT*>(smem);
  uint32_t* s_vals   = reinterpret_cast<uint32_t*>(smem + (size_t)TILE * sizeof(KeyT));  // [TILE] (HAS_VAL)
  // stable ranking: per-wave counters [NW][NB]; unstable: {counts, bin starts} [2][NB].  The pairs' 9-bit pass keeps its
  // counters in 16 bits (a tile holds 5120 rows): with 32-bit counters it needs 82 000 B of LDS, 160 B more than two
  // workgroups per CU can share (level 1 of sorted_order: 8.7 -> 7.4 ms per 1e9 rows); everywhere else 16-bit counters
  // measured slower (sub-dword LDS read-modify-write: level 0 of the pairs 6.15 -> 6.65 ms) and the pass fits twice anyway
  constexpr bool HIST16 = STABLE && HAS_VAL && NBL == 9;
  using HistT           = typename std::conditional<HIST16, uint16_t, uint32_t>::type;
  uint32_t* s_whist  = s_vals + (HAS_VAL ? TILE : 0);
  HistT* s_whist16   = reinterpret_cast<HistT*>(s_whist);
  uint32_t* s_gdelta = s_whist + (HIST16 ? WROWS * NB / 2 : WROWS * NB);                 // [NB]
  uint32_t* s_limit  = s_gdelta + NB;                                                    // [NB] end of each bin's output slot
  uint32_t* s_scan   = s_limit + NB;                                                     // [16]
  uint32_t* s_misc   = s_scan + 16;                                                      // [4]

  SortPlan* plan = a.plan;
  HybridPlan& hy = plan->hy;
  const int lvl  = a.level;
  if (!hy.attempt || plan->hf.state == 3) return;
  const KeyT* kin      = static_cast<const KeyT*>(a.in);
  KeyT* kout           = static_cast<KeyT*>(a.out);
  const uint32_t* vin  = a.vin;
  uint32_t* vout       = a.vout;
  const KeyT desc_mask = (KeyT)a.desc_mask;
  const uint32_t dmask = lvl == 0 ? 0xFFu : ((1u << hy.bits2) - 1u);
  const Digit0 dig     = lvl == 0 ? digit0_of(hy, (int)(8 * sizeof(KeyT))) : Digit0{hy.shift2, 0, dmask, 0u};
  const int nseg       = lvl == 0 ? NRANGE : BINS;
  const unsigned tid   = threadIdx.x;
  const unsigned lane  = lane_id();
  const unsigned w     = tid / GX_WAVE;
  const unsigned epoch = 9u + (unsigned)lvl;  // LSD passes use 1..8 on the same status array

  // ---- take a tile: own XCD's list first, then the others
  if (tid == 0) {
    const unsigned x = ((a.exp & 1) && lvl == 1) ? 0u : ((a.exp & 2) ? (blockIdx.x % NRANGE) : xcc_id());
    uint32_t g       = 0xFFFFFFFFu;
    for (int i = 0; i < NRANGE; ++i) {
      const unsigned y   = (x + i) % NRANGE;
      const uint32_t ntl = hy.list_tile0[lvl][y + 1] - hy.list_tile0[lvl][y];
      if (ntl == 0) continue;
      const uint32_t t = atomicAdd(&plan->cnt.ctr[lvl][y].v, 1u);
      if (t < ntl) {
        g = hy.list_tile0[lvl][y] + t;
        break;
      }
    }
    s_misc[0] = g;
  }
  __syncthreads();
  {
    // segment of global tile g: the one whose tile interval contains it (one table entry per thread)
    const uint32_t g = s_misc[0];
    if (g != 0xFFFFFFFFu && (int)tid < nseg) {
      const uint32_t lo = hy.seg_tile0[lvl][tid], hi = hy.seg_tile0[lvl][tid + 1];
      if (lo <= g && g < hi) {
        s_misc[1] = tid;
        s_misc[2] = g - lo;
      }
    }
  }
  __syncthreads();
  const uint32_t gtile = s_misc[0];
  if (gtile == 0xFFFFFFFFu) return;
  const uint32_t seg   = s_misc[1];
  const uint32_t jt    = s_misc[2];
  const int64_t base   = (int64_t)hy.seg_start[lvl][seg] + (int64_t)jt * TILE;
  const int64_t remain = (int64_t)hy.seg_count[lvl][seg] - (int64_t)jt * TILE;
  const int nvalid     = (int)(remain < (int64_t)TILE ? remain : (int64_t)TILE);
  // level 1: the bucket's cell slots (HybridPlan::ccap / cbase), requested here so that the loads fly under the key loads -- behind
  // the look-back, where they are used, they would sit on the tile's critical path (k_hf_scatter: 3.97 -> 3.68 ms for the same move)
  const uint32_t l1cap  = lvl == 1 ? cell_cap(hy, seg) : 0u;
  const uint32_t l1base = lvl == 1 ? cell_slot(hy, seg, 0u) : 0u;

  // ---- load (wave-striped)
  KeyT key[KPT];
  uint32_t val[HAS_VAL ? KPT : 1];
  const int wbase = (int)w * (KPT * GX_WAVE) + (int)lane;
  if (nvalid == TILE) {
#pragma unroll
    for (int j = 0; j < KPT; ++j) key[j] = kin[base + wbase + j * GX_WAVE];
  } else {
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int idx = wbase + j * GX_WAVE;
      key[j]        = (idx < nvalid) ? kin[base + idx] : KeyT(0);
    }
  }
  if (HAS_VAL) {
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int idx = wbase + j * GX_WAVE;
      val[j]        = (idx < nvalid) ? (vin ? vin[base + idx] : (uint32_t)(base + idx)) : 0u;
    }
  }
  HistT* my_hist = s_whist16 + w * NB;
  uint32_t packed[KPT];
  uint32_t tile_count = 0;
  if (STABLE) {
#pragma unroll
    for (int k = 0; k < NB / ((HIST16 ? 2 : 1) * GX_WAVE); ++k) reinterpret_cast<uint32_t*>(my_hist)[lane + k * GX_WAVE] = 0;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int idx = wbase + j * GX_WAVE;
      uint32_t d    = dig(to_sortable<KeyT, KIND>(key[j], desc_mask));
      if (idx >= nvalid) d = NB - 1;  // padding sorts last (it is also last in input order)
      uint32_t lower, cnt;
      match_rank<NBL>(d, true, ~0ull, lower, cnt);
      const uint32_t prev = my_hist[d];
      if (lower == 0) my_hist[d] = (HistT)(prev + cnt);
      packed[j] = (d << 16) | (prev + lower);
    }
    __syncthreads();
    if (tid < NB) {
      uint32_t sum = 0;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) {
        const uint32_t c         = s_whist16[w2 * NB + tid];
        s_whist16[w2 * NB + tid] = (HistT)sum;
        sum += c;
      }
      tile_count = sum;
    }
  } else {
    if (tid < NB) s_whist[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int idx    = wbase + j * GX_WAVE;
      const uint32_t d = dig(to_sortable<KeyT, KIND>(key[j], desc_mask));
      const uint32_t r = lds_rank(s_whist, d, idx < nvalid);
      packed[j]        = (d << 16) | r;
    }
    __syncthreads();
    if (tid < NB) tile_count = s_whist[tid];
  }
  uint32_t pub_count = tile_count;
  if (STABLE && tid == NB - 1) pub_count -= (uint32_t)(TILE - nvalid);
  if (tid < NB) {
    if ((long long)gtile != a.inject_tile) store_agent_u64(&a.status[(int64_t)gtile * NB + tid], pack_status(jt == 0 ? 2u : 1u, epoch, pub_count));
  }
  const uint32_t bin_start = block_exclusive_scan<BT>(tile_count, 0u, SumOp(), s_scan, (uint32_t*)nullptr);
  if (tid < NB) {
    if (STABLE) {
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) s_whist16[w2 * NB + tid] = (HistT)(s_whist16[w2 * NB + tid] + bin_start);
    } else {
      s_whist[NB + tid] = bin_start;  // row 1: bin starts (row 0 holds the counts)
    }
  }
  __syncthreads();

#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const uint32_t d = packed[j] >> 16;
    if (STABLE) {
      const uint32_t pos = (uint32_t)my_hist[d] + (packed[j] & 0xFFFFu);
      s_keys[pos]        = key[j];
      if (HAS_VAL) s_vals[pos] = val[j];
    } else if (wbase + j * GX_WAVE < nvalid) {
      s_keys[s_whist[NB + d] + (packed[j] & 0xFFFFu)] = key[j];
    }
  }

  if (tid < NB) {
    uint32_t prefix = 0;
    if (jt > 0) {
      int64_t p = (int64_t)gtile - 1;  // predecessors of the same segment have consecutive tile ids
      bool done = false;
      while (!done) {
        unsigned long long v[LBW];
#pragma unroll
        for (int k = 0; k < LBW; ++k) {
          const int64_t q = p - k;
          v[k]            = (q >= 0) ? load_agent_u64(&a.status[q * NB + tid]) : pack_status(2u, epoch, 0u);
        }
#pragma unroll
        for (int k = 0; k < LBW; ++k) {
          if (!done) {
            unsigned long long x = v[k];
            uint32_t spins       = 0;
            unsigned long long spin_t0 = 0;
            while ((x >> 62) == 0 || ((unsigned)(x >> 32) & 0xFFu) != epoch) {
              if (spin_guard(spins, spin_t0, a.plan, a.spin_ticks)) {  // as above
                x = pack_status(2u, epoch, 0u);
                break;
              }
              __builtin_amdgcn_s_sleep(2);
              x = load_agent_u64(&a.status[(p - k) * NB + tid]);
            }
            prefix += (uint32_t)x;
            if ((x >> 62) == 2u) done = true;
          }
        }
        p -= LBW;
      }
      if ((long long)gtile != a.inject_tile) store_agent_u64(&a.status[(int64_t)gtile * NB + tid], pack_status(2u, epoch, prefix + pub_count));
    }
    uint32_t gb, lim = 0xFFFFFFFFu;
    if (lvl == 0) {
      gb = a.base[seg * NB2MAX + tid];
    } else {
      // Level 1 needs no histogram: cell (bucket, bin) owns a fixed slot of `cellcap` keys, this tile's keys go
      // behind those of the bucket's earlier tiles (the look-back prefix), and the bucket's last tile leaves the
      // cell sizes behind for k_plan2.  A cell that outgrows its slot (skewed keys) is cut off and flagged: the
      // LSD passes then sort the column instead.
      const bool live = tid < (1u << hy.bits2);
      const uint32_t cellcap = l1cap;  // (per bucket; the device may also have chosen the larger cells: k_hy_plan stage 1)
      gb              = live ? l1base + tid * l1cap : 0u;
      lim             = live ? gb + cellcap : 0u;
      if (live) {
        if (prefix + pub_count > cellcap && !__hip_atomic_load(&hy.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
          atomicExch(&hy.overflow, 1);  // (only while the flag is down: thousands of overflowing (tile, bin) pairs would queue on one word)
        if (gtile + 1 == hy.seg_tile0[1][seg + 1]) a.cellcount[seg * NB2MAX + tid] = prefix + pub_count;
      }
    }
    s_limit[tid]  = lim;
    s_gdelta[tid] = gb + prefix - bin_start;
  }
  __syncthreads();

#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int i = j * BT + (int)tid;
    if (i < nvalid) {
      const KeyT k       = s_keys[i];
      const uint32_t d   = dig(to_sortable<KeyT, KIND>(k, desc_mask));
      const uint32_t dst = s_gdelta[d] + (uint32_t)i;
      if (dst < s_limit[d]) {
        kout[dst] = k;
        if (HAS_VAL) vout[dst] = s_vals[i];
      }
    }
  }
}

// After the level-1 pass, one wave per level-0 bucket: cell sizes -> output position of every cell (cells in key
// order: a bucket starts at its histogram offset), largest cell; the bucket that finishes last gives the verdict:
// the local sort runs iff no cell outgrew its slot and the cell sizes add up.
__global__ void __launch_bounds__(GX_WAVE) k_plan2(SortPlan* plan, uint32_t* __restrict__ cellcount,
                                                   uint32_t* __restrict__ cellstart, int npass, int cursor_path)
{
  HybridPlan& hy = plan->hy;
  if (!hy.attempt || (plan->hf.state == 3) != (cursor_path != 0)) return;
  constexpr int PER    = NB2MAX / GX_WAVE;  // 8 consecutive cells per lane
  const int b          = blockIdx.x;
  const unsigned lane  = lane_id();
  const uint32_t start = hy.gbin0[b];
  const uint32_t count = hy.hist0[b];
  const int nb2        = 1 << hy.bits2;
  // (splitter mode) an EQUALITY bucket: level 1 has copied its keys to the output and counted them on cell 0 -- the total is
  // checked like any bucket's, then the count is cleared so that no later kernel takes the bucket for one overfull cell
  const bool eqb = cursor_path && plan->sp.on && plan->sp.eq[b];
  // ... a NARROW bucket: level 1 has only counted its cells; the cell starts computed below are what k_sp_fill fills by, and the
  // counts are cleared behind them for the same reason
  const bool nrb = cursor_path && plan->sp.on && sp_narrow(plan->sp, (uint32_t)b, hy.bits2);
  uint32_t c[PER], sum = 0, mx = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int d2 = (int)lane * PER + k;
    c[k]         = d2 < nb2 ? cellcount[b * NB2MAX + d2] : 0u;
    sum += c[k];
    if (eqb) {
      if (d2 == 0) cellcount[b * NB2MAX] = 0u;
      c[k] = 0u;
    }
    if (nrb && d2 < nb2) cellcount[b * NB2MAX + d2] = 0u;
    if (!nrb) mx = c[k] > mx ? c[k] : mx;
  }
  const uint32_t inc = wave_inclusive_sum_dpp(sum);
  uint32_t run       = start + inc - sum;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int d2 = (int)lane * PER + k;
    if (d2 < nb2) cellstart[b * NB2MAX + d2] = run;
    run += c[k];
  }
  const uint32_t total = shfl(inc, GX_WAVE - 1);
  mx                   = wave_reduce(mx, MaxOp());
  if (mx > cell_cap(hy, (uint32_t)b) && lane == 0) atomicExch(&hy.overflow, 1);  // (a cell above its bucket's slot capacity)
  if (cursor_path && !nrb) {  // big cells of this bucket (cells that outgrew their slot: HybridPlan::big); a narrow bucket has none
    uint32_t bc = 0, bk = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      if (c[k] > cell_cap(hy, (uint32_t)b)) {
        ++bc;
        bk += c[k];
      }
    }
    bc = wave_reduce(bc, SumOp());
    bk = wave_reduce(bk, SumOp());
    if (lane == 0) {
      hy.bigcnt[b]  = bc;
      hy.bigkeys[b] = bk;
    }
  }
  if (lane == 0) {
    atomicMax(&hy.max_cell, mx);
    if (total != count) atomicExch(&hy.bad, 1);
    __threadfence();
    if (atomicAdd(&hy.plan2_done, 1u) == (uint32_t)BINS - 1u) {  // every bucket is in
      const int overflow     = atomicAdd(&hy.overflow, 0);
      // (cursor path: its level-1 cursors count every key, so the sizes add up with or without dropped keys)
      const int bad          = (cursor_path || !overflow) && atomicAdd(&hy.bad, 0);
      const uint32_t maxcell = atomicMax(&hy.max_cell, 0u);
      if (bad) atomicCAS(&plan->status, 0, 3);  // (never over a spin fault)
      const int big = (overflow || maxcell > (uint32_t)hy.cell_max) ? 1 : 0;
      // cursor path: big cells are handled on their own (k_big_plan decides); look-back path: they send the column to the LSD passes
      const int ok = (!bad && (cursor_path || !big)) ? 1 : 0;
      hy.ok        = ok;
      hy.big       = (ok && big) ? 1 : 0;
      if (ok && !big) {  // the LSD passes and the copy-only finalizer become no-ops
        for (int p = 0; p < npass; ++p) plan->pass_skip[p] = 1;
        plan->num_active = -1;
      }
    }
  }
}

// Sort one cell (<= 1 << CL2 keys sharing all bits above shift2) on its remaining bits inside LDS:
// nlocal stable 8-bit counting passes, keys resident in registers between the LDS exchanges, one
// HBM read and one HBM write of the cell in total.
// PAIRS (sorted_order: key + row index, index payload = iota at level 0): the cell's keys share all
// bits above shift2, and the stable partition passes left the cell in input order, so the word
// (low shift2 bits of the sortable key) << 14 | (position in the cell) is a distinct 64-bit key whose
// order is the stable order of the pairs.  Only these words go through LDS; the sorted positions
// then gather the original key and index of the cell (L2-resident: the cell was just read).
// (the position field of a packed word has CL2 bits: cells hold at most 1 << CL2 keys)

// PAIRS write-out: the sorted words hold the position of each row inside the cell in their low 14
// bits.  Integer keys are rebuilt from the word (cell prefix | low bits, the transform is an
// involution); float keys (-0.0 / NaN payloads are not recoverable) and the row indices are staged
// through the now free LDS buffer so that every HBM access stays coalesced.
template <typename KeyT, int KIND, bool HAS_VAL, int CL2>
__device__ __forceinline__ void pairs_write_out(KeyT* s_keys, const KeyT* __restrict__ in_cell, KeyT* __restrict__ out,
                                                const uint32_t* __restrict__ vin_cell, uint32_t* __restrict__ vout,
                                                int64_t start, uint32_t m, int shift2, KeyT desc_mask, bool write_keys = true,
                                                bool padded = false)
{
  // padded: the sorted words sit at s_keys[i + (i >> 4)] (k_local_place's layout), not at s_keys[i]
  // write_keys = false (sorted_order: only the permutation is asked for): the 8 B/row key write-back is skipped
  // in_cell / vin_cell point at the cell's slot; out / vout are indexed from `start`
  const KeyT* in      = in_cell - start;
  const uint32_t* vin = vin_cell ? vin_cell - start : nullptr;
  constexpr int LS_KPT = 16, LS_BT = (1 << CL2) / LS_KPT, LS_POS_BITS = CL2;
  const unsigned tid = threadIdx.x;
  const KeyT lowmask = (KeyT(1) << shift2) - KeyT(1);
  const KeyT hi      = to_sortable<KeyT, KIND>(in[start], desc_mask) & ~lowmask;  // shared by the whole cell
  uint32_t pos[LS_KPT];
#pragma unroll
  for (int j = 0; j < LS_KPT; ++j) {
    const int i = j * LS_BT + (int)tid;
    pos[j]      = 0;
    if ((uint32_t)i < m) {
      const KeyT wd = s_keys[padded ? i + (i >> 4) : i];
      pos[j]        = (uint32_t)wd & ((1u << LS_POS_BITS) - 1u);
      if (KIND != K_FLOAT && write_keys) out[start + i] = to_sortable<KeyT, KIND>(hi | ((wd >> LS_POS_BITS) & lowmask), desc_mask);
    }
  }
  __syncthreads();
  if (KIND == K_FLOAT && write_keys) {
#pragma unroll
    for (int j = 0; j < LS_KPT; ++j) {
      const int i = j * LS_BT + (int)tid;
      if ((uint32_t)i < m) s_keys[i] = in[start + i];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < LS_KPT; ++j) {
      const int i = j * LS_BT + (int)tid;
      if ((uint32_t)i < m) out[start + i] = s_keys[pos[j]];
    }
    __syncthreads();
  }
  if (!HAS_VAL) return;
  uint32_t* s_idx = reinterpret_cast<uint32_t*>(s_keys);
#pragma unroll
  for (int j = 0; j < LS_KPT; ++j) {
    const int i = j * LS_BT + (int)tid;
    if ((uint32_t)i < m) s_idx[i] = vin[start + i];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < LS_KPT; ++j) {
    const int i = j * LS_BT + (int)tid;
    if ((uint32_t)i < m) vout[start + i] = s_idx[pos[j]];
  }
}

// ---- k_local_place (round 3; integer keys and packed (key bits, position) words): ONE counting pass on the CL2 bits below
// the level-1 digit puts every key within a few positions of where it belongs -- as many bins as the cell has key slots
// (8192 or 16384), 0.93 keys per bin on uniform keys in full cells -- and two passes of per-THREAD networks over 16
// registers finish the job: thread t sorts positions [16t, 16t + 16) (60 comparators), then merges its upper half with
// the lower half of thread t + 1 (25 comparators), i.e. sorts the window shifted by 8.  A bin of <= 9 keys lies inside an
// aligned or inside a shifted window and bins are ordered among themselves, so the two passes together sort the cell.  A
// cell with a fuller bin (duplicate keys, a cluster; 6e-8 per bin on uniform keys) is left alone and its number appended
// to `todo`: k_local_sort, launched behind, sorts exactly those cells with its sub-bucket path.  The counters are BYTES
// (8 KiB for 8192 bins): a returning add of 1 << 8 * (bin & 3) on the bin's word, the old byte is the key's rank inside
// its bin; the add that sees 255 is about to carry into the neighbouring bin and raises the same flag.
// Against k_local_sort's sub-bucket path: ~85 comparators and one counting pass per 16 keys with 16 independent keys per
// lane in flight at every step, instead of 16 sub-buckets per wave taken two at a time through a chain of six dependent
// LDS round trips each (2.7 of its 5.5 ms, profiles/r3_run21_local_sort_ablation.txt): 5.6 -> 3.6 ms per 1e9 int64 keys,
// 8.9 -> 5.1 ms for sorted_order's pairs (profiles/r3_run22_local_place_ab.txt), 54 VGPRs.
// LDS: the keys with one key of padding per 16 (a thread's window is 17 keys from its neighbour's: conflict-free 8-byte
// accesses) + one byte per bin + one bin-group base per thread + scan words = 78.1 KiB for 8192-key cells (two workgroups
// per CU), 156.1 KiB for 16384-key cells (float keys, pairs above 1.02e9 rows: one workgroup per CU, as k_local_sort).
template <typename KeyT, int KIND, bool HAS_VAL, int CL2>
__device__ __forceinline__ bool place_applies(const HybridPlan& hy, int exp)
{
  // packed words: pairs always (k_hy_plan's pos_bits rule), float keys when the position fits next to the key bits
  constexpr bool PACKED = HAS_VAL || KIND == K_FLOAT;
  const int word_bits   = hy.shift2 + (PACKED ? CL2 : 0);
  return hy.nlocal > 0 && !(exp & (32 | 16 | 8 | 4)) && word_bits >= CL2 && word_bits <= 64;
}
// word of the cell sort: 8 bytes for 64-bit keys and for every packed (key bits, position) word; 32-bit integer keys travel as they are
template <typename KeyT, int KIND, bool HAS_VAL>
struct PlaceWord {
  typedef typename std::conditional<sizeof(KeyT) == 8 || HAS_VAL || KIND == K_FLOAT, uint64_t, uint32_t>::type type;
};
constexpr size_t place_lds_bytes(int cl2, int word_bytes = 8)
{
  return (size_t)((1 << cl2) + (1 << cl2) / 16) * word_bytes + (size_t)(1 << cl2) + (size_t)((1 << cl2) / 16) * 4 + 32 * 4;
}

// what a cell's workgroup must know before its first key load: size, slot capacity, slot, output position
struct CellMeta {
  uint32_t m, cap, slot, start;
};
__device__ __forceinline__ CellMeta cell_meta(const HybridPlan& hy, const uint32_t* __restrict__ hist2, const uint32_t* __restrict__ base2, uint32_t cell)
{
  const uint32_t b = cell >> hy.bits2, d2 = cell & ((1u << hy.bits2) - 1u);
  return CellMeta{hist2[b * NB2MAX + d2], cell_cap(hy, b), cell_slot(hy, b, d2), base2[b * NB2MAX + d2]};
}
// one cell of k_local_place (every thread of the workgroup calls it with the same cell; returns are block-uniform)
// (pre: the cell's CellMeta when the caller has it already -- splitter mode asks for both of its slots' at once)
template <typename KeyT, int KIND, bool HAS_VAL, int CL2, bool SPLIT = false>
__device__ __forceinline__ void local_place_cell(const uint32_t cell, const KeyT* in, KeyT* __restrict__ out, const uint32_t* vin,
                                                 uint32_t* __restrict__ vout, KeyT desc_mask, SortPlan* plan,
                                                 const uint32_t* __restrict__ hist2, const uint32_t* __restrict__ base2,
                                                 uint32_t* __restrict__ todo, int exp, const CellMeta* pre = nullptr)
{
  constexpr int LOCAL_MAX = 1 << CL2, LS_KPT = 16, LS_BT = LOCAL_MAX / LS_KPT, NPB = 1 << CL2;
  constexpr bool PACKED = HAS_VAL || KIND == K_FLOAT;
  typedef typename PlaceWord<KeyT, KIND, HAS_VAL>::type WordT;  // what goes through LDS and the networks
  static_assert(!PACKED || sizeof(KeyT) == 8, "packed words: 64-bit keys");
  HybridPlan& hy = plan->hy;
  const int bits2 = hy.bits2;
  const uint32_t b  = cell >> bits2;
  const uint32_t d2 = cell & ((1u << bits2) - 1u);
  // (the cell's size, output position and slot are requested together: behind the size check the slot lookup would be a second
  //  round trip before the first key load)
  const uint32_t m    = pre ? pre->m : hist2[b * NB2MAX + d2];
  const uint32_t cap  = pre ? pre->cap : cell_cap(hy, b);
  const uint32_t slot = pre ? pre->slot : cell_slot(hy, b, d2);
  const int64_t start = pre ? pre->start : base2[b * NB2MAX + d2];
  if (m == 0 || m > cap) return;  // (a big cell -- cursor path only -- is sorted through X, see HybridPlan::big)
  in += (int64_t)slot - start;  // the cell sits in its slot of the level-1 buffer
  if (HAS_VAL) vin += (int64_t)slot - start;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  WordT* s_keys    = reinterpret_cast<WordT*>(smem);                                                 // LOCAL_MAX + LOCAL_MAX / 16
  uint32_t* s_cnt8 = reinterpret_cast<uint32_t*>(smem + (size_t)(LOCAL_MAX + LOCAL_MAX / 16) * sizeof(WordT));  // NPB bytes
  uint32_t* s_base = s_cnt8 + NPB / 4;                                                               // [LS_BT] first position of bin 16 t
  uint32_t* s_scan = s_base + LS_BT;                                                                 // [32]
  const unsigned tid  = threadIdx.x;
  const unsigned lane = lane_id();
  const int wbase     = (int)(tid / GX_WAVE) * (LS_KPT * GX_WAVE) + (int)lane;
  const int shift2    = hy.shift2;
  const int dshift    = shift2 + (PACKED ? CL2 : 0) - CL2;
  // SPLIT (splitter mode, plain 64-bit keys): the counting digit is the CL2 bits of the key's position inside its bucket's range
  // that follow the cell's own (equal-width sub-slices of the cell; monotone, which is all the placement needs)
  const SpCell spc    = SPLIT ? sp_cell_of(plan->sp, b) : SpCell{0ull, 1ull, 0u, 63};
  const uint32_t spnc = SPLIT ? plan->sp.nc[b] : 1u;
  __shared__ uint2 s_wt[SPLIT ? SP_NPIECE : 1];  // the bucket's warp (read between the two barriers below; the next cell's write comes after both)
  auto bin_of         = [&](WordT wd) -> uint32_t {
    if constexpr (SPLIT) return sp_fine<CL2>(sp_warp(sp_frac(spc, (unsigned long long)wd), s_wt), spnc);
    else return (uint32_t)(wd >> dshift) & (uint32_t)(NPB - 1);
  };

  reinterpret_cast<uint4*>(s_cnt8)[tid] = make_uint4(0u, 0u, 0u, 0u);
  if constexpr (SPLIT) {
    if (tid < (unsigned)SP_NPIECE) s_wt[tid] = plan->sp.wt[b][tid];
  }
  WordT key[LS_KPT];
#pragma unroll
  for (int j = 0; j < LS_KPT; ++j) {
    const int idx = wbase + j * GX_WAVE;
    const KeyT k  = ((uint32_t)idx < m) ? in[start + idx] : KeyT(0);
    const KeyT sk = to_sortable<KeyT, KIND>(k, desc_mask);
    if constexpr (PACKED) key[j] = (WordT)(((sk & ((KeyT(1) << shift2) - KeyT(1))) << CL2) | (KeyT)idx);
    else key[j] = (WordT)sk;
  }
  __syncthreads();
  uint32_t rk[LS_KPT / 4] = {0u, 0u, 0u, 0u};  // rank inside the bin, one byte per key
  uint32_t bins2[SPLIT ? LS_KPT / 2 : 1];      // SPLIT: the bin of every key, two per register (the map costs a multiply: computed once)
  bool full = false;
#pragma unroll
  for (int j = 0; j < LS_KPT; ++j) {
    const int idx = wbase + j * GX_WAVE;
    if constexpr (SPLIT) {
      if ((j & 1) == 0) bins2[j >> 1] = 0;
    }
    if ((uint32_t)idx < m) {
      const uint32_t bin = bin_of(key[j]);
      if constexpr (SPLIT) bins2[j >> 1] |= bin << (16 * (j & 1));
      const uint32_t sh8 = (bin & 3u) * 8u;
      const uint32_t r   = (atomicAdd(&s_cnt8[bin >> 2], 1u << sh8) >> sh8) & 0xFFu;
      full |= r == 0xFFu;
      rk[j >> 2] |= r << (8 * (j & 3));
    }
  }
  __syncthreads();
  // bins 16 tid .. 16 tid + 15: byte-wise exclusive prefix inside the thread (x * 0x01010101 = inclusive prefix of the four
  // bytes of a word while the sums stay below 256: 16 bins x <= 9 keys), then a block scan of the thread totals
  const uint4 c4 = reinterpret_cast<uint4*>(s_cnt8)[tid];
  auto ge10 = [](uint32_t x) { return ((((x & 0x7F7F7F7Fu) + 0x76767676u) | x) & 0x80808080u) != 0u; };
  const bool crowded = ge10(c4.x) | ge10(c4.y) | ge10(c4.z) | ge10(c4.w);
  const uint32_t p0 = c4.x * 0x01010101u, p1 = c4.y * 0x01010101u, p2 = c4.z * 0x01010101u, p3 = c4.w * 0x01010101u;
  const uint32_t t0 = p0 >> 24, t1 = t0 + (p1 >> 24), t2 = t1 + (p2 >> 24), t3 = t2 + (p3 >> 24);
  uint4 e4;
  e4.x = p0 - c4.x;
  e4.y = p1 - c4.y + t0 * 0x01010101u;
  e4.z = p2 - c4.z + t1 * 0x01010101u;
  e4.w = p3 - c4.w + t2 * 0x01010101u;
  // block scan of the thread totals with ONE barrier: every wave publishes {its total, its crowded flag}, every thread adds
  // up the waves before its own (broadcast reads) -- the verdict on the cell rides on the same barrier
  constexpr int LS_NW = LS_BT / GX_WAVE;
  const uint32_t inc  = wave_inclusive_scan(t3, SumOp());
  const bool wbad     = ballot(full || crowded) != 0;
  if (lane == GX_WAVE - 1) s_scan[tid / GX_WAVE] = inc | (wbad ? 0x80000000u : 0u);  // totals stay below 2^15
  __syncthreads();
  uint32_t first = inc - t3, anybad = 0;
#pragma unroll
  for (int i = 0; i < LS_NW; ++i) {
    const uint32_t x = s_scan[i];
    anybad |= x;
    if ((unsigned)i < tid / GX_WAVE) first += x & 0x7FFFFFFFu;
  }
  if (anybad & 0x80000000u) {  // block-uniform
    if (tid == 0) todo[atomicAdd(&hy.todo_count, 1u)] = cell;
    return;
  }
  reinterpret_cast<uint4*>(s_cnt8)[tid] = e4;
  s_base[tid]                           = first;
  __syncthreads();
  const uint8_t* s_off8 = reinterpret_cast<const uint8_t*>(s_cnt8);
#pragma unroll
  for (int j = 0; j < LS_KPT; ++j) {
    const int idx = wbase + j * GX_WAVE;
    if ((uint32_t)idx < m) {
      uint32_t bin;
      if constexpr (SPLIT) bin = (bins2[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
      else bin = bin_of(key[j]);
      const uint32_t pos = s_base[bin >> 4] + (uint32_t)s_off8[bin] + ((rk[j >> 2] >> (8 * (j & 3))) & 0xFFu);
      s_keys[pos + (pos >> 4)] = key[j];
    }
  }
  __syncthreads();
  // a wave whose 1024 positions all lie behind the cell's last key has nothing to sort (part-filled cells: n well below
  // the size class's limit, 10-bit level 1); it only keeps the barriers
  const bool busy = 16u * (tid & ~(unsigned)(GX_WAVE - 1)) < m;
  WordT v[16];
  WordT* mine = s_keys + 17 * tid;  // positions 16 tid .. 16 tid + 15
  if (busy) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (16u * tid + (uint32_t)i < m) ? mine[i] : (WordT)~WordT(0);
    sort16_regs(v);
#pragma unroll
    for (int i = 0; i < 8; ++i) mine[i] = v[i];  // the lower half: final for thread 0, merged by thread tid - 1 otherwise
  }
  __syncthreads();
  if (busy) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i]     = v[8 + i];
      v[8 + i] = (16u * (tid + 1) + (uint32_t)i < m) ? mine[17 + i] : (WordT)~WordT(0);  // m <= 16 LS_BT: never past the last thread
    }
    merge16_regs(v);
#pragma unroll
    for (int i = 0; i < 8; ++i) mine[8 + i] = v[i];
    if (16u * (tid + 1) < m) {
#pragma unroll
      for (int i = 0; i < 8; ++i) mine[17 + i] = v[8 + i];
    }
  }
  __syncthreads();
  if constexpr (PACKED) {
    pairs_write_out<KeyT, KIND, HAS_VAL, CL2>(s_keys, in + start, out, HAS_VAL ? vin + start : nullptr, vout, start, m, shift2, desc_mask, !(exp & 64),
                                              true);
    return;
  } else {
#pragma unroll
    for (int j = 0; j < LS_KPT; ++j) {
      const int i = j * LS_BT + (int)tid;
      if ((uint32_t)i < m) out[start + i] = from_sortable<KeyT, KIND == K_FLOAT ? K_UNSIGNED : KIND>((KeyT)s_keys[i + (i >> 4)], desc_mask);  // (K_FLOAT never gets here: PACKED)
    }
  }
}

// The workgroups walk the cells with a stride (gridDim.x = the number of cells: one cell each, the default; a smaller grid is
// the persistent form, A/B knob gx_sort_set_place_grid).
template <typename KeyT, int KIND, bool HAS_VAL, int CL2>
__global__ void __launch_bounds__((1 << CL2) / 16, 4) k_local_place(const KeyT* in, KeyT* __restrict__ out, const uint32_t* vin,
                                                                    uint32_t* __restrict__ vout, KeyT desc_mask, SortPlan* plan,
                                                                    const uint32_t* __restrict__ hist2, const uint32_t* __restrict__ base2,
                                                                    uint32_t* __restrict__ todo, int exp, int cursor_path)
{
  HybridPlan& hy = plan->hy;
  if (!hy.attempt || !hy.ok || (plan->hf.state == 3) != (cursor_path != 0)) return;
  if (hy.cell_max != (1 << CL2)) return;  // (both cell sizes are launched when the device may choose: see k_hy_plan's cell_alt)
  if (!place_applies<KeyT, KIND, HAS_VAL, CL2>(hy, exp)) return;  // k_local_sort, launched behind, takes every cell
  const uint32_t ncells = (uint32_t)BINS << hy.bits2;
  constexpr bool CAN_SPLIT = sizeof(KeyT) == 8 && !HAS_VAL && KIND != K_FLOAT && CL2 == 13;
  if (CAN_SPLIT && cursor_path && plan->sp.on) {  // (block-uniform; an instantiation of its own: the bit-digit path is untouched)
    if constexpr (CAN_SPLIT) {
      // a bucket uses the FIRST nc of its 1 << bits2 cell slots (about half of them: the grid is sized for half the slots), so the two
      // slots of a workgroup must not be the same cell index of two buckets -- both used or both empty: until run 14 half the
      // workgroups sorted two cells back to back and the other half none.  Every other round walks its group of slots backwards
      // (slot ^ (slots per bucket - 1): a permutation inside the bucket, so every slot is still visited exactly once).
      // The sizes / slots / positions of BOTH slots are requested before either is looked at: one of the two is empty as a rule, and
      // finding that out cost the workgroup a round trip to memory of its own (run 16: 43 % more wave-cycles waiting than on bit
      // digits, 1400 per wave -- the 0.9 ms this stage took longer in splitter mode).
      // (odd rounds visit cell ^ flip -- a permutation of the round's cells only when the grid is a whole number of buckets: the default
      //  grid is; a grid from the A/B knob gx_sort_set_place_grid that is not walks linearly -- ADVICE r5: some cells twice, others never)
      const uint32_t flip = (gridDim.x % (1u << hy.bits2)) == 0u ? (1u << hy.bits2) - 1u : 0u;
      const uint32_t c0 = blockIdx.x, c1 = blockIdx.x + gridDim.x;  // (c0 < ncells: the grid is never larger than the slots)
      const bool two    = c1 < ncells;
      const uint32_t s1 = two ? (c1 ^ flip) : c0;
      const CellMeta m0 = cell_meta(hy, hist2, base2, c0);
      const CellMeta m1 = cell_meta(hy, hist2, base2, s1);
      if (m0.m != 0u) {  // (block-uniform)
        local_place_cell<KeyT, KIND, HAS_VAL, CL2, true>(c0, in, out, vin, vout, desc_mask, plan, hist2, base2, todo, exp, &m0);
        __syncthreads();
      }
      if (two && m1.m != 0u) {
        local_place_cell<KeyT, KIND, HAS_VAL, CL2, true>(s1, in, out, vin, vout, desc_mask, plan, hist2, base2, todo, exp, &m1);
        __syncthreads();
      }
      uint32_t round = 2;  // (a smaller grid -- the A/B knob gx_sort_set_place_grid -- walks on)
      for (uint32_t cell = blockIdx.x + 2u * gridDim.x; cell < ncells && cell >= 2u * gridDim.x; cell += gridDim.x, ++round) {
        local_place_cell<KeyT, KIND, HAS_VAL, CL2, true>((round & 1u) ? (cell ^ flip) : cell, in, out, vin, vout, desc_mask, plan, hist2, base2, todo, exp);
        __syncthreads();
      }
    }
    return;
  }
  for (uint32_t cell = blockIdx.x; cell < ncells; cell += gridDim.x) {
    local_place_cell<KeyT, KIND, HAS_VAL, CL2>(cell, in, out, vin, vout, desc_mask, plan, hist2, base2, todo, exp);
    __syncthreads();  // the next cell reuses the LDS
  }
}

// One cell of k_local_sort (every thread of the workgroup calls it with the same cell; returns are block-uniform).
// IoT: the column's element type when it is narrower than the word the cell is sorted on (32-bit integer keys: KeyT =
// uint64_t words holding the SORTABLE form of the key, so that every transform below is the identity until the store).
template <typename KeyT, int KIND, bool HAS_VAL, int CL2, typename IoT = KeyT>
__device__ __forceinline__ void local_sort_cell(const uint32_t cell, const IoT* in, IoT* __restrict__ out,
                                                const uint32_t* vin, uint32_t* __restrict__ vout,
                                                KeyT desc_mask_in, SortPlan* plan,
                                                const uint32_t* __restrict__ hist2, const uint32_t* __restrict__ base2, int exp)
{
  constexpr bool NARROW = !std::is_same<IoT, KeyT>::value;
  static_assert(!NARROW || (!HAS_VAL && KIND != K_FLOAT && sizeof(KeyT) == 8), "narrow columns: integer keys only");
  // PAIRS = the packed-word mode: pairs, and float keys (whose -0.0 == +0.0 ties must keep input order
  // and whose original bits cannot be rebuilt from the sortable form)
  // (floats whose remaining bits do not leave room for the position keep plain keys and take the
  // stable passes; k_plan never attempts the hybrid path for pairs in that case)
  constexpr int LOCAL_MAX = 1 << CL2, LS_KPT = 16, LS_BT = LOCAL_MAX / LS_KPT, LS_NW = LS_BT / GX_WAVE, LS_POS_BITS = CL2;
  constexpr int SB = CL2 - 6, NSB = 1 << SB;  // LDS split into NSB sub-buckets of ~64 keys
  constexpr bool CAN_PACK = HAS_VAL || KIND == K_FLOAT;
  HybridPlan& hy = plan->hy;
  const bool PAIRS     = CAN_PACK && (hy.shift2 + LS_POS_BITS <= 64);  // k_hy_plan never attempts pairs otherwise
  const KeyT desc_mask = desc_mask_in;
  const int pos_shift  = PAIRS ? LS_POS_BITS : 0;
  // sortable form of a register word: packed words already are, plain keys go through the transform
  auto sortable = [&](KeyT x) -> KeyT { return (PAIRS || NARROW) ? x : to_sortable<KeyT, KIND>(x, desc_mask); };
  // stores: `unsort` takes the sortable form of a key, `put` the form the registers hold (raw keys; sortable words when NARROW)
  constexpr int IK = KIND == K_FLOAT ? K_UNSIGNED : KIND;  // (K_FLOAT keys never leave through unsort: they travel as packed words or raw)
  auto unsort = [&](KeyT x) -> IoT {
    if constexpr (NARROW) return from_sortable<IoT, IK>((IoT)x, (IoT)desc_mask);
    else return from_sortable<KeyT, IK>(x, desc_mask);
  };
  auto put = [&](KeyT x) -> IoT {
    if constexpr (NARROW) return from_sortable<IoT, IK>((IoT)x, (IoT)desc_mask);
    else return x;
  };
  extern __shared__ __attribute__((aligned(16))) char smem[];
  KeyT* s_keys      = reinterpret_cast<KeyT*>(smem);                                           // LOCAL_MAX
  uint32_t* s_whist = reinterpret_cast<uint32_t*>(smem + (size_t)LOCAL_MAX * sizeof(KeyT));   // [LS_NW][256]
  uint32_t* s_scan  = s_whist + LS_NW * BINS;                                                 // [32]
  const int bits2   = hy.bits2;
  const uint32_t b  = cell >> bits2;
  const uint32_t d2 = cell & ((1u << bits2) - 1u);
  const uint32_t m    = hist2[b * NB2MAX + d2];  // cell size (level-1 pass), output position (k_plan2)
  const uint32_t cap  = cell_cap(hy, b);
  const uint32_t slot = cell_slot(hy, b, d2);
  const int64_t start = base2[b * NB2MAX + d2];
  if (m == 0 || m > cap) return;  // (big cells: see HybridPlan::big)
  in += (int64_t)slot - start;  // the cell sits in its slot of the level-1 buffer
  if (HAS_VAL) vin += (int64_t)slot - start;
  const unsigned tid  = threadIdx.x;
  const unsigned lane = lane_id();
  const unsigned w    = tid / GX_WAVE;
  const int wbase     = (int)w * (LS_KPT * GX_WAVE) + (int)lane;
  const int nlocal    = hy.nlocal;
  uint32_t* my_hist   = s_whist + w * BINS;  // this wave's 256 counters
  const bool cursor_path_split = plan->hf.state == 3 && plan->sp.on;

  KeyT key[LS_KPT];
#pragma unroll
  for (int j = 0; j < LS_KPT; ++j) {
    const int idx = wbase + j * GX_WAVE;
    if constexpr (NARROW) key[j] = ((uint32_t)idx < m) ? (KeyT)to_sortable<IoT, KIND>(in[start + idx], (IoT)desc_mask) : KeyT(0);
    else key[j] = ((uint32_t)idx < m) ? in[start + idx] : KeyT(0);
    if (PAIRS) {
      const KeyT sk = to_sortable<KeyT, KIND>(key[j], desc_mask_in);
      key[j]        = ((sk & ((KeyT(1) << hy.shift2) - KeyT(1))) << LS_POS_BITS) | (KeyT)idx;
    }
  }
  // ---- fast path (integer keys): ONE unstable 8-bit MSD split in LDS on the byte below the
  // level-1 digit (LDS atomics, no ballots), then every <=128-key sub-bucket is sorted on the full
  // key by one wave's in-register bitonic network.  Equal integer keys are indistinguishable, so
  // stability is not needed; floats (-0.0 == +0.0 must keep input order) and cells with a
  // sub-bucket above 128 keys take the stable LSD passes below.
  // (splitter mode: the cells share no bit prefix, so the bit-sliced sub-bucket split does not apply -- the stable passes below sort
  //  a crowded cell on every varying byte)
  if ((PAIRS || KIND != K_FLOAT) && nlocal > 0 && !(cursor_path_split)) {
    uint32_t* s_cnt   = s_scan + 32;          // [NSB] sub-bucket counts (own area: the wave rows of s_whist serve wave_split_sort)
    uint32_t* s_start = s_scan + 32 + BINS;   // [NSB] exclusive starts
    const int sshift  = hy.shift2 - SB + pos_shift;  // shift2 >= 8 (k_hy_plan)
    if (tid < NSB) s_cnt[tid] = 0;
    __syncthreads();
    uint32_t rank[LS_KPT];
#pragma unroll
    for (int j = 0; j < LS_KPT; ++j) {
      const int idx = wbase + j * GX_WAVE;
      key[j]        = sortable(key[j]);  // an involution for integer kinds
      rank[j]       = (exp & 8) ? (uint32_t)(idx >> 8)  // ablation 8: no atomics
                                : lds_rank(s_cnt, (uint32_t)(key[j] >> sshift) & (uint32_t)(NSB - 1), (uint32_t)idx < m);
    }
    __syncthreads();
    const uint32_t c   = tid < NSB ? s_cnt[tid] : 0u;
    const int too_big  = __syncthreads_or(c > 128u);
    if (!too_big) {
      const uint32_t st = block_exclusive_scan<LS_BT>(c, 0u, SumOp(), s_scan, (uint32_t*)nullptr);
      if (tid < NSB) s_start[tid] = st;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < LS_KPT; ++j) {
        const int idx = wbase + j * GX_WAVE;
        if ((uint32_t)idx < m) s_keys[s_start[(uint32_t)(key[j] >> sshift) & (uint32_t)(NSB - 1)] + rank[j]] = key[j];
      }
      __syncthreads();
      // sorted sub-bucket in registers (lane l: elements 2l, 2l + 1) -> LDS (packed words: the gathering write-out
      // needs them) or straight to HBM (plain integer keys: 1 KiB per wave and sub-bucket)
      auto emit2 = [&](KeyT* sub, uint32_t cnt, uint32_t o, uint64_t k0, uint64_t k1) {
        const uint32_t e0 = 2 * lane, e1 = 2 * lane + 1;
        if (PAIRS) {
          if (e0 < cnt) sub[e0] = (KeyT)k0;
          if (e1 < cnt) sub[e1] = (KeyT)k1;
        } else {
          if (e0 < cnt) out[start + o + e0] = unsort((KeyT)k0);
          if (e1 < cnt) out[start + o + e1] = unsort((KeyT)k1);
        }
      };
      // crowded bin (or too few bits left for a counting split): the in-register bitonic network
      auto network = [&](KeyT* sub, uint32_t cnt, uint32_t o) {
        uint64_t k0, k1;
        if (cnt <= 64) {
          k0 = lane < cnt ? (uint64_t)sub[lane] : ~0ull;
          wave_bitonic64(k0);
          if (lane < cnt) {
            if (PAIRS) sub[lane] = (KeyT)k0; else out[start + o + lane] = unsort((KeyT)k0);
          }
        } else {
          k0 = (uint64_t)sub[lane];
          k1 = lane + 64 < cnt ? (uint64_t)sub[64 + lane] : ~0ull;
          wave_bitonic128(k0, k1);
          if (PAIRS) {
            sub[lane] = (KeyT)k0;
            if (lane + 64 < cnt) sub[64 + lane] = (KeyT)k1;
          } else {
            out[start + o + lane] = unsort((KeyT)k0);
            if (lane + 64 < cnt) out[start + o + 64 + lane] = unsort((KeyT)k1);
          }
        }
      };
      // every wave takes its sub-buckets two at a time: counting split on the next byte + odd-even clean-up for
      // both, their LDS round trips interleaved (wave_split_sort_x2)
      for (int sb = (int)w; sb < NSB; sb += 2 * LS_NW) {
        if (exp & 4) break;  // ablation: no sorting step
        const int sbB       = sb + LS_NW;
        const uint32_t cntA = s_cnt[sb], oA = s_start[sb];
        const uint32_t cntB = sbB < NSB ? s_cnt[sbB] : 0u, oB = sbB < NSB ? s_start[sbB] : 0u;
        const bool doA = cntA > 1, doB = cntB > 1;
        if (!doA && !doB) continue;
        KeyT* subA = s_keys + oA;
        KeyT* subB = s_keys + oB;
        unsigned okm = 0;
        uint64_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
        if (sshift >= 8 && !(exp & 16))
          okm = wave_split_sort_x2(reinterpret_cast<uint64_t*>(subA), doA ? cntA : 0u, reinterpret_cast<uint64_t*>(subB),
                                   doB ? cntB : 0u, my_hist, sshift - 8, a0, a1, b0, b1);
        if (okm & 1u) emit2(subA, cntA, oA, a0, a1);
        else if (doA) network(subA, cntA, oA);
        if (okm & 2u) emit2(subB, cntB, oB, b0, b1);
        else if (doB) network(subB, cntB, oB);
      }
      if (!PAIRS) {
        // sub-buckets of 0 or 1 keys were skipped by the loop above: they leave here
        if (!(exp & 4)) {
          if (tid < NSB && s_cnt[tid] == 1) out[start + s_start[tid]] = unsort(s_keys[s_start[tid]]);
          return;
        }
      }
      __syncthreads();
      if constexpr (!NARROW) {
        if (PAIRS) {
          pairs_write_out<KeyT, KIND, HAS_VAL, CL2>(s_keys, in + start, out, HAS_VAL ? vin + start : nullptr, vout, start, m, hy.shift2, desc_mask_in, !(exp & 64));
          return;
        }
      }
#pragma unroll
      for (int j = 0; j < LS_KPT; ++j) {
        const int i = j * LS_BT + (int)tid;
        if ((uint32_t)i < m) out[start + i] = unsort(s_keys[i]);  // integer kinds only
      }
      return;
    }
    // undo the transform and fall through to the stable passes
#pragma unroll
    for (int j = 0; j < LS_KPT; ++j) {
      if constexpr (NARROW) key[j] = key[j];  // (words that ARE the sortable form)
      else key[j] = PAIRS ? key[j] : from_sortable<KeyT, IK>(key[j], desc_mask);
    }
    __syncthreads();
  }
  for (int lp = 0; lp < nlocal; ++lp) {
    const int shift      = hy.lshift[lp] + pos_shift;
    const uint32_t dmask = (1u << hy.lbits[lp]) - 1u;
#pragma unroll
    for (int k = 0; k < BINS / GX_WAVE; ++k) my_hist[lane + k * GX_WAVE] = 0;
    uint32_t packed[LS_KPT];
#pragma unroll
    for (int j = 0; j < LS_KPT; ++j) {
      const int idx     = wbase + j * GX_WAVE;
      const bool live   = (uint32_t)idx < m;
      const uint64_t act = ballot(live);
      packed[j]         = 0;
      if (act == 0) continue;  // wave-uniform
      const uint32_t d = (uint32_t)(sortable(key[j]) >> shift) & dmask;
      uint32_t lower, cnt;
      match_rank8(d, live, act, lower, cnt);
      if (live) {
        const uint32_t prev = my_hist[d];
        if (lower == 0) my_hist[d] = prev + cnt;
        packed[j] = (d << 16) | (prev + lower);
      }
    }
    __syncthreads();
    uint32_t tile_count = 0;
    if (tid < BINS) {
      uint32_t sum = 0;
#pragma unroll
      for (int w2 = 0; w2 < LS_NW; ++w2) {
        const uint32_t c         = s_whist[w2 * BINS + tid];
        s_whist[w2 * BINS + tid] = sum;
        sum += c;
      }
      tile_count = sum;
    }
    const uint32_t bin_start = block_exclusive_scan<LS_BT>(tile_count, 0u, SumOp(), s_scan, (uint32_t*)nullptr);
    if (tid < BINS) {
#pragma unroll
      for (int w2 = 0; w2 < LS_NW; ++w2) s_whist[w2 * BINS + tid] += bin_start;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < LS_KPT; ++j) {
      const int idx = wbase + j * GX_WAVE;
      if ((uint32_t)idx < m) s_keys[my_hist[packed[j] >> 16] + (packed[j] & 0xFFFFu)] = key[j];
    }
    __syncthreads();
    if (lp + 1 < nlocal) {
#pragma unroll
      for (int j = 0; j < LS_KPT; ++j) {
        const int idx = wbase + j * GX_WAVE;
        key[j]        = ((uint32_t)idx < m) ? s_keys[idx] : KeyT(0);
      }
      __syncthreads();  // the next pass rewrites s_whist / s_keys
    }
  }
  if (nlocal == 0) {  // nothing left to sort on: the cell is a copy
#pragma unroll
    for (int j = 0; j < LS_KPT; ++j) {
      const int idx = wbase + j * GX_WAVE;
      if ((uint32_t)idx < m) {
        if (PAIRS) {
          out[start + idx] = in[start + idx];
          if (HAS_VAL) vout[start + idx] = vin[start + idx];
        } else {
          out[start + idx] = put(key[j]);
        }
      }
    }
    return;
  }
  if constexpr (!NARROW) {
    if (PAIRS) {
      pairs_write_out<KeyT, KIND, HAS_VAL, CL2>(s_keys, in + start, out, HAS_VAL ? vin + start : nullptr, vout, start, m, hy.shift2, desc_mask_in, !(exp & 64));
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < LS_KPT; ++j) {
    const int i = j * LS_BT + (int)tid;
    if ((uint32_t)i < m) out[start + i] = put(s_keys[i]);
  }
}


// The cells of the hybrid path that k_local_place did not sort: all of them when the placement does not apply (fewer key
// bits left than the counting pass takes, a packed word that does not fit, knob 32), otherwise the crowded cells whose
// numbers it left in `todo`.  The grid is a fixed number of workgroups that walk the list with a stride: a launch that
// finds nothing to do -- the common case behind k_local_place, and the look-back path's instance behind a cursor-path
// sort -- costs a few microseconds instead of the ~0.1 ms that dispatching one workgroup per cell takes.
template <typename KeyT, int KIND, bool HAS_VAL, int CL2, typename IoT = KeyT>
__global__ void __launch_bounds__((1 << CL2) / 16, 4) k_local_sort(const IoT* in, IoT* __restrict__ out,
                                                      const uint32_t* vin, uint32_t* __restrict__ vout,
                                                      KeyT desc_mask_in, SortPlan* plan,
                                                      const uint32_t* __restrict__ hist2, const uint32_t* __restrict__ base2,
                                                      int exp = 0, int cursor_path = 0, const uint32_t* __restrict__ todo = nullptr)
{
  HybridPlan& hy = plan->hy;
  if (!hy.attempt || !hy.ok || (plan->hf.state == 3) != (cursor_path != 0)) return;
  if (hy.cell_max != (1 << CL2)) return;
  const bool listed    = todo != nullptr && place_applies<IoT, KIND, HAS_VAL, CL2>(hy, exp);  // (as k_local_place decided: on the column's type)
  const uint32_t count = listed ? hy.todo_count : ((uint32_t)BINS << hy.bits2);
  for (uint32_t e = blockIdx.x; e < count; e += gridDim.x) {
    local_sort_cell<KeyT, KIND, HAS_VAL, CL2, IoT>(listed ? todo[e] : e, in, out, vin, vout, desc_mask_in, plan, hist2, base2, exp);
    __syncthreads();  // the next cell reuses the LDS
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Round 3: the CURSOR path of the hybrid sort -- integer keys, keys only (cudf::sort of one int64 / uint64 column, the
// BASELINE config-2 workload).  Equal integer keys are indistinguishable, so neither partition level has to be stable
// or deterministic in its order: a tile reserves its run in a bin's slot with ONE returning atomic on the bin's cursor
// instead of waiting for its predecessor's prefix (k_msd_pass's decoupled look-back), tiles come from blockIdx (no
// ticket), and nothing upstream needs the exact histogram any more:
//   k_hf_sample<false>  OR / OR-of-complements over a 1/stride SAMPLE (every stride-th 64-key chunk)  -> digit positions
//   k_hf_sample<true>   per-range histogram of the level-0 digit over the same sample                 -> slot capacities
//   k_hf_scatter<0>     level 0 into padded (range, bin) slots of estimate + 8 sigma + seam allowance; it reads every
//                       key anyway, so it also reduces the EXACT varying-bit masks and counts every slot
//   k_hf_plan2          the verdict: the sample's top varying bit is the column's, no slot overflowed, the counts add up
//                       -> bucket sizes, the local sort's digits (from the exact masks), the region table of level 1;
//                       otherwise the plan is reset and the look-back path (k_hy_hist ... k_local_sort, enqueued behind
//                       and skipped on success) sorts the column from scratch: no host round trip on either branch
//   k_hf_scatter<1>     level 1: the (bucket, range) regions -> padded cell slots, cursor = the cell's size
//   k_plan2, k_local_sort as on the look-back path.
// HBM traffic 16 + 16 + 16 = 48 B/row (+ 2 x 8/stride for the sample) against 56 with the histogram read.  A 10-bit
// level 1 (two bins per thread) keeps 8192-key cells up to 2^31 rows: the slow window of round 2, n in (1.02e9, 2.1e9],
// where the 9-bit pass forced 16384-key cells at half occupancy, is gone for these keys.
// ------------------------------------------------------------------------------------------------------------------
constexpr int HF_CHUNK = GX_WAVE;  // keys per sample chunk: one 512-byte wave load

template <typename KeyT, int KIND, bool HIST>
__global__ void __launch_bounds__(256) k_hf_sample(const KeyT* __restrict__ in, int64_t n, KeyT desc_mask, SortPlan* plan, int stride,
                                                   int64_t range_rows)
{
  HybridPlan& hy = plan->hy;
  FastPlan& hf   = plan->hf;
  // HIST = false: the masks AND a speculative histogram of the TOP byte (the level-0 digit of any column whose top bit
  // varies: full-range integers); HIST = true: the histogram of the digit k_hf_plan chose, only when that is another one
  if (HIST && (hf.state != 1 || hf.hist_ready)) return;
  __shared__ uint32_t s_hist[NRANGE * BINS];
  __shared__ unsigned long long s_red[2 * 4];
  const unsigned tid = threadIdx.x, lane = lane_id();
  for (int i = tid; i < NRANGE * BINS; i += 256) s_hist[i] = 0;
  __syncthreads();
  const Digit0 dig      = HIST ? digit0_of(hy, (int)(8 * sizeof(KeyT))) : Digit0{(int)(8 * sizeof(KeyT) - 8), 0, 0xFFu, 0u};
  const int64_t step    = (int64_t)stride * HF_CHUNK;
  const int64_t nchunks = div_up(n, step);
  const int64_t nw      = (int64_t)gridDim.x * 4;
  constexpr int U       = 8;  // chunks in flight per wave
  KeyT vor = 0, vnor = 0, vfold = 0;
  for (int64_t c0 = (int64_t)blockIdx.x * 4 + tid / GX_WAVE; c0 < nchunks; c0 += nw * U) {
    KeyT raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = (c0 + u * nw) * step + lane;
      raw[u]            = (c0 + u * nw < nchunks && row < n) ? in[row] : KeyT(0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = (c0 + u * nw) * step + lane;
      const bool live   = c0 + u * nw < nchunks && row < n;
      const KeyT k      = to_sortable<KeyT, KIND>(raw[u], desc_mask);
      if (!HIST && live) {
        if constexpr (KIND == K_FTOTAL) {
          if (float_unclean<KeyT>(raw[u])) hf.fail = 1;  // a NaN / -0.0 in the sample: stage 0 sends the column to the look-back path at once
        }
        vor |= k;
        vnor |= (KeyT)~k;
        if (KIND == K_SIGNED) vfold |= (KeyT)(raw[u] ^ (KeyT)(KeyT(0) - (KeyT)(raw[u] >> (8 * sizeof(KeyT) - 1))));
      }
      // a chunk never straddles two ranges (ranges are whole tiles, chunks start at multiples of 64)
      const int64_t r64 = range_rows > 0 ? row / range_rows : (int64_t)(NRANGE - 1);
      const int r       = r64 < NRANGE - 1 ? (int)r64 : NRANGE - 1;
      (void)lds_rank(s_hist + r * BINS, dig(k), live);
    }
  }
  if (!HIST) {
    const unsigned long long wo = wave_reduce((unsigned long long)vor, [](unsigned long long x, unsigned long long y) { return x | y; });
    const unsigned long long wn = wave_reduce((unsigned long long)vnor, [](unsigned long long x, unsigned long long y) { return x | y; });
    if (lane == 0) {
      s_red[tid / GX_WAVE]     = wo;
      s_red[4 + tid / GX_WAVE] = wn;
    }
    __syncthreads();
    if (tid == 0) {
      atomicOr(&hy.or_mask, s_red[0] | s_red[1] | s_red[2] | s_red[3]);
      atomicOr(&hy.nor_mask, s_red[4] | s_red[5] | s_red[6] | s_red[7]);
    }
    if (KIND == K_SIGNED) {
      const unsigned long long wf = wave_reduce((unsigned long long)vfold, [](unsigned long long x, unsigned long long y) { return x | y; });
      if (lane == 0 && wf) atomicOr(&hy.fold_x, wf);
    }
  }
  __syncthreads();
  for (int i = tid; i < NRANGE * BINS; i += 256) {
    const uint32_t c = s_hist[i];
    if (c) atomicAdd(&hf.samp[i / BINS][i % BINS], c);
  }
}

// rows of input range r: ranges are whole tiles, the last one takes the remainder
__device__ __forceinline__ int64_t hf_range_rows(int r, int64_t n, int64_t range_rows)
{
  const int64_t b = (int64_t)r * range_rows < n ? (int64_t)r * range_rows : n;
  const int64_t e = (r == NRANGE - 1) ? n : (b + range_rows < n ? b + range_rows : n);
  return e - b;
}

__device__ __forceinline__ void hf_give_up(SortPlan* plan, int state)
{
  // the look-back path starts from a clean plan
  plan->hy.attempt  = 0;
  plan->hy.or_mask  = 0;
  plan->hy.nor_mask = 0;
  plan->hy.overflow = 0;
  plan->hy.fold     = 0;
  plan->hy.fold_x   = 0;
  plan->hf.state    = state;
}

// One block of 256 threads, three stages:
//   0 (after the mask sample)  digit positions, as k_hy_plan's stage 0
//   1 (after the histogram sample)  slot capacities and positions of level 0
//   2 (after level 0)  the verdict + everything level 1, k_plan2 and the local sort need
__global__ void __launch_bounds__(BINS) k_hf_plan(SortPlan* plan, int stage, int key_bits, int64_t n, int bits2, int cell_max, int stride,
                                                  int64_t range_rows, int tile_rows, unsigned long long slot_rows, float margin, int min_shift2,
                                                  int bits2_max, unsigned long long cell_budget = 0, int signed_keys = 0, int allow_counting = 0,
                                                  int allow_split = 0)
{
  // signed_keys: the sign fold of the level-0 digit may be planned (HybridPlan::fold; never for the sharded sort, whose digit
  // positions come from the masks of all ranks)
  // bits2_max: level-1 bits the launches behind are sized for (bits2 or bits2 + 1): stage 2 takes the extra bit when the EXACT
  // level-0 histogram says the cells would not fit otherwise
  // min_shift2: key bits that must be left below level 1 (8: k_local_sort's sub-bucket split)
  __shared__ uint32_t s_tmp[BINS / GX_WAVE + 1];
  HybridPlan& hy = plan->hy;
  FastPlan& hf   = plan->hf;
  const int t    = threadIdx.x;
  if (stage == 0) {
    if (hf.fail) {  // (float keys) the sample holds a NaN or a -0.0: the look-back path -- stable -- sorts the column
      if (t == 0) hf_give_up(plan, 0);
      return;
    }
    const unsigned long long V = hy.or_mask & hy.nor_mask;
    const int fh               = signed_keys ? fold_height(hy.fold_x, V, key_bits) : 0;  // (of the SAMPLE: level 0 reduces the exact one)
    const int top              = fh ? fh : (V ? 63 - __builtin_clzll(V) : 0);
    const int shift0           = top - 7;
    const int shift2           = shift0 - bits2;
    if (V == 0 || shift2 < min_shift2) {
      // all sampled keys equal: let the look-back path look at the whole column.  Too few varying bits below level 1 in
      // the sample (narrow key ranges: the LSD passes are the right tool): state 4 also spares the look-back path's
      // up-front read -- declining the hybrid path is always safe, the LSD passes sort anything
      // Round 5: when ALL varying bits are the low CS_MAXBITS or fewer (ids in [100, 10001): the reference benchmark's own
      // distribution, sort.cpp:24-26) keys-only sorting is a histogram + a fill -- state 5, k_cs_count / k_cs_scan / k_cs_fill.
      // The count verifies the sample's claim on every key; an outlier sends the column to the LSD passes (state 4).
      if (t == 0) {
        const int cs_top = V ? 63 - __builtin_clzll(V) : 0;
        const bool cs    = allow_counting && V != 0 && cs_top < CS_MAXBITS;
        const unsigned long long ormask = hy.or_mask;
        hf_give_up(plan, V == 0 ? 0 : (cs ? 5 : 4));
        if (cs) {
          hf.cs_bits = cs_top + 1;
          hf.cs_base = ormask & ~((1ull << (cs_top + 1)) - 1ull);
        }
      }
      return;
    }
    const bool spec_ok = shift0 == key_bits - 8;
    if (!spec_ok) {  // the speculative histogram is of the wrong digit: k_hf_sample<true> refills it
      for (int r = 0; r < NRANGE; ++r) hf.samp[r][t] = 0;
    }
    if (t == 0) {
      hf.hist_ready = spec_ok ? 1 : 0;
      hy.attempt   = 1;
      hy.fold      = fh ? 1 : 0;
      hy.shift0    = shift0;
      hy.bits2     = bits2;
      hy.shift2    = shift2;
      hy.cell_max  = cell_max;
      hy.need_hist = 0;
      hf.stride    = stride;
      hf.state     = 1;
    }
    return;
  }
  if (hf.state != 1) return;
  SplitPlan& sp = plan->sp;
  if (stage == 1) {
    if (hf.slots_ready) return;  // (the second launch, behind the splitter planning: nothing was asked for)
    if (sp.req < 0) {            // k_sp_plan could not make a table (one value in every sampled key ...): the LSD passes, as before round 5
      if (t == 0) hf_give_up(plan, 4);
      return;
    }
    // estimate of slot (range r, bin t) = sample count x (rows of the range / sampled rows of the range); capacity =
    // estimate + `margin` standard deviations of that estimate + two sample steps (a bin that is one contiguous run of the
    // input -- sorted or clustered keys -- is seen to within one step at either end) + a constant
    uint32_t cap[NRANGE];
    uint32_t sum = 0;
    for (int r = 0; r < NRANGE; ++r) {
      const uint32_t c = hf.samp[r][t];
      uint32_t total;
      (void)block_exclusive_scan<BINS>(c, 0u, SumOp(), s_tmp, &total);
      const int64_t rows = hf_range_rows(r, n, range_rows);
      const double scale = total ? (double)rows / (double)total : 0.0;
      double cp          = (double)c * scale + (double)margin * scale * __builtin_sqrt((double)c + 1.0) + 2.0 * stride * HF_CHUNK + 64.0;
      if (cp > (double)rows) cp = (double)rows;
      if (cp < 0.0) cp = 0.0;
      cap[r] = ((uint32_t)cp + 15u) & ~15u;  // slots start on 128-byte lines
      sum += cap[r];
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan<BINS>(sum, 0u, SumOp(), s_tmp, &total);
    if ((unsigned long long)total > slot_rows) {  // (cannot happen with the host's bound; checked all the same)
      if (t == 0) hf_give_up(plan, 0);
      return;
    }
    {
      // Hopeless already by the sample?  A bucket of `est` keys has (1 << bits2_max) cells of cell_max keys; what does not fit
      // them sits in cells that overflow (big cells), and big cells are sorted through X, which holds slot_rows / 2 keys.  When
      // the sample puts more than three quarters of the column there (Zipf-like values: 97 % of the keys below 2^24; values
      // spread around zero: two buckets of n / 2) the LSD passes will sort the column whatever level 0 finds: go there now
      // (state 4 also spares the look-back path's attempt).  Run 14: such columns cost 56 - 240 ms, most of it in the levels.
      double est = 0.0;
      for (int r = 0; r < NRANGE; ++r) {
        uint32_t tot_r;
        (void)block_exclusive_scan<BINS>(hf.samp[r][t], 0u, SumOp(), s_tmp, &tot_r);
        if (tot_r) est += (double)hf.samp[r][t] * ((double)hf_range_rows(r, n, range_rows) / (double)tot_r);
      }
      // (a bucket with twice the keys its cells hold is overfull in every cell when its keys are spread at all -- bell-shaped
      //  values -- and at least half of it is in big cells whatever they look like: counted whole)
      const double fits  = (double)((1ull << bits2_max) - 1ull) * (double)cell_max;
      const double over  = est > 2.0 * (fits + (double)cell_max) ? est : (est > fits + (double)cell_max ? est - fits : 0.0);
      const uint32_t o32 = (uint32_t)(over < 4.0e9 ? over : 4.0e9);
      uint32_t osum;
      (void)block_exclusive_scan<BINS>(o32 >> 4, 0u, SumOp(), s_tmp, &osum);
      // Round 5: what bit digits cannot split, SPLITTERS can.  The estimate says the big cells would not fit X (the rule stage 2
      // applies to the exact histogram): ask k_sp_plan for a splitter table; the sample histogram is taken again on the splitter
      // digit and this stage runs once more (its second launch).  With the tables in place (sp.on) the test is skipped: buckets are
      // balanced by construction and equality buckets never reach a cell.
      // (threshold: 3 % of the column in cells that must overflow.  The first cut used X's capacity, the rule stage 2 applies to the
      //  exact histogram; a bell-shaped column of 3e8 rows then stayed on the bit digits -- estimate below the bar -- overflowed
      //  thousands of cells at level 1 and was sorted by the LSD passes AFTER both levels had been spent: 16 ms against 3.5.
      //  Splitters cost a uniform column nothing -- it never gets here -- and an uneven one 1.2 ms at level 0.)
      if (!sp.on && allow_split && !sp.req && (double)osum * 16.0 > 0.03 * (double)n) {
        for (int r = 0; r < NRANGE; ++r) hf.samp[r][t] = 0;
        if (t == 0) sp.req = 1;
        return;
      }
      if (!sp.on && (double)osum * 16.0 > 0.75 * (double)n) {
        if (t == 0) hf_give_up(plan, 4);
        return;
      }
    }
    for (int r = 0; r < NRANGE; ++r) {  // the NRANGE slots of a bin are neighbours
      hf.slot0[r][t] = run;
      hf.cap0[r][t]  = cap[r];
      run += cap[r];
    }
    if (sp.on) {
      // splitter mode: the WARP of bucket t's cell map from the sampled masses of the sixteen pieces of its range (k_sp_sample: n / 32
      // keys, ~7600 per piece of an ordinary bucket -- the masses are known to about 1 %).  Piece j maps onto a stretch of the warped
      // fraction proportional to its mass (floored at 1/64 of an even share: keys the sample missed keep some resolution).
      // Narrow buckets, equality buckets and buckets the sample hardly saw keep the identity.
      uint32_t m[SP_NPIECE];
      uint32_t M = 0;
#pragma unroll
      for (int j = 0; j < SP_NPIECE; ++j) {
        m[j] = sp.sub[t][j];
        M += m[j];
      }
      uint32_t pfx = 320;
      if (sp.eq[t] || sp.w[t] <= 65535ull || M < 1024u) {
#pragma unroll
        for (int j = 0; j < SP_NPIECE; ++j) sp.wt[t][j] = make_uint2((uint32_t)j << SP_PSH, 1u << SP_PSH);
      } else {
        const uint32_t fl = M / (uint32_t)(SP_NPIECE * 64) + 1u;
        double tot        = 0.0;
#pragma unroll
        for (int j = 0; j < SP_NPIECE; ++j) {
          m[j] = m[j] < fl ? fl : m[j];
          tot += (double)m[j];
        }
        uint32_t y  = 0;
        double peak = 1.0;
#pragma unroll
        for (int j = 0; j < SP_NPIECE; ++j) {
          uint32_t d = (uint32_t)((double)m[j] / tot * 4294967040.0);  // (the sixteen of them add up to less than 2^32 - 256 + 16)
          d          = d < 1u ? 1u : d;
          sp.wt[t][j] = make_uint2(y, d);
          y += d;
          // what the warp cannot follow: how the density varies INSIDE a piece.  It is taken to stay below the denser of the two
          // neighbouring pieces' means (extrapolated at the bucket's ends): exact for a STEP inside the piece -- float keys: the density
          // over the sortable form doubles at every power of two, and with the milder "mean of the two pieces that meet at an edge"
          // of runs 13 - 19 every bucket that holds such a step overfilled a cell or two: N(0, 1) doubles 123 big cells, 1e6 keys
          // -- and each of them has its whole bucket read again by the rescue pass (0.8 ms) -- and twice the slope for a smooth
          // density (a bell-shaped tail: 1.04 instead of 1.02: 2 % more cells)
          if (m[j] * 256u >= M) {  // (a piece that holds a couple of cells or more)
            const double c  = (double)m[j];
            const double el = j > 0 ? (double)m[j - 1] : 2.0 * c - (double)m[j + 1];
            const double er = j < SP_NPIECE - 1 ? (double)m[j + 1] : 2.0 * c - (double)m[j - 1];
            const double r  = (el > er ? el : er) / c;
            peak            = r > peak ? r : peak;
          }
        }
        peak += 0.02;  // (the masses' own noise)
        peak = peak > 2.5 ? 2.5 : peak;
        pfx  = (uint32_t)(peak * 256.0);
      }
      sp.pf[t] = pfx;
    }
    if (t == 0) {  // level 0 reduces the EXACT masks into these
      hy.or_mask  = 0;
      hy.nor_mask = 0;
      hy.fold_x   = 0;
      hf.slot_total = total;
      hf.slots_ready = 1;
    }
    return;
  }
  // ---- stage 2
  const unsigned long long V = hy.or_mask & hy.nor_mask;
  uint32_t cnt[NRANGE];
  uint32_t c   = 0;
  // (sharded sort: the digits come from the masks of ALL ranks; this rank's keys may well span fewer bits)
  // the digit positions came from the sample; the exact masks must agree: the top varying bit -- or, under the sign fold, the
  // height of the data below the sign copies and a sign that does vary
  const int top_exact = hy.fold ? (((V >> (key_bits - 1)) & 1ull) && hy.fold_x ? 64 - __builtin_clzll(hy.fold_x) : -1) : 63 - __builtin_clzll(V | 1ull);
  // (splitter mode: no bit position was assumed -- the slot counts below are the whole check)
  int bad      = hf.fail != 0 || (!hf.forced_masks && !sp.on && (V == 0 || top_exact != hy.shift0 + 7));
  // sharded sort: a bit that varies among THIS rank's keys must vary in the all-gathered SAMPLE masks.  That is exact: a bit that
  // varies over all ranks but in no sample has, on some rank, a key that differs in it from that rank's own samples (every rank
  // with rows has at least one), i.e. it varies inside that rank.  (V is the same set on raw and on sortable keys -- they differ
  // by a constant XOR -- so level 0's raw masks compare with the sample's sortable ones.)
  if (hf.forced_masks && (V & ~(hf.forced_or & hf.forced_nor))) bad = 1;
  for (int r = 0; r < NRANGE; ++r) {
    cnt[r] = hf.cur0[r][t];
    if (cnt[r] > hf.cap0[r][t]) bad = 1;
    c += cnt[r];
  }
  uint32_t total;
  const uint32_t exc = block_exclusive_scan<BINS>(c, 0u, SumOp(), s_tmp, &total);
  if ((int64_t)total != n) bad = 1;
  if (__syncthreads_or(bad)) {
    if (t == 0) hf_give_up(plan, 2);
    return;
  }
  hy.hist0[t] = c;
  hy.gbin0[t] = exc;
  // (splitter mode) an equality bucket never reaches a cell: it counts as empty in everything that sizes or judges the cells
  const uint32_t cc = (sp.on && sp.eq[t]) ? 0u : c;
  if (sp.on) {
    // every bucket gets as many of the 1 << bits2_max cell slots as its EXACT size asks for at ~7400 keys per cell (a narrow bucket:
    // all of them -- one value per cell); the tables and grids behind are indexed with bits2_max
    const uint32_t ncmax = 1u << bits2_max;
    uint32_t nc          = (uint32_t)(((unsigned long long)cc * sp.pf[t] / 256ull + 7399ull) / 7400ull);  // sized for the bucket's fullest cell
    nc                   = nc < 1u ? 1u : (nc > ncmax ? ncmax : nc);
    if (sp.nw[t] != 0u && sp.nw[t] <= ncmax && !sp.eq[t]) nc = ncmax;
    // a range of few values (<= 8 per cell slot): a cell holds a whole number of values, so with cells sized for the MEAN a cell of
    // 3 values beside cells of 2 overflows (Zipf-like columns: 4 % of the keys) -- take every slot: 1 - 8 values per cell
    if (sp.w[t] <= 8ull * ncmax) nc = ncmax;
    sp.nc[t] = nc;
    if (t == 0) {
      hy.bits2  = bits2_max;
      hy.shift2 = 64;
    }
    __syncthreads();
  }
  // bits2 came from n alone, i.e. from buckets of n / 256 keys.  Keys whose range is not a power of two (ids below 1e12,
  // timestamps: the top digit uses 233 of 256 bins) have fuller buckets, and at the upper end of a size class -- 1e9 rows sit at
  // 93 % of one -- their cells overflow and the column falls back to the LSD passes (3x slower).  The exact histogram is here:
  // take one more level-1 bit when the fullest bucket needs it (the launches behind are sized for it).
  {
    // (round 4: at least FOUR buckets must be that full.  One hot value -- 1e6 copies of a key -- fills its bucket without filling
    //  that bucket's ordinary cells; its one big cell is sorted on its own (HybridPlan::big), and the extra bit would have
    //  halved every cell of the column for it: local stage 3.3 -> 4.8 ms, profiles/r4_run4_sort_hot1e6_kernel_stats.txt)
    const uint32_t fit = (uint32_t)(0.97 * (double)cell_max);  // mean cell of a bucket; a cell spreads 4.5 sigma = 5 % above it
    const int full     = (!sp.on && ((unsigned long long)cc >> hy.bits2) > (unsigned long long)fit) ? 1 : 0;
    const int more     = __syncthreads_count(full) >= 4;
    if (more && t == 0 && hy.bits2 < bits2_max && hy.shift2 - 1 >= min_shift2) {
      hy.bits2 += 1;
      hy.shift2 -= 1;
    }
    __syncthreads();
  }
  {
    // the same test on the EXACT level-0 histogram, against what X holds (k_big_plan: slot_rows / 2): a lower bound of the keys
    // in big cells that does not fit X means the whole-column LSD fallback is certain -- level 1 (and, for such columns, its
    // thousands of tiles that all bump the cursors of the same few cells: 205 ms for the Zipf-like column of run 15), the cell
    // sort and the big-cell machinery are skipped
    const unsigned long long ncb  = sp.on ? (unsigned long long)sp.nc[t] : (1ull << hy.bits2);
    const unsigned long long fits = (ncb - 1ull) * (unsigned long long)cell_max;
    const unsigned long long full = fits + (unsigned long long)cell_max;
    const uint32_t over           = (unsigned long long)cc > 2ull * full ? cc : ((unsigned long long)cc > full ? (uint32_t)((unsigned long long)cc - fits) : 0u);
    uint32_t osum;
    (void)block_exclusive_scan<BINS>(over >> 4, 0u, SumOp(), s_tmp, &osum);
    if ((unsigned long long)osum * 16ull > slot_rows / 2) {
      if (t == 0) hf_give_up(plan, 4);
      return;
    }
  }
  // (splitter mode: a narrow bucket is only counted and an empty one holds nothing -- one cell's worth of slot space)
  // (... and a slot holds the bucket's FULLEST cell -- pf x the mean: until run 12 the slots were sized for the mean, and the dense-end cells
  //  of every bucket with pf > 1.08 overflowed them by construction: 1 - 2.5 % of the keys of bell-shaped / float columns went through X)
  const unsigned long long cc_pf = (unsigned long long)cc * (sp.on ? sp.pf[t] : 256u) / 256ull;
  const uint32_t cc_slots        = cc_pf < 0xFFFFFFFFull ? (uint32_t)cc_pf : 0xFFFFFFFFu;
  plan_cell_slots(hy, cc_slots, hy.bits2, cell_max, s_tmp, cell_budget,
                  sp.on ? ((cc == 0u || (sp.nw[t] != 0u && sp.nw[t] <= (1u << bits2_max) && !sp.eq[t])) ? 1u : sp.nc[t]) : 0u, sp.on ? cc : 0xFFFFFFFFu);
  uint32_t tiles = 0;
  for (int r = 0; r < NRANGE; ++r) tiles += (cnt[r] + (uint32_t)tile_rows - 1) / (uint32_t)tile_rows;
  uint32_t ttotal;
  uint32_t trun = block_exclusive_scan<BINS>(tiles, 0u, SumOp(), s_tmp, &ttotal);
  for (int r = 0; r < NRANGE; ++r) {  // a tile never straddles two regions
    const int q     = t * NRANGE + r;
    hf.reg_tile0[q] = trun;
    hf.reg_start[q] = hf.slot0[r][t];
    hf.reg_count[q] = cnt[r];
    trun += (cnt[r] + (uint32_t)tile_rows - 1) / (uint32_t)tile_rows;
  }
  if (t == 0) {
    hf.reg_tile0[BINS * NRANGE] = ttotal;
    // (splitter mode: shift2 = 64 -- the cells share no bit prefix: the stable passes of k_local_sort (crowded cells) take every varying byte)
    plan_local_digits(hy, V, hy.shift2);
    __threadfence();
    hf.state = 3;
  }
}

// ------------------------------------------------------------------------------------------
// Splitter planning (SplitPlan above): ONE workgroup of 1024 threads.  16384 keys at even strides of the input -> bitonic
// sort in LDS -> quantiles every 64th key (every 128th when the equality buckets would make more than 255 splitters) ->
// splitter table, per-bucket affine maps, the LUT.  ~0.2 ms (profiles/r4_run31_xp_splitter_level0.txt); runs only when
// k_hf_plan stage 1 asked for it.
// ------------------------------------------------------------------------------------------
template <int KIND>
__global__ void __launch_bounds__(1024) k_sp_plan(const uint64_t* __restrict__ in, int64_t n, uint64_t desc_mask, SortPlan* plan, int bits2)
{
  SplitPlan& sp = plan->sp;
  if (plan->hf.state != 1 || !sp.req || sp.on) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* s = reinterpret_cast<unsigned long long*>(smem);  // [SP_NSAMP]
  __shared__ uint32_t s_wsum[1024 / GX_WAVE + 1];
  __shared__ uint32_t s_total;
  __shared__ uint32_t s_idx[3][BINS];
  __shared__ uint32_t s_worst[3];
  __shared__ unsigned long long s_tab[BINS];
  __shared__ uint32_t s_handled[BINS];
  const int tid = threadIdx.x;
  for (int i = tid; i < SP_NSAMP; i += 1024) s[i] = to_sortable<uint64_t, KIND>(in[(int64_t)(((unsigned long long)i * (unsigned long long)n) / SP_NSAMP)], desc_mask);
  __syncthreads();
  for (int k = 2; k <= SP_NSAMP; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int p = tid; p < SP_NSAMP / 2; p += 1024) {
        const int a   = ((p & ~(j - 1)) << 1) | (p & (j - 1));  // index with bit j clear
        const int b   = a | j;
        const bool up = (a & k) == 0;
        const unsigned long long x = s[a], y = s[b];
        if ((x > y) == up) {
          s[a] = y;
          s[b] = x;
        }
      }
      __syncthreads();
    }
  }
  // quantile t = s[qpos(t)]: every `step`-th sampled key, and a GEOMETRIC refinement at either end (positions 2, 4, .., 32 and
  // NS - 32, .., NS - 2): the outermost buckets are where a smooth density varies by orders of magnitude across one bucket -- equal-
  // width cells cannot follow that (a bell-shaped column: 1.5 % of the keys in overfull cells of the last few buckets, all through
  // the big-cell path) -- so they are cut until the open-ended first / last bucket holds 2 / 16384 of the keys.
  // Thread t emits q when it starts a run of equal quantiles, and q + 1 when it ends a run of
  // >= 2 (a value that holds more than one quantile: an equality bucket [q, q + 1)) unless the next distinct quantile is q + 1
  uint32_t nsp = 0;
  for (int step = 68; step <= 136; step *= 2) {
    const int nreg = (SP_NSAMP - 64) / step;  // regular quantiles: 240, then 120
    const int nq   = nreg + 10;               // + 5 at either end
    auto qpos = [&](int t) -> int { return t < 5 ? (2 << t) : (t < 5 + nreg ? step * (t - 4) : SP_NSAMP - (32 >> (t - 5 - nreg))); };
    unsigned long long q = 0, qprev = 0, qnext = 0;
    bool first = false, eq_end = false;
    if (tid < nq) {
      q     = s[qpos(tid)];
      qprev = tid > 0 ? s[qpos(tid - 1)] : 0;
      qnext = tid < nq - 1 ? s[qpos(tid + 1)] : 0;
      first = tid == 0 || q != qprev;
      const bool last   = tid == nq - 1 || q != qnext;
      const bool in_run = (tid > 0 && q == qprev) || (tid < nq - 1 && q == qnext);
      eq_end            = last && in_run && q != ~0ull && !(tid < nq - 1 && qnext == q + 1);
    }
    const uint32_t mine = (first ? 1u : 0u) + (eq_end ? 1u : 0u);
    const uint32_t incl = wave_inclusive_scan(mine, SumOp());
    if ((tid & (GX_WAVE - 1)) == GX_WAVE - 1) s_wsum[tid / GX_WAVE] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < tid / GX_WAVE; ++w) base += s_wsum[w];
    const uint32_t excl = base + incl - mine;
    if (tid == 1023) s_total = excl + mine;
    if (tid < BINS) s_tab[tid] = ~0ull;
    __syncthreads();
    nsp = s_total;
    if (nsp <= (uint32_t)BINS - 1u) {  // block-uniform
      if (tid < nq) {
        uint32_t o = excl;
        if (first) s_tab[o++] = q;
        if (eq_end) s_tab[o] = q + 1;
      }
      __syncthreads();
      break;
    }
    __syncthreads();
  }
  if (nsp > (uint32_t)BINS - 1u || nsp == 0) {  // (127 quantiles make at most 254 entries; nsp == 0: every sampled key is the same value)
    if (tid == 0) sp.req = -1;                  // stage 1's second launch then sends the column to the LSD passes, as before round 5
    return;
  }
  // ---- GAP splitters.  A bucket whose sampled keys leave more than half of its extent empty in ONE gap straddles two clusters (int
  // keys around two far-apart centres; float keys of both signs: nothing lies between -tiny and +tiny, 2^63 apart in sortable form).
  // No map over its range can cut such a bucket into even cells -- its keys sit at the two ends (two clusters, run 13: 4 cells of 4.8e6
  // keys through X; N(0, 1) doubles: 157 cells, 5e6 keys) -- so the gap gets a bucket of its own: two more splitters, at most twice.
  {
    __shared__ unsigned long long s_g64[1024 / GX_WAVE];
    __shared__ uint32_t s_gwin[1024 / GX_WAVE];
    __shared__ unsigned long long s_gv[2];
    __shared__ uint32_t s_gt;
    auto lower = [&](unsigned long long v) {  // first sample index whose key is >= v
      int a = 0, e = SP_NSAMP;
      while (a < e) {
        const int mid = (a + e) >> 1;
        if (s[mid] < v) a = mid + 1; else e = mid;
      }
      return a;
    };
    for (int round = 0; round < 2 && nsp + 2u <= (uint32_t)BINS - 1u; ++round) {  // (block-uniform)
      unsigned long long g = 0ull, ga = 0ull, gb = 0ull;
      if (tid <= (int)nsp) {
        const unsigned long long tlo = tid == 0 ? 0ull : s_tab[tid - 1];
        const int ia = tid == 0 ? 0 : lower(tlo), ib = tid == (int)nsp ? SP_NSAMP : lower(s_tab[tid]);
        if (ib - ia >= 24) {  // (a tail bucket of a handful of sampled keys has nothing but gaps)
          for (int i = ia; i + 1 < ib; ++i) {
            const unsigned long long a = s[i], d = s[i + 1] - a;
            if (d > g) {
              g  = d;
              ga = a;
              gb = a + d;
            }
          }
          if (g < (1ull << 20) || g <= (s[ib - 1] - s[ia]) / 2ull) g = 0ull;
        }
      }
      const unsigned long long wg = wave_reduce(g, [](unsigned long long x, unsigned long long y) { return x > y ? x : y; });
      if ((tid & (GX_WAVE - 1)) == 0) s_g64[tid / GX_WAVE] = wg;
      __syncthreads();
      unsigned long long gmax = 0ull;
      for (int w = 0; w < 1024 / GX_WAVE; ++w) gmax = s_g64[w] > gmax ? s_g64[w] : gmax;
      if (gmax == 0ull) break;  // (block-uniform)
      const uint32_t cand = (g == gmax) ? (uint32_t)tid : 0xFFFFFFFFu;  // the first bucket that has it
      const uint32_t wc   = wave_reduce(cand, [](uint32_t x, uint32_t y) { return x < y ? x : y; });
      if ((tid & (GX_WAVE - 1)) == 0) s_gwin[tid / GX_WAVE] = wc;
      __syncthreads();
      uint32_t win = 0xFFFFFFFFu;
      for (int w = 0; w < 1024 / GX_WAVE; ++w) win = s_gwin[w] < win ? s_gwin[w] : win;
      if ((uint32_t)tid == win) {
        s_gv[0] = ga + 1ull;
        s_gv[1] = gb;
        s_gt    = (uint32_t)tid;
      }
      const unsigned long long mine = tid < (int)nsp ? s_tab[tid] : ~0ull;
      __syncthreads();
      const uint32_t t = s_gt;  // both new splitters lie inside bucket t: the entries from t on move up by two
      if (tid < (int)nsp && (uint32_t)tid >= t) s_tab[tid + 2] = mine;
      if (tid == 0) {
        s_tab[t]      = s_gv[0];
        s_tab[t + 1u] = s_gv[1];
      }
      nsp += 2u;
      __syncthreads();
    }
  }
  const unsigned long long kmin = s[0], kmax = s[SP_NSAMP - 1];
  // ---- per-bucket maps.  Bucket b holds the keys in [T(b - 1), T(b)) with T(-1) = 0 and T(nsp) = 2^64; the cell map of the first /
  // last bucket is laid over what the SAMPLE saw of it (keys beyond clamp to the end cells -- still monotone)
  bool myeq     = false;
  uint32_t mynw = 0;
  if (tid <= (int)nsp) {
    const unsigned long long tlo = tid == 0 ? 0ull : s_tab[tid - 1];
    const bool lastb             = tid == (int)nsp;
    const unsigned long long thi = lastb ? 0ull : s_tab[tid];  // (0 stands for 2^64)
    const bool eq                = lastb ? tlo == ~0ull : thi - tlo == 1ull;
    myeq                         = eq;
    unsigned long long lo = tlo, hi = thi;
    if (tid == 0 && kmin < (lastb ? ~0ull : thi)) lo = kmin;
    if (lastb) hi = kmax == ~0ull ? ~0ull : kmax + 1ull;
    if (hi <= lo) hi = lo + 1ull;
    const unsigned long long w  = hi - lo;
    const unsigned long long wm = w - 1ull;  // the range's last key, relative
    int nsh       = 63;
    uint32_t mlow = 0;
    if (wm != 0ull) {
      const int wl = 64 - __builtin_clzll(wm);
      nsh          = wl - 31;  // >= 0: right shift; < 0: left shift
      const uint32_t wx = nsh >= 0 ? (uint32_t)(wm >> nsh) : (uint32_t)(wm << (-nsh));  // bit 30 set
      const unsigned long long M = (1ull << 63) / ((unsigned long long)wx + 1ull);      // [2^32, 2^33)
      mlow                       = (uint32_t)(M - (1ull << 32));
    }
    const uint32_t pfx = 320;  // (k_hf_plan stage 1 measures it on the n / 32 sample; this is what a bucket without enough sampled keys keeps)
    // interior buckets map exactly their true range; the first / last one only when it is a single value
    const bool pure = (tid > 0 && !lastb) || eq;
    mynw            = pure && w <= 65535ull ? (uint32_t)w : 0u;
    sp.tab[tid]  = tid < (int)nsp ? s_tab[tid] : ~0ull;
    sp.lo[tid]   = lo;
    sp.w[tid]    = w;
    sp.nsh[tid]  = nsh;
    sp.mlow[tid] = mlow;
    sp.eq[tid]   = eq ? 1u : 0u;
    sp.nw[tid]   = mynw;
    sp.pf[tid]   = pfx;
  } else if (tid < BINS) {
    sp.pf[tid]   = 256;
    sp.tab[tid]  = ~0ull;
    sp.lo[tid]   = 0ull;
    sp.w[tid]    = 1ull;
    sp.nsh[tid]  = 63;
    sp.mlow[tid] = 0;
    sp.eq[tid]   = 0;
    sp.nw[tid]   = 0;
  }
  if (tid < BINS) s_handled[tid] = (myeq || (mynw != 0u && mynw <= (1u << bits2))) ? 1u : 0u;  // buckets that never reach a cell
  const uint32_t neq = (uint32_t)__syncthreads_count(myeq);
  // ---- will it pay?  A value that the 16384-key sample holds twice occurs n / 8192 times or more: it overfills a cell on its own
  // unless its bucket is an equality or a narrow bucket (those are counted and filled).  When such keys are a large part of the
  // column the big-cell path (X, sorted by LSD passes) would carry it -- slower than declining (a Zipf-like column: 29.6 ms against
  // 23.8 in the first run).  Decline then: the LSD passes sort the column as they did before round 5.
  {
    uint32_t xc = 0;
    for (int i = tid; i < SP_NSAMP; i += 1024) {
      const unsigned long long k = s[i];
      const bool dup = (i > 0 && s[i - 1] == k) || (i + 1 < SP_NSAMP && s[i + 1] == k);
      if (dup) {
        uint32_t a = 0, b = nsp;  // bucket = number of splitters <= k
        while (a < b) {
          const uint32_t mid = (a + b) >> 1;
          if (s_tab[mid] <= k) a = mid + 1; else b = mid;
        }
        if (!s_handled[a]) ++xc;
      }
    }
    xc = wave_reduce(xc, SumOp());
    if ((tid & (GX_WAVE - 1)) == 0) s_wsum[tid / GX_WAVE] = xc;
    __syncthreads();
    uint32_t tot = 0;
    for (int w = 0; w < 1024 / GX_WAVE; ++w) tot += s_wsum[w];
    if (tot > (uint32_t)(SP_NSAMP * 3 / 10)) {  // block-uniform
      if (tid == 0) sp.req = -1;
      return;
    }
  }
  // ---- the LUT: cells cut on rel = key - kmin LINEARLY (even or bell-shaped densities) or LOGARITHMICALLY (power laws) or, for float keys
  // of both signs, linearly per sign (SpLutF); the form whose fullest cell holds the fewest splitters wins (ties: the lower form)
  constexpr int NF = 3;
  uint32_t lshift  = 0;
  while (lshift < 63 && ((kmax - kmin) >> lshift) >= (unsigned long long)SP_NLUT) ++lshift;
  SpLutF lf{0ull, 0ull, 0u, 0u};
  const bool both = nsp >= 2u;
  if (both) {  // (block-uniform) the widest gap between neighbouring splitters: every thread walks the table (<= 254 entries, broadcast reads)
    uint32_t gi          = 0;
    unsigned long long g = 0ull;
    for (uint32_t i = 0; i + 1u < nsp; ++i) {
      const unsigned long long d = s_tab[i + 1] - s_tab[i];
      if (d > g) {
        g  = d;
        gi = i;
      }
    }
    const unsigned long long alast = s_tab[gi] > kmin ? s_tab[gi] : kmin;
    lf.nmin = kmin;
    lf.pmin = s_tab[gi + 1];
    while (lf.nsh < 63u && ((alast - kmin) >> lf.nsh) >= (unsigned long long)(SP_NLUT / 2)) ++lf.nsh;
    while (lf.psh < 63u && kmax > lf.pmin && ((kmax - lf.pmin) >> lf.psh) >= (unsigned long long)(SP_NLUT / 2)) ++lf.psh;
  }
  if (tid < NF) s_worst[tid] = (tid == 2 && !both) ? 0xFFFFFFFFu : 0u;
  for (int form = 0; form < NF; ++form)
    if (tid < BINS) s_idx[form][tid] = (uint32_t)tid < nsp ? sp_lut_cell_k<KIND>(s_tab[tid], kmin, (uint32_t)form, lshift, lf) : 0xFFFFFFFFu;
  __syncthreads();
  uint32_t lo2[NF][SP_NLUT / 1024];
  for (int form = 0; form < NF; ++form) {
    for (int r = 0; r < SP_NLUT / 1024; ++r) {
      const uint32_t c = (uint32_t)tid + 1024u * r;
      uint32_t a = 0, b = nsp;  // splitters in cells below c = lower_bound(s_idx, c)
      while (a < b) {
        const uint32_t mid = (a + b) / 2;
        if (s_idx[form][mid] < c) a = mid + 1; else b = mid;
      }
      uint32_t a2 = a, b2 = nsp;  // ... in cell c: upper_bound - lower_bound
      while (a2 < b2) {
        const uint32_t mid = (a2 + b2) / 2;
        if (s_idx[form][mid] <= c) a2 = mid + 1; else b2 = mid;
      }
      const uint32_t ins = a2 - a;
      lo2[form][r]       = a | ((ins < 63u ? ins : 63u) << 9) | (ins == 0u ? 0x8000u : 0u);
      atomicMax(&s_worst[form], ins);
    }
  }
  __syncthreads();
  int pick = s_worst[1] < s_worst[0] ? 1 : 0;
  if (s_worst[2] < s_worst[pick]) pick = 2;
  for (int r = 0; r < SP_NLUT / 1024; ++r) sp.lut[tid + 1024 * r] = (uint16_t)(pick == 0 ? lo2[0][r] : (pick == 1 ? lo2[1][r] : lo2[NF - 1][r]));
  if (tid == 0) {
    sp.nsp     = nsp;
    sp.neq     = neq;
    sp.kmin    = kmin;
    sp.lut_log = (uint32_t)pick;
    sp.lut_steps = s_worst[pick];
    sp.lshift  = lshift;
    sp.f_nmin  = lf.nmin;
    sp.f_pmin  = lf.pmin;
    sp.f_nsh   = lf.nsh;
    sp.f_psh   = lf.psh;
    plan->hy.fold = 0;  // (the sign fold is a property of the bit digit)
    __threadfence();
    sp.on = 1;
  }
}

// the sample histogram once more, on the SPLITTER digit (same chunks and ranges as k_hf_sample<true>): stage 1 sizes the level-0 slots from it
template <int KIND>
__global__ void __launch_bounds__(256) k_sp_sample(const uint64_t* __restrict__ in, int64_t n, uint64_t desc_mask, SortPlan* plan, int stride, int64_t range_rows)
{
  FastPlan& hf        = plan->hf;
  const SplitPlan& sp = plan->sp;
  if (hf.state != 1 || !sp.on || hf.slots_ready) return;
  __shared__ uint32_t s_hist[NRANGE * BINS];
  __shared__ unsigned long long s_tab[BINS];
  __shared__ uint16_t s_lut[SP_NLUT];
  __shared__ uint32_t s_sub[BINS * SP_NPIECE];  // sampled keys per (bucket, sixteenth of its range): the warp of the cell maps (stage 1)
  __shared__ unsigned long long s_lo[BINS], s_w[BINS];
  __shared__ uint32_t s_mlow[BINS];
  __shared__ int32_t s_nsh[BINS];
  const unsigned tid = threadIdx.x, lane = lane_id();
  for (int i = tid; i < NRANGE * BINS; i += 256) s_hist[i] = 0;
  for (int i = tid; i < BINS * SP_NPIECE; i += 256) s_sub[i] = 0;
  s_tab[tid]  = sp.tab[tid];
  s_lo[tid]   = sp.lo[tid];
  s_w[tid]    = sp.w[tid];
  s_mlow[tid] = sp.mlow[tid];
  s_nsh[tid]  = sp.nsh[tid];
  for (int i = tid; i < SP_NLUT; i += 256) s_lut[i] = sp.lut[i];
  __syncthreads();
  const uint32_t nsp = sp.nsp, lut_log = sp.lut_log, lshift = sp.lshift;
  const unsigned long long kmin = sp.kmin;
  const SpLutF lf               = sp_lutf_of(sp);
  const int64_t step    = (int64_t)stride * HF_CHUNK;
  const int64_t nchunks = div_up(n, step);
  const int64_t nw      = (int64_t)gridDim.x * 4;
  constexpr int U       = 8;
  for (int64_t c0 = (int64_t)blockIdx.x * 4 + tid / GX_WAVE; c0 < nchunks; c0 += nw * U) {
    uint64_t raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = (c0 + u * nw) * step + lane;
      raw[u]            = (c0 + u * nw < nchunks && row < n) ? in[row] : 0ull;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = (c0 + u * nw) * step + lane;
      const bool live   = c0 + u * nw < nchunks && row < n;
      const uint64_t k  = to_sortable<uint64_t, KIND>(raw[u], desc_mask);
      const int64_t r64 = range_rows > 0 ? row / range_rows : (int64_t)(NRANGE - 1);
      const int r       = r64 < NRANGE - 1 ? (int)r64 : NRANGE - 1;
      const uint32_t b  = sp_bucket<KIND>(s_tab, s_lut, k, nsp, kmin, lut_log, lshift, lf);
      (void)lds_rank(s_hist + r * BINS, b, live);
      if (live) atomicAdd(&s_sub[b * SP_NPIECE + (sp_frac(SpCell{s_lo[b], s_w[b], s_mlow[b], s_nsh[b]}, k) >> SP_PSH)], 1u);
    }
  }
  __syncthreads();
  for (int i = tid; i < NRANGE * BINS; i += 256) {
    const uint32_t c = s_hist[i];
    if (c) atomicAdd(&hf.samp[i / BINS][i % BINS], c);
  }
  for (int i = tid; i < BINS * SP_NPIECE; i += 256) {
    const uint32_t c = s_sub[i];
    if (c) atomicAdd(&plan->sp.sub[i / SP_NPIECE][i % SP_NPIECE], c);
  }
}

// NARROW buckets of the splitter mode (SplitPlan::nw): the output range of bucket b is filled from its cell starts (k_plan2) --
// cell c holds the ONE value of the bucket's range that the cell map sends to c.  grid (BINS, SP_FILL_Y): workgroup (b, y) takes
// every SP_FILL_Y-th tile of CS_TILE output keys of bucket b.
constexpr int SP_FILL_Y = 16;
constexpr int CS_TILE   = 4096;  // output elements per fill tile (k_sp_fill, k_cs_fill)
template <int KIND>
__global__ void __launch_bounds__(256) k_sp_fill(uint64_t* __restrict__ out, uint64_t desc_mask, const SortPlan* plan, const uint32_t* __restrict__ cellstart)
{
  const HybridPlan& hy = plan->hy;
  const SplitPlan& sp  = plan->sp;
  if (plan->hf.state != 3 || !sp.on || !hy.ok) return;
  const uint32_t b = blockIdx.x;
  const int bits2  = hy.bits2;
  if (!sp_narrow(sp, b, bits2)) return;
  __shared__ uint32_t s_start[NB2MAX + 1];
  __shared__ uint16_t s_valj[NB2MAX];
  __shared__ uint32_t s_g[2];
  const uint32_t ncell = 1u << bits2;
  const uint32_t first = hy.gbin0[b], end = first + hy.hist0[b];
  const SpCell spc     = sp_cell_of(sp, b);
  const uint32_t spnc  = sp.nc[b];  // (= ncell for a narrow bucket)
  for (uint32_t c = threadIdx.x; c < ncell; c += 256) s_start[c] = cellstart[b * NB2MAX + c];
  if (threadIdx.x == 0) s_start[ncell] = end;
  for (uint32_t j = threadIdx.x; j < sp.nw[b]; j += 256) s_valj[sp_cell(sp_frac(spc, spc.lo + j), spnc)] = (uint16_t)j;  // (injective: nw <= ncell)
  __syncthreads();
  for (uint32_t lo = first + blockIdx.y * CS_TILE; lo < end; lo += SP_FILL_Y * CS_TILE) {
    const uint32_t hi = lo + CS_TILE < end ? lo + CS_TILE : end;
    if (threadIdx.x < 2) {  // last cell whose start is <= position (lo | hi - 1): empty cells share their successor's start and are skipped
      const uint32_t pos = threadIdx.x == 0 ? lo : hi - 1;
      uint32_t a = 0, e = ncell;
      while (e - a > 1) {
        const uint32_t mid = (a + e) >> 1;
        if (s_start[mid] <= pos) a = mid; else e = mid;
      }
      s_g[threadIdx.x] = a;
    }
    __syncthreads();
    const uint32_t g0 = s_g[0], g1 = s_g[1];
    if (g0 == g1) {
      const uint64_t v = from_sortable<uint64_t, KIND>(spc.lo + s_valj[g0], desc_mask);
      for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) __builtin_nontemporal_store(v, &out[i]);
    } else {
      for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
        uint32_t a = g0, e = g1 + 1;
        while (e - a > 1) {
          const uint32_t mid = (a + e) >> 1;
          if (s_start[mid] <= i) a = mid; else e = mid;
        }
        __builtin_nontemporal_store(from_sortable<uint64_t, KIND>(spc.lo + s_valj[a], desc_mask), &out[i]);
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// Round 5: COUNTING SORT of keys whose varying bits are the low CS_MAXBITS or fewer (FastPlan::state 5) -- keys only, so
// the sorted column is fully described by how often each value occurs: one read of the column (a histogram in LDS), a
// scan of <= 32768 counters, one write (every output tile looks up the value runs that cover it).  24 -> 16 B/row against
// the two LSD passes such a column took (13.6 ms per 1e9 rows for the reference benchmark's [100, 10001) keys), and the
// LSD plan's own 8 B/row histogram read is skipped as well.  cub::DeviceRadixSort behind cudf::sort has no such path
// (cpp/src/sort/sort_radix.cu:52-161); the result is the same bytes.
// ------------------------------------------------------------------------------------------
constexpr int CS_BT      = 1024;
template <typename KeyT, int KIND>
__global__ void __launch_bounds__(CS_BT) k_cs_count(const KeyT* __restrict__ in, int64_t n, KeyT desc_mask, SortPlan* plan, uint32_t* __restrict__ ghist)
{
  FastPlan& hf = plan->hf;
  if (hf.state != 5) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* s_hist   = reinterpret_cast<uint32_t*>(smem);
  const int bits     = hf.cs_bits;
  const uint32_t nb  = 1u << bits;
  const KeyT lowmask = (KeyT)(nb - 1u);
  const KeyT base    = (KeyT)hf.cs_base;
  for (uint32_t i = threadIdx.x; i < nb; i += CS_BT) s_hist[i] = 0;
  __syncthreads();
  constexpr int VEC = 16 / sizeof(KeyT);  // keys per 16-byte load
  typedef KeyT VecT __attribute__((ext_vector_type(VEC)));
  KeyT bad = 0;
  auto count1 = [&](KeyT raw) {
    const KeyT k = to_sortable<KeyT, KIND>(raw, desc_mask);
    bad |= (KeyT)((k ^ base) & (KeyT)~lowmask);
    atomicAdd(&s_hist[(uint32_t)(k & lowmask)], 1u);
  };
  // (a sliced column may start at any element: the keys before the first 16-byte boundary and behind the last one go one by one)
  int64_t head = (int64_t)(((16u - (uint32_t)(reinterpret_cast<uintptr_t>(in) & 15u)) & 15u) / sizeof(KeyT));
  if (head > n) head = n;
  const int64_t nvec   = (n - head) / VEC;
  const VecT* vin      = reinterpret_cast<const VecT*>(in + head);
  const int64_t stride = (int64_t)gridDim.x * CS_BT;
  constexpr int U = 4;  // independent 16-byte loads in flight per lane
  int64_t i = (int64_t)blockIdx.x * CS_BT + threadIdx.x;
  for (; i + (U - 1) * stride < nvec; i += U * stride) {
    VecT v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(&vin[i + u * stride]);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < VEC; ++e) count1(v[u][e]);
  }
  for (; i < nvec; i += stride) {
    const VecT v = vin[i];
#pragma unroll
    for (int e = 0; e < VEC; ++e) count1(v[e]);
  }
  if (blockIdx.x == 0) {
    const int64_t tail0 = head + nvec * VEC;
    if ((int64_t)threadIdx.x < head) count1(in[threadIdx.x]);
    if ((int64_t)threadIdx.x < n - tail0) count1(in[tail0 + threadIdx.x]);
  }
  if (__syncthreads_or(bad != 0)) {  // (also the barrier before the flush)
    if (threadIdx.x == 0 && !__hip_atomic_load(&hf.cs_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicExch(&hf.cs_fail, 1u);
  }
  for (uint32_t b = threadIdx.x; b < nb; b += CS_BT) {
    const uint32_t c = s_hist[b];
    if (c) atomicAdd(&ghist[b], c);
  }
}

// one workgroup: counts -> first output position of every NON-EMPTY value (nz_start[g], nz_val[g], g < groups; nz_start[groups] = n);
// the verdict: every key inside the sampled range and the counts add up -> the LSD passes become no-ops; else state 4
__global__ void __launch_bounds__(CS_BT) k_cs_scan(SortPlan* plan, const uint32_t* __restrict__ ghist, uint32_t* __restrict__ nz_start,
                                                   uint32_t* __restrict__ nz_val, int64_t n, int npass)
{
  FastPlan& hf = plan->hf;
  if (hf.state != 5) return;
  __shared__ uint32_t s_tmp[CS_BT / GX_WAVE + 1];
  const uint32_t nb  = 1u << hf.cs_bits;
  const uint32_t per = (nb + CS_BT - 1) / CS_BT;  // consecutive bins per thread (<= 32)
  const uint32_t b0  = threadIdx.x * per;
  uint32_t sum = 0, cnt = 0;
  for (uint32_t k = 0; k < per; ++k) {
    const uint32_t b = b0 + k;
    const uint32_t c = b < nb ? ghist[b] : 0u;
    sum += c;
    cnt += c ? 1u : 0u;
  }
  uint32_t total, groups;
  uint32_t run = block_exclusive_scan<CS_BT>(sum, 0u, SumOp(), s_tmp, &total);
  uint32_t g   = block_exclusive_scan<CS_BT>(cnt, 0u, SumOp(), s_tmp, &groups);
  const bool good = (int64_t)total == n && hf.cs_fail == 0;
  if (!good) {
    if (threadIdx.x == 0) hf.state = 4;  // an outlier the sample did not see: the LSD passes sort the column
    return;
  }
  for (uint32_t k = 0; k < per; ++k) {
    const uint32_t b = b0 + k;
    const uint32_t c = b < nb ? ghist[b] : 0u;
    if (c) {
      nz_start[g] = run;
      nz_val[g]   = b;
      ++g;
      run += c;
    }
  }
  if (threadIdx.x == 0) {
    nz_start[groups] = total;
    hf.cs_groups     = groups;
    for (int p = 0; p < npass; ++p) plan->pass_skip[p] = 1;  // as k_plan2 does for a column the hybrid path has sorted
    plan->num_active = -1;
    __threadfence();
    plan->hy.ok = 1;
  }
}

// every tile of CS_TILE output keys: the value runs that cover it (a bisection of nz_start by wave 0), then one value per element
template <typename KeyT, int KIND>
__global__ void __launch_bounds__(256) k_cs_fill(KeyT* __restrict__ out, int64_t n, KeyT desc_mask, const SortPlan* plan,
                                                 const uint32_t* __restrict__ nz_start, const uint32_t* __restrict__ nz_val)
{
  const FastPlan& hf = plan->hf;
  if (hf.state != 5 || !plan->hy.ok) return;
  __shared__ uint32_t s_start[CS_TILE + 2];
  __shared__ uint32_t s_g[2];
  const uint32_t groups = hf.cs_groups;
  const KeyT base       = (KeyT)hf.cs_base;
  const int64_t tiles   = div_up(n, (int64_t)CS_TILE);
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const uint32_t lo = (uint32_t)(tile * CS_TILE);
    const uint32_t hi = (uint32_t)((tile + 1) * CS_TILE < n ? (tile + 1) * CS_TILE : n);
    if (threadIdx.x < 2) {  // last group whose start is <= position (lo | hi - 1)
      const uint32_t pos = threadIdx.x == 0 ? lo : hi - 1;
      uint32_t a = 0, b = groups;
      while (b - a > 1) {
        const uint32_t mid = (a + b) >> 1;
        if (nz_start[mid] <= pos) a = mid; else b = mid;
      }
      s_g[threadIdx.x] = a;
    }
    __syncthreads();
    const uint32_t g0 = s_g[0], g1 = s_g[1];
    if (g0 == g1) {  // one value for the whole tile: the common case
      const KeyT v = from_sortable<KeyT, KIND>((KeyT)(base | (KeyT)nz_val[g0]), desc_mask);
      for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) __builtin_nontemporal_store(v, &out[i]);
    } else {  // g1 - g0 <= CS_TILE: every group holds at least one key
      const uint32_t ng = g1 - g0 + 1;
      for (uint32_t k = threadIdx.x; k < ng; k += 256) s_start[k] = nz_start[g0 + k];
      __syncthreads();
      for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
        uint32_t a = 0, b = ng;
        while (b - a > 1) {
          const uint32_t mid = (a + b) >> 1;
          if (s_start[mid] <= i) a = mid; else b = mid;
        }
        __builtin_nontemporal_store(from_sortable<KeyT, KIND>((KeyT)(base | (KeyT)nz_val[g0 + a]), desc_mask), &out[i]);
      }
    }
    __syncthreads();
  }
}

// the look-back granules must start at zero for k_msd_pass / k_radix_pass -- which do not run when the cursor path has
// sorted the column: then clearing half a gigabyte of status words is skipped too
__global__ void __launch_bounds__(256) k_hf_clear_status(const SortPlan* plan, uint4* __restrict__ status, size_t n16)
{
  if ((plan->hf.state == 3 || plan->hf.state == 5) && plan->hy.ok && !plan->hy.lsd_mode) return;  // (the X sort of the big cells runs the LSD passes too)
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) status[i] = uint4{0u, 0u, 0u, 0u};
}

// Big cells of the cursor path (HybridPlan::big), one workgroup: the cells that outgrew their slot, in cell (= key) order ->
// their first key in X (xoff), the list (cell, xoff) the copy-back searches, the level-0 buckets the rescue pass re-reads.
// Too many keys for X (half the level-0 buffer: the LSD passes need two work areas) or too many cells for the list: the
// whole-column LSD fallback runs, as it did for every overflow before round 4.
constexpr int BIG_LIST = 1 << 16;
__global__ void __launch_bounds__(BINS) k_big_plan(SortPlan* plan, unsigned long long xcap)
{
  HybridPlan& hy = plan->hy;
  if (plan->hf.state != 3 || !hy.ok || !hy.big) return;
  __shared__ uint32_t s_tmp[BINS / GX_WAVE + 1];
  const int b = threadIdx.x;
  uint32_t nbig, xtotal;
  const uint32_t k0 = block_exclusive_scan<BINS>(hy.bigcnt[b], 0u, SumOp(), s_tmp, &nbig);
  const uint32_t x0 = block_exclusive_scan<BINS>(hy.bigkeys[b], 0u, SumOp(), s_tmp, &xtotal);  // (n < 2^31: no overflow)
  if (nbig == 0 || nbig > (uint32_t)BIG_LIST || (unsigned long long)xtotal > xcap) {
    if (b == 0) {
      hy.ok  = 0;  // k_hist_all / k_plan / the LSD passes behind sort the column from scratch
      hy.big = 0;
    }
    return;
  }
  hy.biglist0[b]  = k0;
  hy.bigx0[b]     = x0;
  hy.bigbucket[b] = hy.bigcnt[b] ? 1u : 0u;
  if (b == 0) {
    hy.nbig     = nbig;
    hy.lsd_n    = xtotal;
    hy.lsd_mode = 1;
  }
}
// one wave per level-0 bucket that holds a big cell: the cells' first keys in X and the list entries
__global__ void __launch_bounds__(GX_WAVE) k_big_cells(const SortPlan* plan, const uint32_t* __restrict__ cellcount, uint32_t* __restrict__ xoff,
                                                       uint32_t* __restrict__ biglist)
{
  const HybridPlan& hy = plan->hy;
  const int b = blockIdx.x;
  if (!(hy.ok && hy.lsd_mode) || !hy.bigbucket[b]) return;
  constexpr int PER   = NB2MAX / GX_WAVE;
  const unsigned lane = lane_id();
  const int nb2       = 1 << hy.bits2;
  const uint32_t cap  = cell_cap(hy, (uint32_t)b);
  uint32_t c[PER], cnt = 0, keys = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int d2 = (int)lane * PER + k;
    c[k]         = d2 < nb2 ? cellcount[b * NB2MAX + d2] : 0u;
    if (c[k] > cap) {
      ++cnt;
      keys += c[k];
    }
  }
  uint32_t kk = hy.biglist0[b] + wave_inclusive_sum_dpp(cnt) - cnt;
  uint32_t x  = hy.bigx0[b] + wave_inclusive_sum_dpp(keys) - keys;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    if (c[k] > cap) {
      const uint32_t idx  = (uint32_t)(b * NB2MAX + (int)lane * PER + k);
      xoff[idx]           = x;
      biglist[2 * kk]     = idx;
      biglist[2 * kk + 1] = x;
      ++kk;
      x += c[k];
    }
  }
}

// behind the X sort: sorted X -> the output ranges of the big cells (cells are in key order, so is X)
template <typename KeyT>
__global__ void __launch_bounds__(256) k_big_distribute(const SortPlan* plan, const KeyT* __restrict__ xsorted, KeyT* __restrict__ out,
                                                        const uint32_t* __restrict__ biglist, const uint32_t* __restrict__ cellstart)
{
  const HybridPlan& hy = plan->hy;
  if (!(hy.ok && hy.lsd_mode)) return;
  const uint32_t n = (uint32_t)hy.lsd_n, nbig = hy.nbig;
  const uint32_t stride = gridDim.x * 256u;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += stride) {
    uint32_t lo = 0, hi = nbig;  // the last list entry whose first key is <= i
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (biglist[2 * mid + 1] <= i) lo = mid; else hi = mid;
    }
    out[cellstart[biglist[2 * lo]] + (i - biglist[2 * lo + 1])] = xsorted[i];
  }
}

// keys per thread of k_hf_scatter: 16 x 8 bytes = 64 KiB per tile; 32-bit keys take 24 (48 KiB: 32 per thread spill registers)
template <typename KeyT>
constexpr int hf_kpt()
{
  return sizeof(KeyT) == 8 ? 16 : 24;
}

// LVL 2 (round 4) = the RESCUE pass of the big cells: level 1 once more over the level-0 buckets that hold a big cell, but
// only the keys of big cells are written -- compacted into X at xoff[cell] (k_big_plan), cursors in `cellcur` + 2 * BINS * NB2MAX.
template <typename KeyT, int KIND, int LVL, int NBL>
__global__ void __launch_bounds__(BT, 4) k_hf_scatter(const KeyT* __restrict__ in, KeyT* __restrict__ out, KeyT desc_mask, SortPlan* plan,
                                                     uint32_t* __restrict__ cellcur, uint32_t cellcap, int64_t n, KeyT* __restrict__ fin = nullptr)
{
  // fin (level 1, splitter mode): the sort's final output -- the keys of an EQUALITY bucket are copied straight to their place
  constexpr int KPT = hf_kpt<KeyT>(), TILE = BT * KPT, NB = 1 << NBL, BPT = NB > BT ? NB / BT : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  KeyT* s_keys      = reinterpret_cast<KeyT*>(smem);                                        // [TILE]
  uint32_t* s_cnt   = reinterpret_cast<uint32_t*>(smem + (size_t)TILE * sizeof(KeyT));     // [NB] counts, then bin starts
  uint32_t* s_delta = s_cnt + NB;                                                           // [NB] output position - position in the tile
  uint32_t* s_limit = s_delta + NB;                                                         // [NB] end of the bin's slot
  uint32_t* s_scan  = s_limit + NB;                                                         // [16]
  uint32_t* s_misc  = s_scan + 16;                                                          // [4]
  unsigned long long* s_red = reinterpret_cast<unsigned long long*>(s_misc + 4);           // [2 * NW] (level 0)
  __shared__ uint2 s_wt[(LVL >= 1 && sizeof(KeyT) == 8) ? SP_NPIECE : 1];                  // splitter mode: the bucket's warp (sp_warp)
  HybridPlan& hy = plan->hy;
  FastPlan& hf   = plan->hf;
  if (hf.state != (LVL == 0 ? 1 : 3)) return;
  if (LVL == 0 && plan->sp.on) return;  // splitter mode: k_sp_level0 cuts the column
  if (LVL == 2 && !(hy.ok && hy.lsd_mode)) return;
  const bool split     = LVL >= 1 && sizeof(KeyT) == 8 && plan->sp.on;  // (block-uniform) cells = equal-width slices of the bucket's range
  const uint32_t* xoff = cellcur + 2 * BINS * NB2MAX;  // LVL 2: first key of a big cell in X ...
  uint32_t* rescur     = cellcur + 3 * BINS * NB2MAX;  // ... and the rescue cursors (hist2 | base2 | xoff | rescur)
  const unsigned tid = threadIdx.x;
  const int64_t v    = xcd_swizzle((int64_t)blockIdx.x, (int64_t)gridDim.x);  // XCD x works on a contiguous eighth of the tiles
  int64_t base;
  int nvalid;
  uint32_t seg;
  if (LVL == 0) {
    const int64_t per = (int64_t)gridDim.x / NRANGE;  // whole tiles per range (the last range takes the rest)
    seg               = (per > 0 && v / per < NRANGE - 1) ? (uint32_t)(v / per) : (uint32_t)(NRANGE - 1);
    base              = v * TILE;
    nvalid            = (int)(n - base < (int64_t)TILE ? n - base : (int64_t)TILE);
  } else if (!hf.ext) {
    // the column's own regions (bucket, input range): a table inside the plan, every address below is known at launch
    if (v >= (int64_t)hf.reg_tile0[BINS * NRANGE]) return;
    constexpr int QPT = BINS * NRANGE / BT;  // region table entries per thread
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
      const int q       = (int)tid * QPT + k;
      const uint32_t lo = hf.reg_tile0[q], hi = hf.reg_tile0[q + 1];
      if ((int64_t)lo <= v && v < (int64_t)hi) {
        s_misc[0] = (uint32_t)q;
        s_misc[1] = (uint32_t)(v - lo);
      }
    }
    __syncthreads();
    const uint32_t q  = s_misc[0];
    const uint32_t jt = s_misc[1];
    seg               = q / NRANGE;  // the level-0 bucket
    if (LVL == 2 && !hy.bigbucket[seg]) return;  // (block-uniform: nothing of this bucket is needed again)
    base              = (int64_t)hf.reg_start[q] + (int64_t)jt * TILE;
    const int64_t rem = (int64_t)hf.reg_count[q] - (int64_t)jt * TILE;
    nvalid            = (int)(rem < (int64_t)TILE ? rem : (int64_t)TILE);
  } else {
    // the sharded sort's receive area (gx_sortx_finish): regions (bucket, source rank, input range) in an external table
    const uint32_t nreg    = hf.nreg;
    const uint32_t* rtile0 = hf.x_tile0;
    if (v >= (int64_t)rtile0[nreg]) return;
    for (uint32_t q = tid; q < nreg; q += BT) {
      const uint32_t lo = rtile0[q], hi = rtile0[q + 1];
      if ((int64_t)lo <= v && v < (int64_t)hi) {
        s_misc[0] = q;
        s_misc[1] = (uint32_t)(v - lo);
      }
    }
    __syncthreads();
    const uint32_t q  = s_misc[0];
    const uint32_t jt = s_misc[1];
    seg               = hf.x_bucket[q];  // the level-0 bucket
    if (LVL == 2 && !hy.bigbucket[seg]) return;  // (block-uniform: nothing of this bucket is needed again)
    base              = (int64_t)hf.x_start[q] + (int64_t)jt * TILE;
    const int64_t rem = (int64_t)hf.x_count[q] - (int64_t)jt * TILE;
    nvalid            = (int)(rem < (int64_t)TILE ? rem : (int64_t)TILE);
  }
  const uint32_t dmask = LVL == 0 ? 0xFFu : ((1u << hy.bits2) - 1u);
  const Digit0 dig     = LVL == 0 ? digit0_of(hy, (int)(8 * sizeof(KeyT))) : Digit0{hy.shift2, 0, dmask, 0u};
  // the bucket's cell slots (HybridPlan::ccap / cbase), requested HERE: the two loads fly under the key loads and the ranking.
  // Asked for where they are used, next to the cursor atomics, they held the atomics back by a round trip per tile
  const uint32_t cap_s  = LVL >= 1 ? cell_cap(hy, seg) : 0u;
  const uint32_t base_s = LVL >= 1 ? cell_slot(hy, seg, 0u) : 0u;
  SpCell spc{0ull, 1ull, 0u, 63};
  if (split) {
    if (plan->sp.eq[seg]) {  // an equality bucket (block-uniform): every key is the same value -- no cells, no cell sort
      if (LVL == 2) return;
      if (tid == 0) s_misc[2] = atomicAdd(&cellcur[seg * NB2MAX], (uint32_t)nvalid);  // (k_plan2 checks the bucket's total against this)
      __syncthreads();
      const int64_t dst0 = (int64_t)hy.gbin0[seg] + (int64_t)s_misc[2];
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const int idx = j * BT + (int)tid;
        if (idx < nvalid) fin[dst0 + idx] = in[base + idx];
      }
      return;
    }
    spc = sp_cell_of(plan->sp, seg);
    if (tid < (unsigned)SP_NPIECE) s_wt[tid] = plan->sp.wt[seg][tid];  // (read behind the barrier that follows the clearing of s_cnt, on either path)
  }
  const uint32_t spnc = split ? plan->sp.nc[seg] : 1u;
  auto spdig          = [&](KeyT k) -> uint32_t {
    if constexpr (sizeof(KeyT) == 8) return sp_cell(sp_warp(sp_frac(spc, (unsigned long long)k), s_wt), spnc);
    else return dig(k);
  };
  if (split && sp_narrow(plan->sp, seg, hy.bits2)) {
    // a NARROW bucket (block-uniform): no more values in its range than it has cells, so every cell holds ONE value and the cell
    // sizes are all there is to know -- count (LDS histogram, one atomic per non-empty cell), nothing is moved; k_sp_fill writes
    // the bucket's output from the counts.  (Zipf-like columns: a third of the keys are values of 10^4 - 10^6 copies each.)
    if (LVL == 2) return;
    for (int b = tid; b < NB; b += BT) s_cnt[b] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int idx   = j * BT + (int)tid;
      const bool live = idx < nvalid;
      const KeyT k    = to_sortable<KeyT, KIND>(in[base + (live ? idx : 0)], desc_mask);
      (void)lds_rank(s_cnt, spdig(k), live);
    }
    __syncthreads();
    for (int b = tid; b < NB; b += BT) {
      const uint32_t c = s_cnt[b];
      if (c) atomicAdd(&cellcur[seg * NB2MAX + b], c);
    }
    return;
  }

  KeyT key[KPT];
  if (nvalid == TILE) {
#pragma unroll
    for (int j = 0; j < KPT; ++j) key[j] = in[base + j * BT + (int)tid];
  } else {
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int idx = j * BT + (int)tid;
      key[j]        = in[base + (idx < nvalid ? idx : 0)];  // padding repeats the tile's first key: neutral for the masks below
    }
  }
  for (int b = tid; b < NB; b += BT) s_cnt[b] = 0;
  if constexpr (LVL == 0 && KIND == K_FTOTAL) {  // float keys: every key is checked -- one NaN or -0.0 and the stable path sorts the column
    bool unclean = false;
#pragma unroll
    for (int j = 0; j < KPT; ++j) unclean |= float_unclean<KeyT>(key[j]);
    if (unclean) hf.fail = 1;
  }
  if (LVL == 0) {
    // exact varying-bit masks of the column, on the raw keys: the sortable form of an integer is the key XOR a constant,
    // so the set of bits that differ somewhere is the same
    // (float keys, K_FTOTAL: the flip depends on the sign, so the masks are taken on the sortable form itself)
    auto mform = [&](KeyT k) -> KeyT {
      if constexpr (KIND == K_FTOTAL) return to_sortable<KeyT, KIND>(k, KeyT(0));
      else return k;
    };
    KeyT vor = mform(key[0]), vnor = (KeyT)~mform(key[0]);
#pragma unroll
    for (int j = 1; j < KPT; ++j) {
      vor |= mform(key[j]);
      vnor |= (KeyT)~mform(key[j]);
    }
    const unsigned long long wo = wave_reduce((unsigned long long)vor, [](unsigned long long x, unsigned long long y) { return x | y; });
    const unsigned long long wn = wave_reduce((unsigned long long)vnor, [](unsigned long long x, unsigned long long y) { return x | y; });
    if (lane_id() == 0) {
      s_red[tid / GX_WAVE]      = wo;
      s_red[NW + tid / GX_WAVE] = wn;
    }
    if (KIND == K_SIGNED && hy.fold) {  // the sign fold is verified on the exact OR of (key ^ sign extension)
      KeyT vf = 0;
#pragma unroll
      for (int j = 0; j < KPT; ++j) vf |= (KeyT)(key[j] ^ (KeyT)(KeyT(0) - (KeyT)(key[j] >> (8 * sizeof(KeyT) - 1))));
      const unsigned long long wf = wave_reduce((unsigned long long)vf, [](unsigned long long x, unsigned long long y) { return x | y; });
      if (lane_id() == 0 && (wf & ~__hip_atomic_load(&hy.fold_x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) atomicOr(&hy.fold_x, wf);
    }
  }
  __syncthreads();
  if (LVL == 0 && tid == 0) {  // atomics only while this workgroup still has a bit to add
    unsigned long long o = 0, no = 0;
    for (int k = 0; k < NW; ++k) {
      o |= s_red[k];
      no |= s_red[NW + k];
    }
    if (o & ~__hip_atomic_load(&hy.or_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&hy.or_mask, o);
    if (no & ~__hip_atomic_load(&hy.nor_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&hy.nor_mask, no);
  }
  // 64-bit keys: digit and rank of key j in one word; 32-bit keys (32 per thread): two 16-bit ranks per word, the digit is
  // recomputed from the key at the scatter (registers: 32 keys + 32 words would spill)
  constexpr bool RANK16 = sizeof(KeyT) == 4;
  uint32_t packed[RANK16 ? KPT / 2 : KPT];
  if constexpr (RANK16) {
#pragma unroll
    for (int j = 0; j < KPT / 2; ++j) packed[j] = 0;
  }
  auto rank_all = [&](auto&& digf) {
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const bool live  = j * BT + (int)tid < nvalid;
      const KeyT k     = to_sortable<KeyT, KIND>(key[j], desc_mask);
      const uint32_t d = digf(k);
      const uint32_t r = lds_rank(s_cnt, d, live);
      if constexpr (RANK16) packed[j >> 1] |= r << (16 * (j & 1));
      else packed[j] = (d << 16) | r;
    }
  };
  if (split) rank_all(spdig); else rank_all(dig);  // (one branch per tile, not per key: the bit-digit path is what it was)
  __syncthreads();
  // ---- one returning atomic per non-empty bin reserves the tile's run; the scan runs while it is in flight
  uint32_t c[BPT], g[BPT], sbase[BPT], scap[BPT];
  bool skip[BPT];
  uint32_t csum = 0;
#pragma unroll
  for (int k = 0; k < BPT; ++k) {
    const uint32_t bin = tid * BPT + k;
    c[k] = g[k] = sbase[k] = scap[k] = 0;
    skip[k] = false;
    if (bin < (uint32_t)NB) {
      c[k] = s_cnt[bin];
      if (LVL == 0) {
        sbase[k] = hf.slot0[seg][bin];
        scap[k]  = hf.cap0[seg][bin];
        if (c[k]) g[k] = atomicAdd(&hf.cur0[seg][bin], c[k]);
      } else if (LVL == 1 && bin <= dmask) {
        sbase[k] = base_s + bin * cap_s;
        scap[k]  = cap_s;
        if (c[k]) g[k] = atomicAdd(&cellcur[seg * NB2MAX + bin], c[k]);
      } else if (LVL == 2 && bin <= dmask) {
        const uint32_t total = cellcur[seg * NB2MAX + bin];  // the cell's true size (level 1 counted every key)
        if (total > cap_s) {
          sbase[k] = xoff[seg * NB2MAX + bin];
          scap[k]  = total;
          if (c[k]) g[k] = atomicAdd(&rescur[seg * NB2MAX + bin], c[k]);
        } else {
          skip[k] = true;  // not a big cell: its keys stay where level 1 put them (limit 0 below: nothing is written)
        }
      }
      csum += c[k];
    }
  }
  uint32_t st = block_exclusive_scan<BT>(csum, 0u, SumOp(), s_scan, (uint32_t*)nullptr);
  bool spilled = false;
#pragma unroll
  for (int k = 0; k < BPT; ++k) {
    const uint32_t bin = tid * BPT + k;
    if (bin < (uint32_t)NB) {
      if (c[k] && !skip[k] && g[k] + c[k] > scap[k]) spilled = true;  // the surplus is dropped at the write-out
      s_cnt[bin]   = st;
      s_delta[bin] = sbase[k] + g[k] - st;
      s_limit[bin] = sbase[k] + scap[k];
      st += c[k];
    }
  }
  // The flag is raised only while it is still down (a load, no read-modify-write): a column whose cells overflow by the million
  // (bell-shaped or Zipf-like values) used to send an atomic per (tile, bin) to this ONE word -- 2.4e7 to 1.2e8 of them, 205 and
  // 707 ms (profiles/r4_run15_sort_zipf_kernel_stats.txt, r4_run18_sort_robustness.txt).  Nothing is added for a tile without a spill.
  if (spilled) {
    if (LVL == 0) hf.fail = 1;
    else if (LVL == 1) {
      if (!__hip_atomic_load(&hy.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicExch(&hy.overflow, 1);
    } else atomicExch(&hy.bad, 2);  // (cannot happen: the rescue writes exactly the keys level 1 counted)
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    if (j * BT + (int)tid < nvalid) {
      if constexpr (RANK16) {
        const uint32_t d = dig(to_sortable<KeyT, KIND>(key[j], desc_mask));
        s_keys[s_cnt[d] + ((packed[j >> 1] >> (16 * (j & 1))) & 0xFFFFu)] = key[j];
      } else {
        s_keys[s_cnt[packed[j] >> 16] + (packed[j] & 0xFFFFu)] = key[j];
      }
    }
  }
  __syncthreads();
  auto write_all = [&](auto&& digf) {
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int i = j * BT + (int)tid;
      if (i < nvalid) {
        const KeyT k       = s_keys[i];
        const uint32_t d   = digf(to_sortable<KeyT, KIND>(k, desc_mask));
        const uint32_t dst = s_delta[d] + (uint32_t)i;
        if (dst < s_limit[d]) out[dst] = k;
      }
    }
  };
  if (split) write_all(spdig); else write_all(dig);
}

// ---- level 0 in SPLITTER mode (64-bit keys): k_hf_scatter<.., 0, 8> with bucket = sp_bucket(key) in place of the bit digit.
// The search is the expensive part (a LUT read + a short scan of the sorted table per key, both in LDS: +2.5 ms per 1e9 keys
// against a plain histogram, profiles/r4_run32_xp_splitter_level0.txt), so it runs ONCE per key: the bucket travels with the key
// through the LDS reorder as one byte.  14 keys per thread (56 KiB of keys + 7 KiB of bucket bytes + 6 KiB of tables): two
// workgroups per CU, as the bit-digit kernel.
constexpr int SP_KPT  = 14;
// (13 keys per thread -- 39.3 KiB of LDS, four workgroups per CU instead of three -- measured in run 15: 5.44 against 5.36 ms, no gain)
constexpr int SP_TILE = BT * SP_KPT;
constexpr size_t sp_level0_lds() { return (size_t)SP_TILE * 8 + (size_t)(3 * BINS + 16 + 4) * 4 + (size_t)2 * NW * 8 + (size_t)BINS * 8 + (size_t)SP_NLUT * 2 + (size_t)SP_TILE; }
// F2: the LUT form this instantiation searches with (1: form 2, SpLutF; 0: forms 0 / 1).  Both are launched, the one the plan did not
// pick returns at once (~0.05 ms): with the form tested per key the unrolled ranking loop of EVERY split-mode column slowed by 0.15 ms
// (run 14), and with two copies of the loop inside one kernel the compiler stopped interleaving the fourteen searches (70 instead
// of 84 registers, level 0 4.9 -> 5.4 ms, run 15).
template <int KIND, int F2>
__global__ void __launch_bounds__(BT, 4) k_sp_level0(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t desc_mask, SortPlan* plan, int64_t n)
{
  constexpr int KPT = SP_KPT, TILE = SP_TILE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* s_keys  = reinterpret_cast<uint64_t*>(smem);                                       // [TILE]
  uint32_t* s_cnt   = reinterpret_cast<uint32_t*>(smem + (size_t)TILE * 8);                   // [BINS] counts, then bin starts
  uint32_t* s_delta = s_cnt + BINS;                                                           // [BINS] output position - position in the tile
  uint32_t* s_limit = s_delta + BINS;                                                         // [BINS] end of the bin's slot
  uint32_t* s_scan  = s_limit + BINS;                                                         // [16]
  uint32_t* s_misc  = s_scan + 16;                                                            // [4]
  unsigned long long* s_red = reinterpret_cast<unsigned long long*>(s_misc + 4);             // [2 * NW]
  unsigned long long* s_tab = s_red + 2 * NW;                                                 // [BINS]
  uint16_t* s_lut   = reinterpret_cast<uint16_t*>(s_tab + BINS);                              // [SP_NLUT]
  uint8_t* s_bid    = reinterpret_cast<uint8_t*>(s_lut + SP_NLUT);                            // [TILE] bucket of the key at this tile position
  HybridPlan& hy      = plan->hy;
  FastPlan& hf        = plan->hf;
  const SplitPlan& sp = plan->sp;
  if (hf.state != 1 || !sp.on || !hf.slots_ready) return;
  if ((sp.lut_log == 2u) != (F2 == 1)) return;
  const unsigned tid = threadIdx.x;
  const int64_t v    = xcd_swizzle((int64_t)blockIdx.x, (int64_t)gridDim.x);
  const int64_t per  = (int64_t)gridDim.x / NRANGE;  // whole tiles per range (the last range takes the rest)
  const uint32_t seg = (per > 0 && v / per < NRANGE - 1) ? (uint32_t)(v / per) : (uint32_t)(NRANGE - 1);
  const int64_t base = v * TILE;
  const int nvalid   = (int)(n - base < (int64_t)TILE ? n - base : (int64_t)TILE);
  uint64_t key[KPT];
  if (nvalid == TILE) {
#pragma unroll
    for (int j = 0; j < KPT; ++j) key[j] = in[base + j * BT + (int)tid];
  } else {
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int idx = j * BT + (int)tid;
      key[j]        = in[base + (idx < nvalid ? idx : 0)];  // padding repeats the tile's first key: neutral for the masks below
    }
  }
  for (int b = tid; b < BINS; b += BT) {
    s_cnt[b] = 0;
    s_tab[b] = sp.tab[b];
  }
  for (int i = tid; i < SP_NLUT; i += BT) s_lut[i] = sp.lut[i];
  if constexpr (KIND == K_FTOTAL) {  // float keys: one NaN or -0.0 and the stable path sorts the column (k_hf_plan stage 2 sees hf.fail)
    bool unclean = false;
#pragma unroll
    for (int j = 0; j < KPT; ++j) unclean |= float_unclean<uint64_t>(key[j]);
    if (unclean) hf.fail = 1;
  }
  {  // exact varying-bit masks of the column (the LSD fallback and the stable passes of crowded cells skip constant bytes by them)
    auto mform = [&](uint64_t k) -> uint64_t {  // (float keys: the flip depends on the sign -- masks of the sortable form itself)
      if constexpr (KIND == K_FTOTAL) return to_sortable<uint64_t, KIND>(k, 0ull);
      else return k;
    };
    uint64_t vor = mform(key[0]), vnor = ~mform(key[0]);
#pragma unroll
    for (int j = 1; j < KPT; ++j) {
      vor |= mform(key[j]);
      vnor |= ~mform(key[j]);
    }
    const unsigned long long wo = wave_reduce((unsigned long long)vor, [](unsigned long long x, unsigned long long y) { return x | y; });
    const unsigned long long wn = wave_reduce((unsigned long long)vnor, [](unsigned long long x, unsigned long long y) { return x | y; });
    if (lane_id() == 0) {
      s_red[tid / GX_WAVE]      = wo;
      s_red[NW + tid / GX_WAVE] = wn;
    }
  }
  __syncthreads();
  if (tid == 0) {
    unsigned long long o = 0, no = 0;
    for (int k = 0; k < NW; ++k) {
      o |= s_red[k];
      no |= s_red[NW + k];
    }
    if (o & ~__hip_atomic_load(&hy.or_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&hy.or_mask, o);
    if (no & ~__hip_atomic_load(&hy.nor_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&hy.nor_mask, no);
  }
  const uint32_t nsp = sp.nsp, lut_log = sp.lut_log, lshift = sp.lshift;
  const unsigned long long kmin = sp.kmin;
  const SpLutF lf               = sp_lutf_of(sp);
  uint32_t packed[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) packed[j] = 0;
  const uint32_t steps = sp.lut_steps;
  if (steps <= 4u) {  // (block-uniform; the common case: the fullest LUT cell of a bell-shaped / lognormal / float column holds 1 - 2 splitters)
    // the search WITHOUT branches: the LUT word names the first candidate, then `steps` times "one further if that splitter is <= key"
    // (past the key's own LUT cell every splitter is larger: extra steps change nothing).  The per-key scan loop this replaces
    // diverged, and its dependent LDS reads could not be interleaved over the thread's fourteen keys -- run 16: eight times the
    // cycles waiting on LDS of the bit-digit kernel, 50 % more wave-cycles.
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const uint64_t k = to_sortable<uint64_t, KIND>(key[j], desc_mask);
      packed[j]        = (uint32_t)s_lut[sp_lut_cell_k<KIND, F2>(k, kmin, lut_log, lshift, lf)] & 0x1FFu;
    }
    for (uint32_t st = 0; st < steps; ++st) {
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const uint64_t k  = to_sortable<uint64_t, KIND>(key[j], desc_mask);
        const uint32_t bb = packed[j];
        packed[j]         = bb + ((bb < nsp && s_tab[bb] <= k) ? 1u : 0u);
      }
    }
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const bool live  = j * BT + (int)tid < nvalid;
      const uint32_t d = packed[j];
      const uint32_t r = lds_rank(s_cnt, d, live);
      packed[j]        = (d << 16) | r;
    }
  } else {
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const bool live  = j * BT + (int)tid < nvalid;
      const uint64_t k = to_sortable<uint64_t, KIND>(key[j], desc_mask);
      const uint32_t d = sp_bucket<KIND, F2>(s_tab, s_lut, k, nsp, kmin, lut_log, lshift, lf);
      const uint32_t r = lds_rank(s_cnt, d, live);
      packed[j]        = (d << 16) | r;
    }
  }
  __syncthreads();
  // one returning atomic per non-empty bin reserves the tile's run; the scan runs while it is in flight
  uint32_t c = 0, g = 0, sbase = 0, scap = 0;
  if (tid < (unsigned)BINS) {
    c     = s_cnt[tid];
    sbase = hf.slot0[seg][tid];
    scap  = hf.cap0[seg][tid];
    if (c) g = atomicAdd(&hf.cur0[seg][tid], c);
  }
  uint32_t st = block_exclusive_scan<BT>(c, 0u, SumOp(), s_scan, (uint32_t*)nullptr);
  if (tid < (unsigned)BINS) {
    if (c && g + c > scap) hf.fail = 1;  // the surplus is dropped at the write-out; the verdict sends the column elsewhere
    s_cnt[tid]   = st;
    s_delta[tid] = sbase + g - st;
    s_limit[tid] = sbase + scap;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    if (j * BT + (int)tid < nvalid) {
      const uint32_t d   = packed[j] >> 16;
      const uint32_t pos = s_cnt[d] + (packed[j] & 0xFFFFu);
      s_keys[pos]        = key[j];
      s_bid[pos]         = (uint8_t)d;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int i = j * BT + (int)tid;
    if (i < nvalid) {
      const uint32_t d   = s_bid[i];
      const uint32_t dst = s_delta[d] + (uint32_t)i;
      if (dst < s_limit[d]) out[dst] = s_keys[i];
    }
  }
}

static thread_local int g_algorithm = 0;
static thread_local int g_order_mode = 0;

// optional per-launch timing with HIP events on the caller's stream (bench.py's roofline leg)
struct Profile {
  bool enabled = false;
  bool created = false;
  int level    = 1;  // 1: every event; 2: only the two around the first partition level (hev[0], hev[1]) -- an event between two kernels costs
                     // the stream ~15 us, and a sort records 23 of them (0.35 ms of an 11-ms sort): the timed steps of bench.py take level 2
  int npass    = 0;
  hipEvent_t ev[2 * MAX_PASSES + 2];
  hipEvent_t hev[5];  // hybrid: before msd0, hist2, msd1, local sort, after
  bool hybrid_marked = false;
};
// Event SLOTS (gx_sort_profile_slot): a caller that times K calls back to back gives each call its own events and reads them all
// AFTER the K calls -- reading a call's events right behind it waits for the call and puts a host round trip between two timed steps
// (bench.py until round 6: +0.3 ms per 11-ms step against the same step through the C++ API, which never profiled).
constexpr int PROF_SLOTS = 64;
static thread_local Profile g_profs[PROF_SLOTS];  // per calling thread, like every knob below: a thread that profiles or forces a path affects its own calls only
static thread_local int g_prof_slot = 0;
#define g_prof g_profs[g_prof_slot]
static inline void prof_mark(int idx, hipStream_t s)
{
  if (g_prof.enabled && g_prof.level == 1) (void)hipEventRecord(g_prof.ev[idx], s);
}
static inline void prof_mark_h(int idx, hipStream_t s)
{
  if (g_prof.enabled && (g_prof.level == 1 || idx <= 1)) (void)hipEventRecord(g_prof.hev[idx], s);
}
static thread_local int g_hybrid = 1;  // 0 disables the hybrid MSD path (A/B knob)
static thread_local int g_lbw    = 16;  // predecessors per look-back round of the keys-only hybrid partition passes (knob: 4, 8, 16;
                           // 16 measured 2-3 % ahead of 4: profiles/r2_run21_bench_sort_lookback_window.jsonl)
static thread_local int g_msd_kpt = 16;  // keys per thread of the partition passes (8 and 12 measured slower: 5.7 / 4.7 vs 4.0 ms)

template <typename KeyT>
constexpr int kpt_for(bool has_val)
{
  return (sizeof(KeyT) == 8 && has_val) ? 10 : 16;
}

template <typename KeyT, bool HAS_VAL, int KPT>
constexpr size_t pass_lds_bytes()
{
  return (size_t)BT * KPT * sizeof(KeyT) + (HAS_VAL ? (size_t)BT * KPT * 4 : 0) +
         (size_t)(NW * BINS + BINS + 16 + 4) * 4;
}

// hybrid configuration: a pure function of (n, key kind, payload, knobs), so the scratch query and the run agree
struct HybridCfg {
  bool on;
  int cl2;    // log2 of the local-sort cell capacity (13 or 14)
  int bits2;  // level-1 bits (1..9)
  int kpt;    // keys per thread of the partition passes
  int cl2_alt;  // 14 when the device may switch a 13-bit plan to 16384-key cells (pairs: k_hy_plan's cell_alt), else 0
};
static thread_local int g_cell = 0;  // A/B knob: 0 = auto, 8192 / 16384 = force the local-sort cell capacity
// workgroups of k_local_sort: they walk the cells (or k_local_place's todo list) with a stride, see the kernel
static inline unsigned local_sort_grid(int cells) { return (unsigned)(cells < 4096 ? cells : 4096); }
static thread_local int g_place_grid = 0;  // A/B knob: workgroups of k_local_place; 0 = one per cell
static inline unsigned local_place_grid(int cells) { return (unsigned)((g_place_grid > 0 && g_place_grid < cells) ? g_place_grid : cells); }
template <typename KeyT, int KIND, bool HAS_VAL>
static HybridCfg hybrid_cfg(int64_t n, bool iota_payload, int algo)
{
  HybridCfg c{false, 14, 8, HAS_VAL ? 10 : g_msd_kpt, 0};
  if (sizeof(KeyT) != 8 || (HAS_VAL && !iota_payload) || algo != 0 || !g_hybrid || n < (1ll << 22)) return c;
  // 8192-key cells (two local-sort workgroups per CU) + a 9-bit level 1 for INTEGER keys while 2^17 cells suffice -- keys only,
  // and since round 3 pairs too (sorted_order: the packed (key bits, position) words need 13 position bits instead of 14);
  // float keys keep the 16384-key cells of round 1
  const bool small_ok = KIND != K_FLOAT;
  c.cl2 = (small_ok && g_cell != 16384 && (double)n / (double)(1 << 17) <= 0.955 * 8192.0) ? 13 : 14;
  if (g_cell == 8192 && small_ok) c.cl2 = 13;
  const double cell = (double)(1 << c.cl2);
  const int maxb2   = small_ok ? 9 : 8;  // the 9-bit level-1 pass exists for integer keys-only sorts
  int B = 9;
  while (B < 8 + maxb2 && (double)n / (double)(1ull << B) > 0.955 * cell) ++B;
  if ((double)n / (double)(1ull << B) > 0.97 * cell) return c;  // cells would overflow: LSD passes
  c.bits2 = B - 8;
  c.on    = true;
  if (HAS_VAL && c.cl2 == 13 && g_cell == 0) c.cl2_alt = 14;
  return c;
}

// cursor path (integer keys, keys only): configuration, again a pure function of (n, kind, knobs)
struct FastCfg {
  bool on;
  int bits2;         // level-1 bits (1..10), 8192-key cells
  int bits2_max;     // ... or one more, decided on the device from the exact level-0 histogram (k_hf_plan stage 2)
  int stride;        // sample: every stride-th 64-key chunk
  size_t slot_rows;  // keys the padded level-0 output holds
};
static thread_local int g_soft_fault      = 0;     // 1: an abandoned look-back wait is REPORTED through the status word (the caller reads it); 0: it traps
static thread_local int g_spin_ms         = 0;     // look-back wait limit in ms (0: SPIN_SECONDS); tests shorten it
static thread_local long long g_inject_tile = -1;  // TEST HOOK: the tile of every look-back pass that never publishes (-1: none)
static thread_local int g_cursor          = 1;     // 0 disables the cursor path (A/B knob)
static thread_local int g_counting        = 1;     // 0 disables the counting sort of narrow key ranges (A/B knob: the LSD passes run)
static thread_local int g_float_cursor    = 1;     // 0: float64 keys stay on the look-back path (A/B knob)
static thread_local int g_split           = 1;     // 0 disables the splitter mode of the cursor path (A/B knob: uneven columns are declined as before round 5)
static thread_local int g_exp             = 0;     // ablation bits of k_local_sort (measurement only: the result is NOT sorted under most of them)
static thread_local float g_cursor_margin = 8.0f;  // standard deviations of slack per level-0 slot (tests: < 0 forces the fallback)
template <typename KeyT, int KIND, bool HAS_VAL>
static FastCfg fast_cfg(int64_t n, int algo, bool hybrid_on)
{
  FastCfg f{false, 9, 9, 32, 0};
  // 64-bit keys: wherever the hybrid path applies; 32-bit integer keys (round 3): the same two partition levels, the cells sorted
  // by k_local_place on 32-bit words, the LSD passes as the only fallback
  // (round 5) float64 keys only: the cursor path runs on the total-order flip (K_FTOTAL) and verifies on every key that the column holds
  // no NaN and no -0.0 -- the values on which an unordered sort and the reference's stable one could differ; else the look-back path
  if ((sizeof(KeyT) != 8 && sizeof(KeyT) != 4) || HAS_VAL || (KIND == K_FLOAT && (sizeof(KeyT) != 8 || !g_float_cursor)) || algo != 0 || !g_cursor || n < (1ll << 25)) return f;
  if (sizeof(KeyT) == 8 ? !hybrid_on : !g_hybrid) return f;
  int B = 9;
  while (B < 18 && (double)n / (double)(1ull << B) > 0.955 * 8192.0) ++B;
  if ((double)n / (double)(1ull << B) > 0.97 * 8192.0) return f;
  f.bits2  = B - 8;
  f.bits2_max = f.bits2 < 10 ? f.bits2 + 1 : 10;
  f.stride = n >= (1ll << 27) ? 32 : 8;
  // sum of the slot capacities k_hf_plan hands out (Cauchy-Schwarz over the 2048 slots; the kernel checks it again)
  const double m   = g_cursor_margin > 0 ? g_cursor_margin : 0.0;
  const double dev = m * 1.25 * f.stride * __builtin_sqrt((double)(NRANGE * BINS) * ((double)n / f.stride + 2.0 * NRANGE * BINS));
  f.slot_rows      = (size_t)((double)n * 1.002 + dev) + (size_t)(NRANGE * BINS) * (size_t)(2 * f.stride * HF_CHUNK + 64 + 16) + 65536;
  f.on             = true;
  return f;
}

template <typename KeyT, int KIND, bool HAS_VAL>
int sort_impl(const void* keys_in, void* keys_out, const int32_t* vals_in, int32_t* vals_out, int64_t n,
              int descending, bool radix_nan_rule, void* tmp, size_t* tmp_bytes, hipStream_t stream)
{
  constexpr int KPT   = kpt_for<KeyT>(HAS_VAL);
  constexpr int TILE  = BT * KPT;
  constexpr int NPASS = sizeof(KeyT);
  if (n < 0 || tmp_bytes == nullptr) return GX_EINVAL;
  if (n > 0x7FFFFFFFll) return GX_EINVAL;  // offsets are 32-bit: cudf::size_type rows (types.hpp:76)
  const int64_t ntiles = n > 0 ? div_up(n, TILE) : 0;
  const int algo       = g_algorithm;

  Carver c(tmp);
  SortPlan* plan             = c.take<SortPlan>(1);
  unsigned long long* status = nullptr;
  uint32_t* tile_hist        = nullptr;
  uint32_t* partials         = nullptr;
  // hybrid MSD path: 64-bit keys; pairs only with the iota payload (sorted_order), whose value is the
  // tie-break the packed local sort relies on
  HybridCfg hc     = hybrid_cfg<KeyT, KIND, HAS_VAL>(n, vals_in == nullptr, algo);
  const FastCfg fc = fast_cfg<KeyT, KIND, HAS_VAL>(n, algo, hc.on);
  // cell buffer of the cursor path: per-bucket slots (HybridPlan::ccap) -- n keys, 1/16 for the neighbour smoothing, the slack per cell
  const size_t fc_cells = fc.on ? (size_t)n + (size_t)n / 16 + ((size_t)BINS << fc.bits2_max) * cell_slack(1 << 13) + 65536 : 0;
  // the look-back path is the cursor path's first fallback (a sample that was not representative).  Up to 2^17 cells of 8192 keys
  // (n <= 1.03e9) its FIXED cell slots are about the size of that buffer; above, it would switch to 16384-key slots (17.2 GB for
  // 1.25e9 keys of 10 GB): there the LSD passes are the fallback, as for 32-bit keys, and the scratch stays proportional to n
  if (fc.on && hc.on && hc.cl2 == 14 && (((size_t)BINS << hc.bits2) << 14) > fc_cells + (size_t)n / 4) hc.on = false;
  const bool try_hybrid = hc.on;
  const int hyb_kpt     = hc.kpt;
  const int nb1         = hc.bits2 > 8 ? NB9 : BINS;  // bins (and look-back granules per tile) of the level-1 pass
  uint32_t* base1 = c.take<uint32_t>((size_t)NRANGE * NB2MAX);
  const bool cells = try_hybrid || fc.on;  // (the cursor path of 32-bit keys runs without the look-back hybrid)
  // cell sizes | cell output positions | (cursor path: big cells) first key in X | rescue cursors
  uint32_t* hist2 = cells ? c.take<uint32_t>((size_t)4 * BINS * NB2MAX) : nullptr;
  uint32_t* base2 = cells ? hist2 + BINS * NB2MAX : nullptr;
  uint32_t* xoff  = cells ? hist2 + 2 * BINS * NB2MAX : nullptr;
  uint32_t* todo  = cells ? c.take<uint32_t>((size_t)BINS * NB2MAX) : nullptr;  // cells k_local_place leaves to k_local_sort
  uint32_t* biglist = fc.on ? c.take<uint32_t>((size_t)2 * BIG_LIST) : nullptr;  // big cells: (cell, first key in X)
  const int64_t msd_tile     = (int64_t)BT * hyb_kpt;  // tile of the hybrid partition passes
  const int64_t msd_ntiles   = n > 0 ? div_up(n, msd_tile) : 0;
  const int64_t status_tiles = (msd_ntiles > ntiles ? msd_ntiles : ntiles) + BINS + 2 * NRANGE;  // segment tails add at most one tile each
  const size_t status_words  = (size_t)status_tiles * (try_hybrid ? nb1 : BINS);
  if (algo != 1) {
    status = c.take<unsigned long long>(status_words);
  } else {
    tile_hist = c.take<uint32_t>((size_t)ntiles * BINS);
    partials  = c.take<uint32_t>(scan::partials_count(ntiles * BINS));
  }
  // the level-1 pass writes every cell into its own slot (no joint histogram pass).  Round 4: slots are sized per bucket from the
  // exact level-0 histogram (HybridPlan::ccap: the bucket's mean cell + 6 sigma + 64 keys), so the cell buffer holds n keys + that
  // slack per cell instead of (256 << bits2) slots of the full cell capacity (8.6 / 17.2 GB -> 9.3 GB for the 1e9-row sort)
  // (cursor path only: the look-back path -- floats, pairs, the cursor path's fallback -- keeps (256 << bits2) slots of the full capacity)
  size_t padded = try_hybrid ? ((size_t)BINS << hc.bits2) << (hc.cl2_alt ? hc.cl2_alt : hc.cl2) : 0;
  if (fc_cells > padded) padded = fc_cells;
  const size_t nb_buf = padded > (size_t)n ? padded : (size_t)n;
  KeyT* kb_scratch = c.take<KeyT>(nb_buf);
  KeyT* slot0_buf  = fc.on ? c.take<KeyT>(fc.slot_rows) : nullptr;  // cursor path: padded level-0 output
  KeyT* ka_scratch = keys_out ? nullptr : c.take<KeyT>((size_t)n);
  uint32_t* vb     = HAS_VAL ? c.take<uint32_t>(nb_buf) : nullptr;
  if (tmp == nullptr) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  if (n > 0 && (keys_in == nullptr || (HAS_VAL && vals_out == nullptr))) return GX_EINVAL;
  if (n > 0 && keys_in == keys_out) return GX_EINVAL;

  GX_HIP_TRY(hipMemsetAsync(plan, 0, sizeof(SortPlan), stream));
  if (n == 0) return 0;
  if (algo != 1 && !fc.on) GX_HIP_TRY(hipMemsetAsync(status, 0, status_words * sizeof(unsigned long long), stream));
  if (cells) GX_HIP_TRY(hipMemsetAsync(hist2, 0, (size_t)4 * BINS * NB2MAX * sizeof(uint32_t), stream));
  const int64_t range_rows = try_hybrid ? div_up(msd_ntiles, NRANGE) * msd_tile : div_up(ntiles, NRANGE) * TILE;

  const KeyT desc_mask = descending ? KeyT(~KeyT(0)) : KeyT(0);
  g_prof.npass = NPASS;
  int64_t hblocks = div_up(n, (int64_t)BT * 8);
  if (hblocks > 2048) hblocks = 2048;
  hblocks = div_up(hblocks, NRANGE) * NRANGE;  // block b serves input range b % NRANGE
  prof_mark(0, stream);
  g_prof.hybrid_marked = false;
  bool cursor_marked = false;
  // CK: the key kind the cursor path's kernels are instantiated for -- float64 columns travel on the total-order flip (K_FTOTAL)
  constexpr int CK = (KIND == K_FLOAT) ? (int)K_FTOTAL : KIND;
  if constexpr ((sizeof(KeyT) == 8 || sizeof(KeyT) == 4) && !HAS_VAL && (KIND != K_FLOAT || sizeof(KeyT) == 8)) {
    if (fc.on) {
      // cursor path: speculative plan from a sample, verified by level 0; on a miss everything below is a no-op and the
      // look-back path further down sorts the column
      constexpr int FT = BT * hf_kpt<KeyT>();  // k_hf_scatter's tile
      constexpr int MIN_SHIFT2 = 8;  // key bits that must be left below level 1 (k_local_sort's sub-bucket split takes 7 + 1)
      constexpr int WORD_BYTES = (int)sizeof(typename PlaceWord<KeyT, CK, HAS_VAL>::type);
      const int64_t ftiles  = div_up(n, (int64_t)FT);
      const int64_t frange  = (ftiles / NRANGE) * FT;  // rows per input range (whole tiles; the last range takes the rest)
      auto lds_hf = [&](int nb) { return (size_t)FT * sizeof(KeyT) + (size_t)(3 * nb + 16 + 4) * 4 + (size_t)2 * NW * 8; };
      typedef void (*HfK)(const KeyT*, KeyT*, KeyT, SortPlan*, uint32_t*, uint32_t, int64_t, KeyT*);
      HfK kf0 = k_hf_scatter<KeyT, CK, 0, 8>;
      // (the level-1 kernel and the cell grids are sized for bits2_max: the device may take the extra bit)
      HfK kf1 = fc.bits2_max <= 8 ? (HfK)k_hf_scatter<KeyT, CK, 1, 8> : (fc.bits2_max == 9 ? (HfK)k_hf_scatter<KeyT, CK, 1, 9> : (HfK)k_hf_scatter<KeyT, CK, 1, 10>);
      HfK kf2 = fc.bits2_max <= 8 ? (HfK)k_hf_scatter<KeyT, CK, 2, 8> : (fc.bits2_max == 9 ? (HfK)k_hf_scatter<KeyT, CK, 2, 9> : (HfK)k_hf_scatter<KeyT, CK, 2, 10>);
      const int nbf = fc.bits2_max <= 8 ? 256 : (1 << fc.bits2_max);
      static std::atomic<bool> fattr_set{false};
      if (!fattr_set) {
        GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hf_scatter<KeyT, CK, 2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hf(256)));
        GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hf_scatter<KeyT, CK, 2, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hf(512)));
        GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hf_scatter<KeyT, CK, 2, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hf(1024)));
        GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hf_scatter<KeyT, CK, 0, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hf(256)));
        GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hf_scatter<KeyT, CK, 1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hf(256)));
        GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hf_scatter<KeyT, CK, 1, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hf(512)));
        GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hf_scatter<KeyT, CK, 1, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hf(1024)));
        GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_local_sort<uint64_t, CK, HAS_VAL, 13, KeyT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(((size_t)8 << 13) + (size_t)(((1 << 13) / 16 / GX_WAVE) * BINS + 32 + 2 * BINS) * 4)));
        GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_local_place<KeyT, CK, HAS_VAL, 13>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)place_lds_bytes(13, WORD_BYTES)));
        fattr_set = true;
      }
      const int64_t step = (int64_t)fc.stride * HF_CHUNK;
      int64_t sblocks    = div_up(div_up(n, step), (int64_t)4 * 4);
      if (sblocks > 2048) sblocks = 2048;
      const KeyT* kin = static_cast<const KeyT*>(keys_in);
      KeyT* bufA      = keys_out ? static_cast<KeyT*>(keys_out) : ka_scratch;
      hipLaunchKernelGGL((k_hf_sample<KeyT, CK, false>), dim3((unsigned)sblocks), dim3(256), 0, stream, kin, n, desc_mask, plan, fc.stride, frange);
      hipLaunchKernelGGL(k_hf_plan, dim3(1), dim3(BINS), 0, stream, plan, 0, (int)(8 * sizeof(KeyT)), n, fc.bits2, 1 << 13, fc.stride, frange, FT,
                         (unsigned long long)fc.slot_rows, g_cursor_margin, MIN_SHIFT2, fc.bits2_max, 0ull, CK == K_SIGNED ? 1 : 0, (KIND == K_FLOAT) ? 0 : g_counting);
      {
        // counting sort of a column whose varying bits are its low <= 15 (state 5; no-ops otherwise): histogram in LDS, scan, fill
        // (counters | group starts | group values live in the cell tables, unused on this branch and zeroed above)
        static std::atomic<bool> cattr_set{false};
        if (!cattr_set) {
          GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cs_count<KeyT, CK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4u << CS_MAXBITS)));
          cattr_set = true;
        }
        hipLaunchKernelGGL((k_cs_count<KeyT, CK>), dim3(256), dim3(CS_BT), (size_t)4 << CS_MAXBITS, stream, kin, n, desc_mask, plan, hist2);
        hipLaunchKernelGGL(k_cs_scan, dim3(1), dim3(CS_BT), 0, stream, plan, (const uint32_t*)hist2, base2, xoff, n, NPASS);
        hipLaunchKernelGGL((k_cs_fill<KeyT, CK>), dim3(4096), dim3(256), 0, stream, bufA, n, desc_mask, (const SortPlan*)plan, (const uint32_t*)base2,
                           (const uint32_t*)xoff);
      }
      hipLaunchKernelGGL((k_hf_sample<KeyT, CK, true>), dim3((unsigned)sblocks), dim3(256), 0, stream, kin, n, desc_mask, plan, fc.stride, frange);
      const int allow_split = (sizeof(KeyT) == 8 && g_split) ? 1 : 0;
      hipLaunchKernelGGL(k_hf_plan, dim3(1), dim3(BINS), 0, stream, plan, 1, (int)(8 * sizeof(KeyT)), n, fc.bits2, 1 << 13, fc.stride, frange, FT,
                         (unsigned long long)fc.slot_rows, g_cursor_margin, MIN_SHIFT2, fc.bits2_max, 0ull, 0, 0, allow_split);
      // splitter mode (round 5; no-ops unless stage 1 asked for it): plan the splitters from 16384 sampled keys, take the sample
      // histogram again on the splitter digit, size the level-0 slots from it (stage 1 once more), cut the column with k_sp_level0
      int64_t ftiles_s = 0;
      if constexpr (sizeof(KeyT) == 8) {
        if (allow_split) {
          static std::atomic<bool> sattr_set{false};
          if (!sattr_set) {
            GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sp_plan<CK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SP_NSAMP * 8)));
            GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sp_level0<CK, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp_level0_lds()));
            GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sp_level0<CK, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp_level0_lds()));
            sattr_set = true;
          }
          ftiles_s               = div_up(n, (int64_t)SP_TILE);
          const int64_t frange_s = (ftiles_s / NRANGE) * SP_TILE;
          hipLaunchKernelGGL((k_sp_plan<CK>), dim3(1), dim3(1024), (size_t)SP_NSAMP * 8, stream, kin, n, (uint64_t)desc_mask, plan, fc.bits2_max);
          // (half the workgroups of k_hf_sample: each flushes 2048 + 4096 counters)
          hipLaunchKernelGGL((k_sp_sample<CK>), dim3((unsigned)(sblocks > 1024 ? 1024 : sblocks)), dim3(256), 0, stream, kin, n, (uint64_t)desc_mask, plan, fc.stride, frange_s);
          hipLaunchKernelGGL(k_hf_plan, dim3(1), dim3(BINS), 0, stream, plan, 1, (int)(8 * sizeof(KeyT)), n, fc.bits2, 1 << 13, fc.stride, frange_s, FT,
                             (unsigned long long)fc.slot_rows, g_cursor_margin, MIN_SHIFT2, fc.bits2_max, 0ull, 0, 0, allow_split);
        }
      }
      prof_mark(1, stream);
      prof_mark_h(0, stream);
      hipLaunchKernelGGL(kf0, dim3((unsigned)ftiles), dim3(BT), lds_hf(256), stream, kin, slot0_buf, desc_mask, plan, hist2, 1u << 13, n, (KeyT*)nullptr);
      if constexpr (sizeof(KeyT) == 8) {
        if (allow_split)
        {
          hipLaunchKernelGGL((k_sp_level0<CK, 0>), dim3((unsigned)ftiles_s), dim3(BT), sp_level0_lds(), stream, kin, slot0_buf, (uint64_t)desc_mask, plan, n);
          hipLaunchKernelGGL((k_sp_level0<CK, 1>), dim3((unsigned)ftiles_s), dim3(BT), sp_level0_lds(), stream, kin, slot0_buf, (uint64_t)desc_mask, plan, n);
        }
      }
      hipLaunchKernelGGL(k_hf_plan, dim3(1), dim3(BINS), 0, stream, plan, 2, (int)(8 * sizeof(KeyT)), n, fc.bits2, 1 << 13, fc.stride, frange, FT,
                         (unsigned long long)fc.slot_rows, g_cursor_margin, MIN_SHIFT2, fc.bits2_max, (unsigned long long)nb_buf);
      prof_mark_h(1, stream);
      hipLaunchKernelGGL(kf1, dim3((unsigned)(ftiles + NRANGE * BINS)), dim3(BT), lds_hf(nbf), stream, slot0_buf, kb_scratch, desc_mask, plan, hist2, 1u << 13, n, bufA);
      prof_mark_h(2, stream);
      hipLaunchKernelGGL(k_plan2, dim3(BINS), dim3(GX_WAVE), 0, stream, plan, hist2, base2, NPASS, 1);
      prof_mark_h(3, stream);
      if constexpr (sizeof(KeyT) == 8) {
        if (allow_split)  // splitter mode: the narrow buckets (one value per cell) are filled from their cell starts
          hipLaunchKernelGGL((k_sp_fill<CK>), dim3(BINS, SP_FILL_Y), dim3(256), 0, stream, bufA, (uint64_t)desc_mask, (const SortPlan*)plan, (const uint32_t*)base2);
      }
      // (one workgroup per cell of the plan n suggests; when the device took the extra level-1 bit each of them walks two cells)
      hipLaunchKernelGGL((k_local_place<KeyT, CK, HAS_VAL, 13>), dim3(local_place_grid(BINS << fc.bits2)), dim3((1 << 13) / 16),
                         place_lds_bytes(13, WORD_BYTES), stream, kb_scratch, bufA, (const uint32_t*)nullptr, (uint32_t*)nullptr, desc_mask, plan, hist2,
                         base2, todo, g_exp, 1);
      // the cells k_local_place left: 64-bit words (32-bit keys are widened to their sortable form on the way in)
      hipLaunchKernelGGL((k_local_sort<uint64_t, CK, HAS_VAL, 13, KeyT>), dim3(local_sort_grid(BINS << fc.bits2_max)), dim3((1 << 13) / 16),
                         ((size_t)8 << 13) + (size_t)(((1 << 13) / 16 / GX_WAVE) * BINS + 32 + 2 * BINS) * 4, stream, (const KeyT*)kb_scratch, bufA,
                         (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint64_t)desc_mask, plan, hist2, base2, g_exp, 1, (const uint32_t*)todo);
      prof_mark_h(4, stream);
      g_prof.hybrid_marked = g_prof.enabled;
      cursor_marked        = true;
      // big cells (a hot value): plan X, fetch their keys again from the level-1 input into X = the level-1 buffer (every other
      // cell has left it by now); the LSD passes below then sort X between the two halves of the level-0 buffer
      hipLaunchKernelGGL(k_big_plan, dim3(1), dim3(BINS), 0, stream, plan, (unsigned long long)(fc.slot_rows / 2));
      hipLaunchKernelGGL(k_big_cells, dim3(BINS), dim3(GX_WAVE), 0, stream, (const SortPlan*)plan, (const uint32_t*)hist2, xoff, biglist);
      hipLaunchKernelGGL(kf2, dim3((unsigned)(ftiles + NRANGE * BINS)), dim3(BT), lds_hf(nbf), stream, slot0_buf, kb_scratch, desc_mask, plan, hist2, 1u << 13, n, bufA);
      hipLaunchKernelGGL(k_hf_clear_status, dim3(2048), dim3(256), 0, stream, plan, reinterpret_cast<uint4*>(status), status_words / 2);
    }
  }
  if constexpr (sizeof(KeyT) == 8) {
    if (try_hybrid) {
      // hybrid MSD path: every kernel below is a no-op unless the device-side plan enables it
      constexpr size_t pay   = HAS_VAL ? 4 : 0;
      constexpr bool STABLE  = KIND == K_FLOAT || HAS_VAL;
      constexpr bool SMALLOK = KIND != K_FLOAT;  // 8192-key cells and the 9-bit level 1 exist for integer keys (keys only and pairs)
      constexpr int SKPT     = HAS_VAL ? 10 : 16;  // keys per thread of their partition passes
      auto lds_msd = [&](int kpt, int nb) {  // the pairs' 9-bit pass: 16-bit per-wave counters (k_msd_pass, HIST16)
        const int wrows = STABLE ? ((HAS_VAL && nb == NB9) ? NW / 2 : NW) : 2;
        return (size_t)BT * kpt * (sizeof(KeyT) + pay) + (size_t)(wrows * nb + 2 * nb + 16 + 4) * 4;
      };
      auto lds_loc = [&](int cl2) { return ((size_t)sizeof(KeyT) << cl2) + (size_t)(((1 << cl2) / 16 / GX_WAVE) * BINS + 32 + 2 * BINS) * 4; };
      typedef void (*MsdK)(MsdArgs);
      MsdK kmsd0 = HAS_VAL ? (MsdK)k_msd_pass<KeyT, KIND, HAS_VAL, 10, 4, 8>
                           : (hyb_kpt == 8 ? (MsdK)k_msd_pass<KeyT, KIND, HAS_VAL, 8, 4, 8>
                                           : (hyb_kpt == 12 ? (MsdK)k_msd_pass<KeyT, KIND, HAS_VAL, 12, 4, 8> : (MsdK)k_msd_pass<KeyT, KIND, HAS_VAL, 16, 4, 8>));
      MsdK kmsd1 = kmsd0;
      auto kloc  = k_local_sort<KeyT, KIND, HAS_VAL, 14>;
      int ls_bt  = (1 << 14) / 16;
      static std::atomic<bool> hattr_set{false};
      if (!hattr_set) {
        const int lds_mmax = (int)lds_msd(16, BINS);
        GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msd_pass<KeyT, KIND, HAS_VAL, 8, 4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_mmax));
        GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msd_pass<KeyT, KIND, HAS_VAL, 10, 4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_mmax));
        GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msd_pass<KeyT, KIND, HAS_VAL, 12, 4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_mmax));
        GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msd_pass<KeyT, KIND, HAS_VAL, 16, 4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_mmax));
        GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kloc), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_loc(14)));
        if constexpr (SMALLOK) {
          GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msd_pass<KeyT, KIND, HAS_VAL, SKPT, 4, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_msd(SKPT, NB9)));
          GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_local_sort<KeyT, KIND, HAS_VAL, 13>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_loc(13)));
          GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_local_place<KeyT, KIND, HAS_VAL, 13>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)place_lds_bytes(13)));
        }
        GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_local_place<KeyT, KIND, HAS_VAL, 14>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)place_lds_bytes(14)));
        hattr_set = true;
      }
      int kpt1 = hyb_kpt;
      if constexpr (SMALLOK) {
        if (hc.bits2 > 8) {
          kmsd1 = (MsdK)k_msd_pass<KeyT, KIND, HAS_VAL, SKPT, 4, 9>;
          kpt1  = SKPT;
        }
        if constexpr (!HAS_VAL) if (g_lbw != 4 && hyb_kpt == 16) {  // A/B knob: predecessors examined per look-back round
          static std::atomic<bool> lattr_set{false};
          if (!lattr_set) {
            GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msd_pass<KeyT, KIND, HAS_VAL, 16, 8, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_msd(16, BINS)));
            GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msd_pass<KeyT, KIND, HAS_VAL, 16, 16, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_msd(16, BINS)));
            GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msd_pass<KeyT, KIND, HAS_VAL, 16, 8, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_msd(16, NB9)));
            GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msd_pass<KeyT, KIND, HAS_VAL, 16, 16, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_msd(16, NB9)));
            lattr_set = true;
          }
          kmsd0 = g_lbw == 8 ? (MsdK)k_msd_pass<KeyT, KIND, HAS_VAL, 16, 8, 8> : (MsdK)k_msd_pass<KeyT, KIND, HAS_VAL, 16, 16, 8>;
          if (hc.bits2 > 8) kmsd1 = g_lbw == 8 ? (MsdK)k_msd_pass<KeyT, KIND, HAS_VAL, 16, 8, 9> : (MsdK)k_msd_pass<KeyT, KIND, HAS_VAL, 16, 16, 9>;
          else kmsd1 = kmsd0;
        }
        if (hc.cl2 == 13) {
          kloc  = k_local_sort<KeyT, KIND, HAS_VAL, 13>;
          ls_bt = (1 << 13) / 16;
        }
      }
      if (kpt1 != hyb_kpt) return GX_EINTERNAL;  // the 9-bit pass exists for 16 keys per thread only (g_msd_kpt knob)
      // ---- up-front: varying bits + level-0 histogram (one read of the keys), plan
      hipLaunchKernelGGL((k_hy_hist<KeyT, KIND, true>), dim3((unsigned)hblocks), dim3(BT), 0, stream,
                         static_cast<const KeyT*>(keys_in), n, desc_mask, plan, range_rows);
      hipLaunchKernelGGL(k_hy_plan, dim3(1), dim3(BINS), 0, stream, plan, 0, (int)(8 * sizeof(KeyT)), n, hc.bits2, 1 << hc.cl2,
                         HAS_VAL ? hc.cl2 : 0, range_rows, (int)msd_tile, base1, 0, KIND == K_SIGNED ? 1 : 0);
      hipLaunchKernelGGL((k_hy_hist<KeyT, KIND, false>), dim3((unsigned)hblocks), dim3(BT), 0, stream,
                         static_cast<const KeyT*>(keys_in), n, desc_mask, plan, range_rows);
      hipLaunchKernelGGL(k_hy_plan, dim3(1), dim3(BINS), 0, stream, plan, 1, (int)(8 * sizeof(KeyT)), n, hc.bits2, 1 << hc.cl2,
                         HAS_VAL ? hc.cl2 : 0, range_rows, (int)msd_tile, base1, hc.cl2_alt ? (1 << hc.cl2_alt) : 0);
      if (!cursor_marked) prof_mark(1, stream);
      KeyT* bufA = keys_out ? static_cast<KeyT*>(keys_out) : ka_scratch;
      KeyT* bufB = kb_scratch;
      uint32_t* valA = reinterpret_cast<uint32_t*>(vals_out);
      uint32_t* valB = vb;
      MsdArgs m;
      m.vin  = nullptr;  // iota
      m.vout = valA;
      m.plan      = plan;
      m.status    = status;
      m.n         = n;
      m.desc_mask = (uint64_t)desc_mask;
      m.in        = keys_in;
      m.out       = bufA;
      m.base      = base1;
      m.level     = 0;
      m.exp       = ((HAS_VAL && keys_out == nullptr) ? 64 : 0) | (g_exp & 32);  // bit 6: the caller wants the permutation only (sorted_order): no key
                                                                                  // write-back; bit 5 (knob): k_local_sort's sub-bucket path for every cell
      m.cellcount = hist2;
      m.cellcap   = 1u << hc.cl2;
      m.spin_ticks  = (unsigned long long)g_spin_ms * 100000ull | (g_soft_fault ? SPIN_SOFT_BIT : 0ull);
      m.inject_tile = g_inject_tile;
      if (!cursor_marked) prof_mark_h(0, stream);
      hipLaunchKernelGGL(kmsd0, dim3((unsigned)(msd_ntiles + NRANGE)), dim3(BT), lds_msd(hyb_kpt, BINS), stream, m);
      if (!cursor_marked) prof_mark_h(1, stream);
      m.in    = bufA;
      m.out   = bufB;
      m.vin   = valA;
      m.vout  = valB;
      m.level = 1;
      hipLaunchKernelGGL(kmsd1, dim3((unsigned)(msd_ntiles + BINS + NRANGE)), dim3(BT), lds_msd(hyb_kpt, nb1), stream, m);
      if (!cursor_marked) prof_mark_h(2, stream);
      hipLaunchKernelGGL(k_plan2, dim3(BINS), dim3(GX_WAVE), 0, stream, plan, hist2, base2, NPASS, 0);
      if (!cursor_marked) prof_mark_h(3, stream);
      // the cells: k_local_place sorts all but the crowded ones, k_local_sort those (and every cell when the placement does not apply)
      auto kplace = k_local_place<KeyT, KIND, HAS_VAL, 14>;
      if constexpr (SMALLOK) {
        if (hc.cl2 == 13) kplace = k_local_place<KeyT, KIND, HAS_VAL, 13>;
      }
      hipLaunchKernelGGL(kplace, dim3(local_place_grid(BINS << hc.bits2)), dim3(ls_bt), place_lds_bytes(hc.cl2), stream, (const KeyT*)bufB, bufA,
                         (const uint32_t*)valB, valA, desc_mask, plan, hist2, base2, todo, m.exp, 0);
      hipLaunchKernelGGL(kloc, dim3(local_sort_grid(BINS << hc.bits2)), dim3(ls_bt), lds_loc(hc.cl2), stream, bufB, bufA, valB, valA, desc_mask,
                         plan, hist2, base2, m.exp, 0, (const uint32_t*)todo);
      if (hc.cl2_alt == 14) {  // the cell sort on 16384-key cells, should k_hy_plan have switched to them (no-ops otherwise)
        hipLaunchKernelGGL((k_local_place<KeyT, KIND, HAS_VAL, 14>), dim3(local_place_grid(BINS << hc.bits2)), dim3((1 << 14) / 16), place_lds_bytes(14), stream,
                           (const KeyT*)bufB, bufA, (const uint32_t*)valB, valA, desc_mask, plan, hist2, base2, todo, m.exp, 0);
        hipLaunchKernelGGL((k_local_sort<KeyT, KIND, HAS_VAL, 14>), dim3(local_sort_grid(BINS << hc.bits2)), dim3((1 << 14) / 16), lds_loc(14), stream, bufB, bufA,
                           valB, valA, desc_mask, plan, hist2, base2, m.exp, 0, (const uint32_t*)todo);
      }
      if (!cursor_marked) prof_mark_h(4, stream);
      if (!cursor_marked) g_prof.hybrid_marked = g_prof.enabled;
    }
  }
  // LSD path: byte histograms + plan (no-ops when the hybrid path has sorted the column)
  hipLaunchKernelGGL((k_hist_all<KeyT, KIND>), dim3((unsigned)hblocks), dim3(BT), 0, stream,
                     static_cast<const KeyT*>(keys_in), n, desc_mask, plan, range_rows, (const KeyT*)kb_scratch);
  hipLaunchKernelGGL(k_plan, dim3(1), dim3(BINS), 0, stream, plan, NPASS, n, KIND == K_SIGNED ? 1 : 0);
  if (!try_hybrid && !cursor_marked) prof_mark(1, stream);

  PassArgs a;
  a.kbuf[0]   = const_cast<void*>(keys_in);
  a.kbuf[1]   = keys_out ? keys_out : static_cast<void*>(ka_scratch);
  a.kbuf[2]   = kb_scratch;
  a.kbufB[0]  = kb_scratch;                                              // X (HybridPlan::lsd_mode)
  a.kbufB[1]  = slot0_buf;                                               // sorted X
  a.kbufB[2]  = slot0_buf ? slot0_buf + fc.slot_rows / 2 : nullptr;
  a.vbuf[0]   = reinterpret_cast<uint32_t*>(const_cast<int32_t*>(vals_in));
  a.vbuf[1]   = reinterpret_cast<uint32_t*>(vals_out);
  a.vbuf[2]   = vb;
  a.plan      = plan;
  a.status    = status;
  a.tile_off  = tile_hist;
  a.n         = n;
  a.ntiles    = ntiles;
  a.desc_mask = (uint64_t)desc_mask;
  a.order_mode = g_order_mode;
  a.spin_ticks  = (unsigned long long)g_spin_ms * 100000ull | (g_soft_fault ? SPIN_SOFT_BIT : 0ull);
  a.inject_tile = g_inject_tile;

  constexpr size_t lds = pass_lds_bytes<KeyT, HAS_VAL, KPT>();
  auto kern_lb         = (algo == 2) ? k_radix_pass<KeyT, KIND, HAS_VAL, KPT, 1>
                                       : (cells ? k_radix_pass<KeyT, KIND, HAS_VAL, KPT, 4, true> : k_radix_pass<KeyT, KIND, HAS_VAL, KPT, 4>);
  auto kern_pre        = k_radix_pass<KeyT, KIND, HAS_VAL, KPT, 0>;
  static std::atomic<bool> attr_set{false};  // per template instantiation
  if (!attr_set) {
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_radix_pass<KeyT, KIND, HAS_VAL, KPT, 4>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_radix_pass<KeyT, KIND, HAS_VAL, KPT, 1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_radix_pass<KeyT, KIND, HAS_VAL, KPT, 4, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern_pre),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  for (int pass = 0; pass < NPASS; ++pass) {
    a.pass = pass;
    prof_mark(2 + 2 * pass, stream);
    if (algo != 1) {
      const int64_t lb_grid = (cells && algo != 2) ? div_up(ntiles, PASS_TPB) : ntiles;
      hipLaunchKernelGGL(kern_lb, dim3((unsigned)lb_grid), dim3(BT), lds, stream, a);
    } else {
      hipLaunchKernelGGL((k_tile_hist<KeyT, KIND, KPT>), dim3((unsigned)ntiles), dim3(BT), 0, stream, a, tile_hist);
      scan::PlainLoader<uint32_t, uint32_t> ld{tile_hist, nullptr, 0u};
      int rc = scan::device_scan<uint32_t, uint32_t>(ld, ntiles * BINS, 0u, SumOp(), false, tile_hist, partials,
                                                     stream, &plan->pass_skip[pass]);
      if (rc) return rc;
      hipLaunchKernelGGL(kern_pre, dim3((unsigned)ntiles), dim3(BT), lds, stream, a);
    }
    prof_mark(3 + 2 * pass, stream);
  }
  {
    int64_t blocks = div_up(n, 256 * 8);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((k_finalize_copy<KeyT, HAS_VAL>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    if (fc.on)
      hipLaunchKernelGGL((k_big_distribute<KeyT>), dim3(2048), dim3(256), 0, stream, plan, (const KeyT*)slot0_buf,
                         keys_out ? static_cast<KeyT*>(keys_out) : ka_scratch, (const uint32_t*)biglist, (const uint32_t*)base2);
    if (KIND == K_FLOAT && descending && radix_nan_rule && sizeof(KeyT) >= 4) {
      hipLaunchKernelGGL((k_reverse_nan_block<KeyT, HAS_VAL>), dim3(256), dim3(256), 0, stream,
                         static_cast<KeyT*>(a.kbuf[1]), a.vbuf[1], n);
    }
  }
  GX_LAUNCH_CHECK();
  return 0;
}


// ==================================================================================================================
// The SHARDED sort's two halves (include/cudf_amd/gx.h gx_sortx_*): the exchange of a distributed sort
// sits BETWEEN the cursor path's two partition levels.  Every rank runs level 0 on its shard with digit positions all ranks
// agree on (the OR of everybody's varying-bit masks), whole level-0 bins are dealt to ranks in contiguous runs balanced by the
// all-gathered histogram, a rank sends the span of its level-0 buffer that holds a peer's bins -- slots of (bin, input range),
// padding included, ONE message per peer -- and the receiver runs level 1 + the cell sort over regions (bucket, source rank,
// input range) of what arrived.  No separate range-partition pass on the sender, no level 0 on the receiver
// (the reference's shape: sample -> boundaries -> shuffle -> local sort, cudf_polars collectives/sort.py).
// ==================================================================================================================
struct SortxCfg {
  int bits2, bits2_max, stride;
  size_t slot_rows;
};
static inline SortxCfg sortx_cfg(int64_t n)
{
  SortxCfg f{1, 2, 8, 0};
  int B = 9;
  while (B < 18 && (double)n / (double)(1ull << B) > 0.955 * 8192.0) ++B;
  f.bits2     = B - 8;
  f.bits2_max = f.bits2 < 10 ? f.bits2 + 1 : 10;
  f.stride    = n >= (1ll << 27) ? 32 : 8;
  const double dev = 8.0 * 1.25 * f.stride * __builtin_sqrt((double)(NRANGE * BINS) * ((double)n / f.stride + 2.0 * NRANGE * BINS));
  f.slot_rows = (size_t)((double)n * 1.002 + dev) + (size_t)(NRANGE * BINS) * (size_t)(2 * f.stride * HF_CHUNK + 64 + 16) + 65536;
  return f;
}
constexpr int SORTX_MAXREG = BINS * 16 * NRANGE;  // regions a receiver can see: every bin from 16 ranks x 8 input ranges

// Receiver: one block, thread t = level-0 bucket t.  From the external region table (bucket-major; breg0[t] = first region of
// bucket t) everything k_hf_plan's stage 2 leaves behind: bucket sizes and output bases, the level-1 digit (one more bit when
// four buckets are that full), the tile numbering of the regions, the local sort's digits.
__global__ void __launch_bounds__(BINS) k_hfx_plan(SortPlan* plan, long long n, int bits2, int bits2_max, int cell_max, int min_shift2, int tile_rows,
                                                   unsigned long long or_mask, unsigned long long nor_mask, uint32_t nreg,
                                                   const uint32_t* __restrict__ breg0, uint32_t* __restrict__ x_tile0,
                                                   const uint32_t* __restrict__ x_start, const uint32_t* __restrict__ x_count,
                                                   const uint32_t* __restrict__ x_bucket, unsigned long long cell_budget)
{
  __shared__ uint32_t s_tmp[BINS / GX_WAVE + 1];
  HybridPlan& hy = plan->hy;
  FastPlan& hf   = plan->hf;
  const int t    = threadIdx.x;
  const unsigned long long V = or_mask & nor_mask;
  const int top    = V ? 63 - __builtin_clzll(V) : 0;
  const int shift0 = top - 7;
  uint32_t c = 0, tiles = 0;
  for (uint32_t q = breg0[t]; q < breg0[t + 1]; ++q) {
    c += x_count[q];
    tiles += (x_count[q] + (uint32_t)tile_rows - 1) / (uint32_t)tile_rows;
  }
  uint32_t total, ttotal;
  const uint32_t exc = block_exclusive_scan<BINS>(c, 0u, SumOp(), s_tmp, &total);
  uint32_t trun      = block_exclusive_scan<BINS>(tiles, 0u, SumOp(), s_tmp, &ttotal);
  const uint32_t fit = (uint32_t)(0.97 * (double)cell_max);
  int b2             = bits2;
  const int more     = __syncthreads_count(((unsigned long long)c >> bits2) > (unsigned long long)fit ? 1 : 0) >= 4;
  if (more && b2 < bits2_max && shift0 - b2 - 1 >= min_shift2) ++b2;
  const int shift2 = shift0 - b2;
  if (V == 0 || shift2 < min_shift2 || (long long)total != n) return;  // state stays 0: gx_sortx_status reports the failure
  hy.hist0[t] = c;
  hy.gbin0[t] = exc;
  plan_cell_slots(hy, c, b2, cell_max, s_tmp, cell_budget);
  for (uint32_t q = breg0[t]; q < breg0[t + 1]; ++q) {
    x_tile0[q] = trun;
    trun += (x_count[q] + (uint32_t)tile_rows - 1) / (uint32_t)tile_rows;
  }
  if (t == 0) {
    x_tile0[nreg] = ttotal;
    hy.attempt  = 1;
    hy.shift0   = shift0;
    hy.bits2    = b2;
    hy.shift2   = shift2;
    hy.cell_max = cell_max;
    hy.or_mask  = or_mask;
    hy.nor_mask = nor_mask;
    plan_local_digits(hy, V, shift2);
    hy.fold     = 0;  // (the sharded sort's digits come from the masks of all ranks: no sign fold)
    hf.ext      = 1;
    hf.nreg     = nreg;
    hf.x_tile0  = x_tile0;
    hf.x_start  = x_start;
    hf.x_count  = x_count;
    hf.x_bucket = x_bucket;
    __threadfence();
    hf.state = 3;
  }
}

template <typename KeyT>
struct SortxLayout {
  SortPlan* plan;
  uint32_t *hist2, *base2, *xoff, *todo, *biglist;
  unsigned long long* status;
  size_t status_words;
  KeyT *cells, *level0;
  size_t cells_rows, level0_rows;  // level0 = this rank's slots (slot_rows of its own n) followed by the receive area
  uint32_t *x_tile0, *x_start, *x_count, *x_bucket, *breg0;
  size_t total;
};
template <typename KeyT>
static SortxLayout<KeyT> sortx_layout(void* tmp, int64_t n, int64_t recv_rows_max)
{
  SortxLayout<KeyT> L{};
  Carver c(tmp);
  const SortxCfg cs = sortx_cfg(n), cr = sortx_cfg(recv_rows_max);
  L.plan          = c.take<SortPlan>(1);
  L.hist2         = c.take<uint32_t>((size_t)4 * BINS * NB2MAX);
  L.base2         = L.hist2 ? L.hist2 + BINS * NB2MAX : nullptr;
  L.xoff          = L.hist2 ? L.hist2 + 2 * BINS * NB2MAX : nullptr;
  L.todo          = c.take<uint32_t>((size_t)BINS * NB2MAX);
  L.biglist       = c.take<uint32_t>((size_t)2 * BIG_LIST);
  L.level0_rows   = cs.slot_rows + (size_t)recv_rows_max + 65536;
  L.cells_rows    = (size_t)recv_rows_max + (size_t)recv_rows_max / 16 + ((size_t)BINS << cr.bits2_max) * cell_slack(1 << 13) + 65536;  // per-bucket cell slots (HybridPlan::ccap)
  const size_t xt = L.level0_rows / 2 / (size_t)(BT * 16) + BINS + 2 * NRANGE;  // tiles of the largest X the LSD passes may sort
  L.status_words  = xt * BINS;
  L.status        = c.take<unsigned long long>(L.status_words);
  L.cells         = c.take<KeyT>(L.cells_rows);
  L.level0        = c.take<KeyT>(L.level0_rows);
  L.x_tile0       = c.take<uint32_t>((size_t)SORTX_MAXREG + 1);
  L.x_start       = c.take<uint32_t>((size_t)SORTX_MAXREG);
  L.x_count       = c.take<uint32_t>((size_t)SORTX_MAXREG);
  L.x_bucket      = c.take<uint32_t>((size_t)SORTX_MAXREG);
  L.breg0         = c.take<uint32_t>((size_t)BINS + 1);
  L.total         = c.total();
  return L;
}

// sender, step 1: varying-bit masks + speculative top-byte histogram of a sample of this rank's keys
template <typename KeyT, int KIND>
int sortx_sample(const void* keys, int64_t n, int64_t recv_rows_max, void* tmp, size_t* tmp_bytes, hipStream_t stream)
{
  auto L = sortx_layout<KeyT>(tmp, n, recv_rows_max);
  if (!tmp) {
    *tmp_bytes = L.total;
    return 0;
  }
  if (*tmp_bytes < L.total) return GX_ETMP;
  const SortxCfg cs = sortx_cfg(n);
  GX_HIP_TRY(hipMemsetAsync(L.plan, 0, sizeof(SortPlan), stream));
  GX_HIP_TRY(hipMemsetAsync(L.hist2, 0, (size_t)4 * BINS * NB2MAX * sizeof(uint32_t), stream));
  if (n == 0) return 0;
  constexpr int FT      = BT * hf_kpt<KeyT>();
  const int64_t ftiles  = div_up(n, (int64_t)FT);
  const int64_t frange  = (ftiles / NRANGE) * FT;
  const int64_t step    = (int64_t)cs.stride * HF_CHUNK;
  int64_t sblocks       = div_up(div_up(n, step), (int64_t)4 * 4);
  if (sblocks > 2048) sblocks = 2048;
  hipLaunchKernelGGL((k_hf_sample<KeyT, KIND, false>), dim3((unsigned)sblocks), dim3(256), 0, stream, static_cast<const KeyT*>(keys), n, KeyT(0), L.plan, cs.stride, frange);
  GX_LAUNCH_CHECK();
  return 0;
}

// sender, step 2: level 0 with the digit positions of masks2 = {OR, NOR} over ALL ranks (sortable form)
template <typename KeyT, int KIND>
int sortx_level0(const void* keys, int64_t n, int64_t recv_rows_max, const unsigned long long* masks2_host, int bits2_hint, void* tmp, hipStream_t stream)
{
  auto L = sortx_layout<KeyT>(tmp, n, recv_rows_max);
  if (n == 0) return 0;
  const SortxCfg cs = sortx_cfg(n);
  constexpr int FT      = BT * hf_kpt<KeyT>();
  constexpr int MIN_SHIFT2 = 8;
  const int64_t ftiles  = div_up(n, (int64_t)FT);
  const int64_t frange  = (ftiles / NRANGE) * FT;
  const int64_t step    = (int64_t)cs.stride * HF_CHUNK;
  int64_t sblocks       = div_up(div_up(n, step), (int64_t)4 * 4);
  if (sblocks > 2048) sblocks = 2048;
  const size_t lds0 = (size_t)FT * sizeof(KeyT) + (size_t)(3 * 256 + 16 + 4) * 4 + (size_t)2 * NW * 8;
  static std::atomic<bool> attr_set{false};  // per instantiation
  if (!attr_set) {
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hf_scatter<KeyT, KIND, 0, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0));
    attr_set = true;
  }
  // the masks of all ranks replace the sample's; forced_masks tells the verdict after level 0 not to compare the local top bit
  GX_HIP_TRY(hipMemcpyAsync(&L.plan->hy.or_mask, masks2_host, 2 * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
  const uint32_t one = 1;
  GX_HIP_TRY(hipMemcpyAsync(&L.plan->hf.forced_masks, &one, sizeof(one), hipMemcpyHostToDevice, stream));
  GX_HIP_TRY(hipMemcpyAsync(&L.plan->hf.forced_or, masks2_host, 2 * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
  GX_HIP_TRY(hipStreamSynchronize(stream));  // (host sources)
  const KeyT* kin = static_cast<const KeyT*>(keys);
  // bits2 only decides whether enough key bits are left below level 1 (the receiver picks its own from what it receives)
  hipLaunchKernelGGL(k_hf_plan, dim3(1), dim3(BINS), 0, stream, L.plan, 0, (int)(8 * sizeof(KeyT)), n, bits2_hint, 1 << 13, cs.stride, frange, FT,
                     (unsigned long long)cs.slot_rows, 8.0f, MIN_SHIFT2, bits2_hint);
  hipLaunchKernelGGL((k_hf_sample<KeyT, KIND, true>), dim3((unsigned)sblocks), dim3(256), 0, stream, kin, n, KeyT(0), L.plan, cs.stride, frange);
  hipLaunchKernelGGL(k_hf_plan, dim3(1), dim3(BINS), 0, stream, L.plan, 1, (int)(8 * sizeof(KeyT)), n, bits2_hint, 1 << 13, cs.stride, frange, FT,
                     (unsigned long long)cs.slot_rows, 8.0f, MIN_SHIFT2, bits2_hint);
  hipLaunchKernelGGL((k_hf_scatter<KeyT, KIND, 0, 8>), dim3((unsigned)ftiles), dim3(BT), lds0, stream, kin, L.level0, KeyT(0), L.plan, L.hist2, 1u << 13, n, (KeyT*)nullptr);
  hipLaunchKernelGGL(k_hf_plan, dim3(1), dim3(BINS), 0, stream, L.plan, 2, (int)(8 * sizeof(KeyT)), n, bits2_hint, 1 << 13, cs.stride, frange, FT,
                     (unsigned long long)cs.slot_rows, 8.0f, MIN_SHIFT2, bits2_hint);
  GX_LAUNCH_CHECK();
  return 0;
}

// receiver: level 1 + cell sort over the regions of the level-0 area (own slots + what arrived), sorted keys to `out`
template <typename KeyT, int KIND>
int sortx_finish(int64_t n_send, int64_t recv_rows_max, int64_t n, const unsigned long long* masks2_host, const uint32_t* reg_start_host,
                 const uint32_t* reg_count_host, const uint32_t* reg_bucket_host, int nreg, void* out, void* tmp, hipStream_t stream)
{
  auto L = sortx_layout<KeyT>(tmp, n_send, recv_rows_max);
  if (nreg < 0 || nreg > SORTX_MAXREG || n > recv_rows_max + n_send) return GX_EINVAL;
  // a fresh plan: the sender's is spent (its tables have been read by the caller)
  GX_HIP_TRY(hipMemsetAsync(L.plan, 0, sizeof(SortPlan), stream));
  GX_HIP_TRY(hipMemsetAsync(L.hist2, 0, (size_t)4 * BINS * NB2MAX * sizeof(uint32_t), stream));
  if (n == 0) return 0;
  const SortxCfg cr = sortx_cfg(n), cmax = sortx_cfg(recv_rows_max);
  if (cr.bits2_max > cmax.bits2_max) return GX_EINVAL;
  std::vector<uint32_t> breg0(BINS + 1, 0);
  for (int q = 0; q < nreg; ++q) {
    if (reg_bucket_host[q] >= (uint32_t)BINS || (q > 0 && reg_bucket_host[q] < reg_bucket_host[q - 1])) return GX_EINVAL;  // bucket-major
    ++breg0[reg_bucket_host[q] + 1];
  }
  for (int b = 0; b < BINS; ++b) breg0[b + 1] += breg0[b];
  GX_HIP_TRY(hipMemcpyAsync(L.x_start, reg_start_host, sizeof(uint32_t) * nreg, hipMemcpyHostToDevice, stream));
  GX_HIP_TRY(hipMemcpyAsync(L.x_count, reg_count_host, sizeof(uint32_t) * nreg, hipMemcpyHostToDevice, stream));
  GX_HIP_TRY(hipMemcpyAsync(L.x_bucket, reg_bucket_host, sizeof(uint32_t) * nreg, hipMemcpyHostToDevice, stream));
  GX_HIP_TRY(hipMemcpyAsync(L.breg0, breg0.data(), sizeof(uint32_t) * (BINS + 1), hipMemcpyHostToDevice, stream));
  GX_HIP_TRY(hipStreamSynchronize(stream));  // (host sources)
  constexpr int FT         = BT * hf_kpt<KeyT>();
  constexpr int MIN_SHIFT2 = 8;
  constexpr int WORD_BYTES = (int)sizeof(typename PlaceWord<KeyT, KIND, false>::type);
  auto lds_hf = [&](int nb) { return (size_t)FT * sizeof(KeyT) + (size_t)(3 * nb + 16 + 4) * 4 + (size_t)2 * NW * 8; };
  typedef void (*HfK)(const KeyT*, KeyT*, KeyT, SortPlan*, uint32_t*, uint32_t, int64_t, KeyT*);
  HfK kf1 = cr.bits2_max <= 8 ? (HfK)k_hf_scatter<KeyT, KIND, 1, 8> : (cr.bits2_max == 9 ? (HfK)k_hf_scatter<KeyT, KIND, 1, 9> : (HfK)k_hf_scatter<KeyT, KIND, 1, 10>);
  HfK kf2 = cr.bits2_max <= 8 ? (HfK)k_hf_scatter<KeyT, KIND, 2, 8> : (cr.bits2_max == 9 ? (HfK)k_hf_scatter<KeyT, KIND, 2, 9> : (HfK)k_hf_scatter<KeyT, KIND, 2, 10>);
  const int nbf = cr.bits2_max <= 8 ? 256 : (1 << cr.bits2_max);
  const size_t lds_ls = ((size_t)8 << 13) + (size_t)(((1 << 13) / 16 / GX_WAVE) * BINS + 32 + 2 * BINS) * 4;
  constexpr int KPT_L = kpt_for<KeyT>(false);
  constexpr size_t lds_pass = pass_lds_bytes<KeyT, false, KPT_L>();
  static std::atomic<bool> attr_set{false};  // per instantiation
  if (!attr_set) {
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hf_scatter<KeyT, KIND, 1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hf(256)));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hf_scatter<KeyT, KIND, 1, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hf(512)));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hf_scatter<KeyT, KIND, 1, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hf(1024)));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hf_scatter<KeyT, KIND, 2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hf(256)));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hf_scatter<KeyT, KIND, 2, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hf(512)));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hf_scatter<KeyT, KIND, 2, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hf(1024)));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_local_sort<uint64_t, KIND, false, 13, KeyT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ls));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_local_place<KeyT, KIND, false, 13>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)place_lds_bytes(13, WORD_BYTES)));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_radix_pass<KeyT, KIND, false, KPT_L, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pass));
    attr_set = true;
  }
  // tiles of the regions: one per FT keys + one tail per region
  const int64_t l1_grid = div_up(n, (int64_t)FT) + nreg;
  KeyT* bufA = static_cast<KeyT*>(out);
  hipLaunchKernelGGL(k_hfx_plan, dim3(1), dim3(BINS), 0, stream, L.plan, (long long)n, cr.bits2, cr.bits2_max, 1 << 13, MIN_SHIFT2, FT, masks2_host[0], masks2_host[1],
                     (uint32_t)nreg, (const uint32_t*)L.breg0, L.x_tile0, (const uint32_t*)L.x_start, (const uint32_t*)L.x_count, (const uint32_t*)L.x_bucket,
                     (unsigned long long)L.cells_rows);
  hipLaunchKernelGGL(kf1, dim3((unsigned)l1_grid), dim3(BT), lds_hf(nbf), stream, (const KeyT*)L.level0, L.cells, KeyT(0), L.plan, L.hist2, 1u << 13, n, (KeyT*)nullptr);
  hipLaunchKernelGGL(k_plan2, dim3(BINS), dim3(GX_WAVE), 0, stream, L.plan, L.hist2, L.base2, (int)sizeof(KeyT), 1);
  hipLaunchKernelGGL((k_local_place<KeyT, KIND, false, 13>), dim3(local_place_grid(BINS << cr.bits2)), dim3((1 << 13) / 16), place_lds_bytes(13, WORD_BYTES), stream,
                     (const KeyT*)L.cells, bufA, (const uint32_t*)nullptr, (uint32_t*)nullptr, KeyT(0), L.plan, L.hist2, L.base2, L.todo, 0, 1);
  hipLaunchKernelGGL((k_local_sort<uint64_t, KIND, false, 13, KeyT>), dim3(local_sort_grid(BINS << cr.bits2_max)), dim3((1 << 13) / 16), lds_ls, stream,
                     (const KeyT*)L.cells, bufA, (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint64_t)0, L.plan, L.hist2, L.base2, 0, 1, (const uint32_t*)L.todo);
  // big cells: X in the cell buffer, the LSD passes between the two halves of the level-0 area (every region has been read by then)
  const size_t xcap = L.level0_rows / 2;
  hipLaunchKernelGGL(k_big_plan, dim3(1), dim3(BINS), 0, stream, L.plan, (unsigned long long)xcap);
  hipLaunchKernelGGL(k_big_cells, dim3(BINS), dim3(GX_WAVE), 0, stream, (const SortPlan*)L.plan, (const uint32_t*)L.hist2, L.xoff, L.biglist);
  hipLaunchKernelGGL(kf2, dim3((unsigned)l1_grid), dim3(BT), lds_hf(nbf), stream, (const KeyT*)L.level0, L.cells, KeyT(0), L.plan, L.hist2, 1u << 13, n, (KeyT*)nullptr);
  hipLaunchKernelGGL(k_hf_clear_status, dim3(2048), dim3(256), 0, stream, L.plan, reinterpret_cast<uint4*>(L.status), L.status_words / 2);
  int64_t hblocks = div_up(n, (int64_t)BT * 8);
  if (hblocks > 2048) hblocks = 2048;
  hblocks = div_up(hblocks, NRANGE) * NRANGE;
  // (mode A -- the whole-column LSD fallback -- has no column to fall back to here: its length is 0, a failed plan is reported
  //  by gx_sortx_status and the caller takes the range-partition path)
  hipLaunchKernelGGL((k_hist_all<KeyT, KIND>), dim3((unsigned)hblocks), dim3(BT), 0, stream, (const KeyT*)L.cells, (int64_t)0, KeyT(0), L.plan, (int64_t)0, (const KeyT*)L.cells);
  hipLaunchKernelGGL(k_plan, dim3(1), dim3(BINS), 0, stream, L.plan, (int)sizeof(KeyT), (int64_t)0, KIND == K_SIGNED ? 1 : 0);
  PassArgs a{};
  a.kbuf[0] = a.kbuf[1] = a.kbuf[2] = L.cells;
  a.kbufB[0] = L.cells;
  a.kbufB[1] = L.level0;
  a.kbufB[2] = L.level0 + xcap;
  a.plan      = L.plan;
  a.status    = L.status;
  a.n         = 0;
  a.ntiles    = 0;
  a.desc_mask = 0;
  const int64_t xtiles = div_up((int64_t)(n < (int64_t)xcap ? n : (int64_t)xcap), (int64_t)(BT * KPT_L));
  for (int pass = 0; pass < (int)sizeof(KeyT); ++pass) {
    a.pass = pass;
    hipLaunchKernelGGL((k_radix_pass<KeyT, KIND, false, KPT_L, 4, true>), dim3((unsigned)div_up(xtiles, (int64_t)PASS_TPB)), dim3(BT), lds_pass, stream, a);
  }
  hipLaunchKernelGGL((k_finalize_copy<KeyT, false>), dim3(1024), dim3(256), 0, stream, a);
  hipLaunchKernelGGL((k_big_distribute<KeyT>), dim3(2048), dim3(256), 0, stream, (const SortPlan*)L.plan, (const KeyT*)L.level0, bufA, (const uint32_t*)L.biglist,
                     (const uint32_t*)L.base2);
  GX_LAUNCH_CHECK();
  return 0;
}

template <bool HAS_VAL>
int dispatch(int dtype, const void* keys_in, void* keys_out, const int32_t* vals_in, int32_t* vals_out,
             int64_t n, int descending, bool radix_nan_rule, void* tmp, size_t* tmp_bytes, hipStream_t s)
{
  switch (dtype) {
    case GX_INT8: return sort_impl<uint8_t, K_SIGNED, HAS_VAL>(keys_in, keys_out, vals_in, vals_out, n, descending, radix_nan_rule, tmp, tmp_bytes, s);
    case GX_BOOL8:
    case GX_UINT8: return sort_impl<uint8_t, K_UNSIGNED, HAS_VAL>(keys_in, keys_out, vals_in, vals_out, n, descending, radix_nan_rule, tmp, tmp_bytes, s);
    case GX_INT16: return sort_impl<uint16_t, K_SIGNED, HAS_VAL>(keys_in, keys_out, vals_in, vals_out, n, descending, radix_nan_rule, tmp, tmp_bytes, s);
    case GX_UINT16: return sort_impl<uint16_t, K_UNSIGNED, HAS_VAL>(keys_in, keys_out, vals_in, vals_out, n, descending, radix_nan_rule, tmp, tmp_bytes, s);
    case GX_INT32: return sort_impl<uint32_t, K_SIGNED, HAS_VAL>(keys_in, keys_out, vals_in, vals_out, n, descending, radix_nan_rule, tmp, tmp_bytes, s);
    case GX_UINT32: return sort_impl<uint32_t, K_UNSIGNED, HAS_VAL>(keys_in, keys_out, vals_in, vals_out, n, descending, radix_nan_rule, tmp, tmp_bytes, s);
    case GX_FLOAT32: return sort_impl<uint32_t, K_FLOAT, HAS_VAL>(keys_in, keys_out, vals_in, vals_out, n, descending, radix_nan_rule, tmp, tmp_bytes, s);
    case GX_INT64: return sort_impl<uint64_t, K_SIGNED, HAS_VAL>(keys_in, keys_out, vals_in, vals_out, n, descending, radix_nan_rule, tmp, tmp_bytes, s);
    case GX_UINT64: return sort_impl<uint64_t, K_UNSIGNED, HAS_VAL>(keys_in, keys_out, vals_in, vals_out, n, descending, radix_nan_rule, tmp, tmp_bytes, s);
    case GX_FLOAT64: return sort_impl<uint64_t, K_FLOAT, HAS_VAL>(keys_in, keys_out, vals_in, vals_out, n, descending, radix_nan_rule, tmp, tmp_bytes, s);
    default: return GX_EDTYPE;
  }
}


// ---- nullable column: split rows by validity (stable), radix sort the valid run ----------------
struct BitLoader {
  const uint32_t* valid;
  __device__ __forceinline__ uint32_t operator()(int64_t i) const { return bit_is_set(valid, i) ? 1u : 0u; }
};

template <typename ElemT>
__global__ void __launch_bounds__(256) k_split_by_validity(const ElemT* __restrict__ keys,
                                                           const uint32_t* __restrict__ valid,
                                                           const uint32_t* __restrict__ pos, int64_t n,
                                                           ElemT* dense_keys, int32_t* dense_idx, int32_t* null_idx)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t p = pos[i];
    if (bit_is_set(valid, i)) {
      dense_keys[p] = keys[i];
      dense_idx[p]  = (int32_t)i;
    } else {
      null_idx[i - p] = (int32_t)i;
    }
  }
}

template <typename ElemT>
int sorted_order_nullable(int dtype, const void* keys, const uint32_t* valid, int64_t n, int64_t null_count,
                          int descending, int nulls_before, int32_t* out, void* tmp, size_t* tmp_bytes,
                          hipStream_t stream)
{
  const int64_t nvalid = n - null_count;
  size_t sort_bytes    = 0;
  int rc = dispatch<true>(dtype, nullptr, nullptr, nullptr, nullptr, nvalid, descending, false, nullptr,
                          &sort_bytes, stream);
  if (rc) return rc;
  Carver c(tmp);
  char* sort_tmp      = c.take<char>(sort_bytes);
  uint32_t* pos       = c.take<uint32_t>((size_t)n);
  uint32_t* partials  = c.take<uint32_t>(scan::partials_count(n));
  ElemT* dense_keys   = c.take<ElemT>((size_t)nvalid);
  int32_t* dense_idx  = c.take<int32_t>((size_t)nvalid);
  if (tmp == nullptr) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  if (n == 0) return 0;
  const bool nulls_first = (nulls_before != 0) != (descending != 0);
  rc = scan::device_scan<uint32_t, uint32_t>(BitLoader{valid}, n, 0u, SumOp(), false, pos, partials, stream);
  if (rc) return rc;
  int64_t blocks = div_up(n, 256 * 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((k_split_by_validity<ElemT>), dim3((unsigned)blocks), dim3(256), 0, stream,
                     static_cast<const ElemT*>(keys), valid, pos, n, dense_keys, dense_idx,
                     out + (nulls_first ? 0 : nvalid));
  GX_LAUNCH_CHECK();
  size_t sb = sort_bytes;
  return dispatch<true>(dtype, dense_keys, nullptr, dense_idx, out + (nulls_first ? null_count : 0), nvalid,
                        descending, false, sort_tmp, &sb, stream);
}

// ---- sorted_order of a 32-bit integer column as a KEYS-ONLY sort of 64-bit words (round 3) ------------------------------
// (sortable key << rbits) | row, rbits = the bits of n - 1, is a distinct 64-bit word whose unsigned order is the stable order
// of the pairs, so the cursor path of the 64-bit keys-only sort -- unstable, no look-back, k_local_place -- produces
// cudf::sorted_order / stable_sorted_order (sorted_order_radix.cu:83-94) of an int32 / uint32 column: pack (12 B/row), sort
// (48 B/row), unpack (12 B/row) instead of four stable LSD pair passes of 16 B/row each.  n >= 2^25, no nulls, the row payload
// implicit.  The row sits directly below the key (no constant bits in between), so whatever digit window the sort's levels
// and k_local_place's counting passes take, every bit in it varies: keys of a narrow range (dictionary codes, group ids) are
// split further by the top bits of the row.
template <typename K32, int KIND>
__global__ void __launch_bounds__(256) k_pack_words(const K32* __restrict__ keys, int64_t n, K32 desc_mask, int rbits, uint64_t* __restrict__ words)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    words[i] = ((uint64_t)to_sortable<K32, KIND>(keys[i], desc_mask) << rbits) | (uint64_t)i;
}
__global__ void __launch_bounds__(256) k_unpack_rows(const uint64_t* __restrict__ words, int64_t n, int rbits, int32_t* __restrict__ rows)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const uint64_t rmask = (1ull << rbits) - 1ull;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) rows[i] = (int32_t)(uint32_t)(words[i] & rmask);
}
// OFF by default (gx_sort_set_order_words): 1e9 well-spread int32 keys 22.2 -> 17.0 ms, but the sort's cells are cut by KEY bits
// alone until its two levels are through, so keys with ~1000 rows each (1e6 distinct: group ids) overflow their cells and the
// column falls back to eight LSD passes over the 64-bit words: 16-21 -> 55-59 ms (profiles/r3_run31_sorted_order_int32_words.txt).
static thread_local int g_order_words = 0;
static inline bool order_words32_applies(int dtype, int64_t n)
{
  return g_order_words && (dtype == GX_INT32 || dtype == GX_UINT32) && n >= (1ll << 25) && g_algorithm == 0 && g_hybrid && g_cursor;
}
template <typename K32, int KIND>
int sorted_order_words32(const void* keys, int64_t n, int descending, int32_t* out, void* tmp, size_t* tmp_bytes, hipStream_t stream)
{
  size_t sort_bytes = 0;
  int rc = sort_impl<uint64_t, K_UNSIGNED, false>(nullptr, nullptr, nullptr, nullptr, n, 0, false, nullptr, &sort_bytes, stream);
  if (rc) return rc;
  Carver c(tmp);
  char* sort_tmp  = c.take<char>(sort_bytes);  // first: the plan header callers inspect (gx_sort_status / gx_sort_info) sits at tmp
  uint64_t* w_in  = c.take<uint64_t>((size_t)n);
  uint64_t* w_out = c.take<uint64_t>((size_t)n);
  if (tmp == nullptr) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  if (keys == nullptr || out == nullptr) return GX_EINVAL;
  int64_t blocks = div_up(n, 256 * 8);
  if (blocks > 8192) blocks = 8192;
  int rbits = 1;
  while (((int64_t)1 << rbits) < n) ++rbits;  // rows 0 .. n - 1 fit rbits <= 31 bits
  hipLaunchKernelGGL((k_pack_words<K32, KIND>), dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<const K32*>(keys), n,
                     descending ? K32(~K32(0)) : K32(0), rbits, w_in);
  size_t sb = sort_bytes;
  rc = sort_impl<uint64_t, K_UNSIGNED, false>(w_in, w_out, nullptr, nullptr, n, 0, false, sort_tmp, &sb, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(k_unpack_rows, dim3((unsigned)blocks), dim3(256), 0, stream, w_out, n, rbits, out);
  GX_LAUNCH_CHECK();
  return 0;
}

}  // namespace sort
}  // namespace gx

extern "C" {

// ---- the sharded sort's halves (see gx.h)
#define GX_SORTX_DISPATCH(CALL64S, CALL64U, CALL32S, CALL32U)  \
  switch (dtype) {                                              \
    case GX_INT64: return CALL64S;                              \
    case GX_UINT64: return CALL64U;                             \
    case GX_INT32: return CALL32S;                              \
    case GX_UINT32: return CALL32U;                             \
    default: return GX_EDTYPE;                                  \
  }
int gx_sortx_sample(int dtype, const void* keys, int64_t n, int64_t recv_rows_max, void* tmp, size_t* tmp_bytes, gx_stream_t s)
{
  using namespace gx::sort;
  if (n < 0 || recv_rows_max < 0 || !tmp_bytes || n > 0x7FFFFFFFll || recv_rows_max > 0x7FFFFFFFll) return GX_EINVAL;
  if (tmp && n > 0 && !keys) return GX_EINVAL;
  hipStream_t st = (hipStream_t)s;
  GX_SORTX_DISPATCH((sortx_sample<uint64_t, gx::K_SIGNED>(keys, n, recv_rows_max, tmp, tmp_bytes, st)),
                    (sortx_sample<uint64_t, gx::K_UNSIGNED>(keys, n, recv_rows_max, tmp, tmp_bytes, st)),
                    (sortx_sample<uint32_t, gx::K_SIGNED>(keys, n, recv_rows_max, tmp, tmp_bytes, st)),
                    (sortx_sample<uint32_t, gx::K_UNSIGNED>(keys, n, recv_rows_max, tmp, tmp_bytes, st)))
}
int gx_sortx_masks(const void* tmp, uint64_t* masks2_host, gx_stream_t s)
{
  if (!tmp || !masks2_host) return GX_EINVAL;
  const auto* plan = static_cast<const gx::sort::SortPlan*>(tmp);
  GX_HIP_TRY(hipMemcpyAsync(masks2_host, &plan->hy.or_mask, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, (hipStream_t)s));
  GX_HIP_TRY(hipStreamSynchronize((hipStream_t)s));
  return 0;
}
int gx_sortx_level0(int dtype, const void* keys, int64_t n, int64_t recv_rows_max, const uint64_t* masks2_host, void* tmp, gx_stream_t s)
{
  using namespace gx::sort;
  if (n < 0 || !tmp || !masks2_host || (n > 0 && !keys)) return GX_EINVAL;
  hipStream_t st = (hipStream_t)s;
  const int hint = sortx_cfg(recv_rows_max).bits2_max;
  const auto* m  = reinterpret_cast<const unsigned long long*>(masks2_host);
  GX_SORTX_DISPATCH((sortx_level0<uint64_t, gx::K_SIGNED>(keys, n, recv_rows_max, m, hint, tmp, st)),
                    (sortx_level0<uint64_t, gx::K_UNSIGNED>(keys, n, recv_rows_max, m, hint, tmp, st)),
                    (sortx_level0<uint32_t, gx::K_SIGNED>(keys, n, recv_rows_max, m, hint, tmp, st)),
                    (sortx_level0<uint32_t, gx::K_UNSIGNED>(keys, n, recv_rows_max, m, hint, tmp, st)))
}
int gx_sortx_tables(const void* tmp, uint32_t* cur0_host, uint32_t* slot0_host, uint32_t* slot_total_host, int32_t* state_host, gx_stream_t s)
{
  using namespace gx::sort;
  if (!tmp || !cur0_host || !slot0_host || !slot_total_host || !state_host) return GX_EINVAL;
  const auto* plan = static_cast<const SortPlan*>(tmp);
  hipStream_t st   = (hipStream_t)s;
  GX_HIP_TRY(hipMemcpyAsync(cur0_host, &plan->hf.cur0[0][0], sizeof(uint32_t) * NRANGE * BINS, hipMemcpyDeviceToHost, st));
  GX_HIP_TRY(hipMemcpyAsync(slot0_host, &plan->hf.slot0[0][0], sizeof(uint32_t) * NRANGE * BINS, hipMemcpyDeviceToHost, st));
  GX_HIP_TRY(hipMemcpyAsync(slot_total_host, &plan->hf.slot_total, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  GX_HIP_TRY(hipMemcpyAsync(state_host, &plan->hf.state, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  GX_HIP_TRY(hipStreamSynchronize(st));
  return 0;
}
void* gx_sortx_level0_buffer(int dtype, void* tmp, int64_t n, int64_t recv_rows_max, int64_t* own_rows, int64_t* total_rows)
{
  using namespace gx::sort;
  if (!tmp || n < 0 || recv_rows_max < 0) return nullptr;
  if (dtype == GX_INT64 || dtype == GX_UINT64) {
    auto L = sortx_layout<uint64_t>(tmp, n, recv_rows_max);
    if (own_rows) *own_rows = (int64_t)sortx_cfg(n).slot_rows;
    if (total_rows) *total_rows = (int64_t)L.level0_rows;
    return L.level0;
  }
  if (dtype == GX_INT32 || dtype == GX_UINT32) {
    auto L = sortx_layout<uint32_t>(tmp, n, recv_rows_max);
    if (own_rows) *own_rows = (int64_t)sortx_cfg(n).slot_rows;
    if (total_rows) *total_rows = (int64_t)L.level0_rows;
    return L.level0;
  }
  return nullptr;
}
int gx_sortx_finish(int dtype, int64_t n_send, int64_t recv_rows_max, int64_t n, const uint64_t* masks2_host, const uint32_t* reg_start_host,
                    const uint32_t* reg_count_host, const uint32_t* reg_bucket_host, int nreg, void* out, void* tmp, gx_stream_t s)
{
  using namespace gx::sort;
  if (n < 0 || n_send < 0 || !tmp || !masks2_host || (nreg > 0 && (!reg_start_host || !reg_count_host || !reg_bucket_host)) || (n > 0 && !out)) return GX_EINVAL;
  hipStream_t st = (hipStream_t)s;
  const auto* m  = reinterpret_cast<const unsigned long long*>(masks2_host);
  GX_SORTX_DISPATCH((sortx_finish<uint64_t, gx::K_SIGNED>(n_send, recv_rows_max, n, m, reg_start_host, reg_count_host, reg_bucket_host, nreg, out, tmp, st)),
                    (sortx_finish<uint64_t, gx::K_UNSIGNED>(n_send, recv_rows_max, n, m, reg_start_host, reg_count_host, reg_bucket_host, nreg, out, tmp, st)),
                    (sortx_finish<uint32_t, gx::K_SIGNED>(n_send, recv_rows_max, n, m, reg_start_host, reg_count_host, reg_bucket_host, nreg, out, tmp, st)),
                    (sortx_finish<uint32_t, gx::K_UNSIGNED>(n_send, recv_rows_max, n, m, reg_start_host, reg_count_host, reg_bucket_host, nreg, out, tmp, st)))
}
int gx_sortx_status(const void* tmp, int32_t* ok_host, gx_stream_t s)
{
  if (!tmp || !ok_host) return GX_EINVAL;
  const auto* plan = static_cast<const gx::sort::SortPlan*>(tmp);
  int32_t state = 0, ok = 0, status = 0;
  GX_HIP_TRY(hipMemcpyAsync(&state, &plan->hf.state, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)s));
  GX_HIP_TRY(hipMemcpyAsync(&ok, &plan->hy.ok, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)s));
  GX_HIP_TRY(hipMemcpyAsync(&status, &plan->status, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)s));
  GX_HIP_TRY(hipStreamSynchronize((hipStream_t)s));
  *ok_host = (state == 3 && ok == 1 && status == 0) ? 1 : 0;
  return 0;
}
#undef GX_SORTX_DISPATCH

int gx_sort_keys(int dtype, const void* in, void* out, int64_t n, int descending, void* tmp,
                 size_t* tmp_bytes, gx_stream_t stream)
{
  if (tmp != nullptr && n > 0 && out == nullptr) return GX_EINVAL;
  return gx::sort::dispatch<false>(dtype, in, out, nullptr, nullptr, n, descending, true, tmp, tmp_bytes, stream);
}

int gx_sort_pairs(int key_dtype, const void* keys_in, void* keys_out, const int32_t* vals_in,
                  int32_t* vals_out, int64_t n, int descending, void* tmp, size_t* tmp_bytes,
                  gx_stream_t stream)
{
  return gx::sort::dispatch<true>(key_dtype, keys_in, keys_out, vals_in, vals_out, n, descending, true, tmp,
                                  tmp_bytes, stream);
}

int gx_order_map_applies(int dtype, int64_t n);                                                                                            // gx_order.hip
int gx_sorted_order_words(int dtype, const void* keys, int64_t n, int descending, int32_t* out, void* tmp, size_t* tmp_bytes, gx_stream_t s);  // gx_order.hip
int gx_sorted_order(int dtype, const void* keys, const uint32_t* valid, int64_t n, int64_t null_count,
                    int descending, int nulls_before, int32_t* out_indices, void* tmp, size_t* tmp_bytes,
                    gx_stream_t stream)
{
  if (n < 0 || null_count < 0 || null_count > n || tmp_bytes == nullptr) return GX_EINVAL;
  if (valid == nullptr || null_count == 0) {
    // round 6: 64-bit keys from 2^25 rows -- the argsort as a keys-only sort of (monotone rank, row) words (gx_order.hip): the cost of
    // the keys-only paths on ANY value distribution instead of the look-back pairs levels that decline uneven columns to LSD passes
    if (gx_order_map_applies(dtype, n)) return gx_sorted_order_words(dtype, keys, n, descending, out_indices, tmp, tmp_bytes, stream);
    if (gx::sort::order_words32_applies(dtype, n)) {
      return dtype == GX_INT32 ? gx::sort::sorted_order_words32<uint32_t, gx::K_SIGNED>(keys, n, descending, out_indices, tmp, tmp_bytes, stream)
                               : gx::sort::sorted_order_words32<uint32_t, gx::K_UNSIGNED>(keys, n, descending, out_indices, tmp, tmp_bytes, stream);
    }
    return gx::sort::dispatch<true>(dtype, keys, nullptr, nullptr, out_indices, n, descending, true, tmp,
                                    tmp_bytes, stream);
  }
  switch (gx_dtype_size(dtype)) {
    case 1: return gx::sort::sorted_order_nullable<uint8_t>(dtype, keys, valid, n, null_count, descending, nulls_before, out_indices, tmp, tmp_bytes, stream);
    case 2: return gx::sort::sorted_order_nullable<uint16_t>(dtype, keys, valid, n, null_count, descending, nulls_before, out_indices, tmp, tmp_bytes, stream);
    case 4: return gx::sort::sorted_order_nullable<uint32_t>(dtype, keys, valid, n, null_count, descending, nulls_before, out_indices, tmp, tmp_bytes, stream);
    case 8: return gx::sort::sorted_order_nullable<uint64_t>(dtype, keys, valid, n, null_count, descending, nulls_before, out_indices, tmp, tmp_bytes, stream);
    default: return GX_EDTYPE;
  }
}

void gx_sort_set_algorithm(int algo)
{
  // algo 1 + 16*m: three-kernel passes with tile order m (measurement only: 1 plain blockIdx, 2 ticket)
  gx::sort::g_order_mode = (algo >> 4) & 3;
  algo &= 15;
  gx::sort::g_algorithm = (algo >= 0 && algo <= 2) ? algo : 0;
}

int gx_sort_profile(int enable)
{
  for (auto& p : gx::sort::g_profs) {
    if (enable && !p.created) {
      for (auto& e : p.ev) GX_HIP_TRY(hipEventCreate(&e));
      for (auto& e : p.hev) GX_HIP_TRY(hipEventCreate(&e));
      p.created = true;
    }
    p.enabled = enable != 0;
    p.level   = enable == 2 ? 2 : 1;
  }
  return 0;
}

int gx_sort_profile_slot(int slot)
{
  if (slot < 0 || slot >= gx::sort::PROF_SLOTS) return GX_EINVAL;
  gx::sort::g_prof_slot = slot;
  return 0;
}

int gx_sort_profile_read(float* hist_ms, float* pass_ms, int* npass)
{
  auto& p = gx::sort::g_profs[gx::sort::g_prof_slot];
  if (!p.created || !hist_ms || !pass_ms || !npass) return GX_EINVAL;
  if (p.level == 2) {  // (none of these events was recorded)
    *npass   = 0;
    *hist_ms = 0.0f;
    return 0;
  }
  *npass = p.npass;
  GX_HIP_TRY(hipEventSynchronize(p.ev[3 + 2 * (p.npass - 1)]));
  GX_HIP_TRY(hipEventElapsedTime(hist_ms, p.ev[0], p.ev[1]));
  for (int i = 0; i < p.npass; ++i) GX_HIP_TRY(hipEventElapsedTime(&pass_ms[i], p.ev[2 + 2 * i], p.ev[3 + 2 * i]));
  return 0;
}

int gx_sort_profile_read_hybrid(float* ms4)
{
  auto& p = gx::sort::g_profs[gx::sort::g_prof_slot];
  if (!p.created || !ms4) return GX_EINVAL;
  if (!p.hybrid_marked) return GX_EINVAL;
  if (p.level == 2) {  // the first partition level only
    GX_HIP_TRY(hipEventSynchronize(p.hev[1]));
    GX_HIP_TRY(hipEventElapsedTime(&ms4[0], p.hev[0], p.hev[1]));
    ms4[1] = ms4[2] = ms4[3] = 0.0f;
    return 0;
  }
  GX_HIP_TRY(hipEventSynchronize(p.hev[4]));
  for (int i = 0; i < 4; ++i) GX_HIP_TRY(hipEventElapsedTime(&ms4[i], p.hev[i], p.hev[i + 1]));
  return 0;
}

void gx_sort_set_hybrid(int enable) { gx::sort::g_hybrid = enable ? 1 : 0; }

void gx_sort_set_experiment(int bits) { gx::sort::g_exp = bits & 0x3C; }

void gx_sort_set_place_grid(int workgroups) { gx::sort::g_place_grid = workgroups > 0 ? workgroups : 0; }

void gx_sort_set_order_words(int enable) { gx::sort::g_order_words = enable ? 1 : 0; }

void gx_sort_set_counting(int enable) { gx::sort::g_counting = enable ? 1 : 0; }
void gx_sort_set_splitters(int enable) { gx::sort::g_split = enable ? 1 : 0; }
void gx_sort_set_float_cursor(int enable) { gx::sort::g_float_cursor = enable ? 1 : 0; }
void gx_sort_set_spin_limit_ms(int ms) { gx::sort::g_spin_ms = ms > 0 ? ms : 0; }
void gx_sort_set_fault_mode(int soft) { gx::sort::g_soft_fault = soft ? 1 : 0; }
void gx_sort_inject_lost_tile(long long tile) { gx::sort::g_inject_tile = tile >= 0 ? tile : -1; }
int gx_sort_split_info(const void* tmp, int32_t* info4_host, gx_stream_t stream)
{
  if (!tmp || !info4_host) return GX_EINVAL;
  const auto* plan = static_cast<const gx::sort::SortPlan*>(tmp);
  int32_t on = 0, bits2 = 0;
  uint32_t nsp_neq[2] = {0, 0};
  GX_HIP_TRY(hipMemcpyAsync(&on, &plan->sp.on, sizeof(on), hipMemcpyDeviceToHost, (hipStream_t)stream));
  GX_HIP_TRY(hipMemcpyAsync(nsp_neq, &plan->sp.nsp, sizeof(nsp_neq), hipMemcpyDeviceToHost, (hipStream_t)stream));
  GX_HIP_TRY(hipMemcpyAsync(&bits2, &plan->hy.bits2, sizeof(bits2), hipMemcpyDeviceToHost, (hipStream_t)stream));
  GX_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  info4_host[0] = on;
  info4_host[1] = (int32_t)nsp_neq[0];
  info4_host[2] = (int32_t)nsp_neq[1];
  info4_host[3] = bits2;
  return 0;
}
void gx_sort_set_cursor_path(int enable, float margin_sigmas)
{
  gx::sort::g_cursor        = enable ? 1 : 0;
  gx::sort::g_cursor_margin = margin_sigmas == 0.0f ? 8.0f : margin_sigmas;
}

int gx_sort_cursor_state(const void* tmp, int32_t* state_host, gx_stream_t stream)
{
  if (!tmp || !state_host) return GX_EINVAL;
  const auto* plan = static_cast<const gx::sort::SortPlan*>(tmp);
  GX_HIP_TRY(hipMemcpyAsync(state_host, &plan->hf.state, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  GX_HIP_TRY(hipStreamSynchronize(stream));
  return 0;
}

void gx_sort_set_lookback(int window) { gx::sort::g_lbw = (window == 4 || window == 8) ? window : 16; }

void gx_sort_set_cell(int keys) { gx::sort::g_cell = (keys == 8192 || keys == 16384) ? keys : 0; }

int gx_sort_info(const void* tmp, int32_t* info8_host, gx_stream_t stream)
{
  if (!tmp || !info8_host) return GX_EINVAL;
  const auto* plan = static_cast<const gx::sort::SortPlan*>(tmp);
  // attempt, ok, shift0, shift2, bits2, nlocal are the first six int32 of HybridPlan
  GX_HIP_TRY(hipMemcpyAsync(info8_host, &plan->hy, 6 * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  GX_HIP_TRY(hipMemcpyAsync(info8_host + 6, &plan->hy.max_cell, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  GX_HIP_TRY(hipMemcpyAsync(info8_host + 7, &plan->num_active, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  GX_HIP_TRY(hipStreamSynchronize(stream));
  return 0;
}

int gx_sort_big_info(const void* tmp, int64_t* info3_host, gx_stream_t stream)
{
  if (!tmp || !info3_host) return GX_EINVAL;
  const auto* plan = static_cast<const gx::sort::SortPlan*>(tmp);
  int32_t mode = 0;
  uint32_t nbig = 0;
  unsigned long long xn = 0;
  GX_HIP_TRY(hipMemcpyAsync(&mode, &plan->hy.lsd_mode, sizeof(mode), hipMemcpyDeviceToHost, stream));
  GX_HIP_TRY(hipMemcpyAsync(&nbig, &plan->hy.nbig, sizeof(nbig), hipMemcpyDeviceToHost, stream));
  GX_HIP_TRY(hipMemcpyAsync(&xn, &plan->hy.lsd_n, sizeof(xn), hipMemcpyDeviceToHost, stream));
  GX_HIP_TRY(hipStreamSynchronize(stream));
  info3_host[0] = mode;
  info3_host[1] = nbig;
  info3_host[2] = (int64_t)xn;
  return 0;
}

int gx_sort_place_info(const void* tmp, int32_t* todo_cells_host, gx_stream_t stream)
{
  if (!tmp || !todo_cells_host) return GX_EINVAL;
  const auto* plan = static_cast<const gx::sort::SortPlan*>(tmp);
  GX_HIP_TRY(hipMemcpyAsync(todo_cells_host, &plan->hy.todo_count, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  GX_HIP_TRY(hipStreamSynchronize(stream));
  return 0;
}

size_t gx_sort_plan_bytes(void) { return sizeof(gx::sort::SortPlan); }

int gx_sort_status(const void* tmp, int* status_host, gx_stream_t stream)
{
  if (!tmp || !status_host) return GX_EINVAL;
  const auto* plan = static_cast<const gx::sort::SortPlan*>(tmp);
  GX_HIP_TRY(hipMemcpyAsync(status_host, &plan->status, sizeof(int), hipMemcpyDeviceToHost, stream));
  GX_HIP_TRY(hipStreamSynchronize(stream));
  return 0;
}

int gx_sort_status_async(const void* tmp, int* status_host_pinned, gx_stream_t stream)
{
  if (!tmp || !status_host_pinned) return GX_EINVAL;
  const auto* plan = static_cast<const gx::sort::SortPlan*>(tmp);
  GX_HIP_TRY(hipMemcpyAsync(status_host_pinned, &plan->status, sizeof(int), hipMemcpyDeviceToHost, stream));
  return 0;
}

}  // extern "C"
