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
#include "gx_scan.hpp"

namespace gx {
namespace sort {

constexpr int BINS       = 256;
constexpr int MAX_PASSES = 8;
constexpr int BT         = 512;  // threads per workgroup (8 waves)
constexpr int NW         = BT / GX_WAVE;

// Device-resident plan, first bytes of the caller's scratch.
struct SortPlan {
  uint32_t hist[MAX_PASSES][BINS];  // digit histograms of the whole column
  uint32_t gbin[MAX_PASSES][BINS];  // exclusive scan of hist: global base of each bin
  int32_t pass_skip[MAX_PASSES];
  int32_t pass_src[MAX_PASSES];  // buffer selector: 0 = input, 1 = output (A), 2 = scratch (B)
  int32_t pass_dst[MAX_PASSES];
  uint32_t tickets[MAX_PASSES];
  uint32_t pool_next[MAX_PASSES][8];  // persistent kernel: takes per XCD pool
  int32_t num_active;
  int32_t status;  // 0 ok, 1 look-back spin timed out
};

// ------------------------------------------------------------------------------------------
// up-front histogram of every digit
// ------------------------------------------------------------------------------------------
template <typename KeyT, int KIND>
__global__ void __launch_bounds__(BT) k_hist_all(const KeyT* __restrict__ in, int64_t n, KeyT desc_mask,
                                                 SortPlan* plan)
{
  constexpr int NPASS = sizeof(KeyT);
  __shared__ uint32_t s_hist[NPASS * BINS];
  for (int i = threadIdx.x; i < NPASS * BINS; i += BT) s_hist[i] = 0;
  __syncthreads();
  const unsigned lane  = lane_id();
  constexpr int UNROLL = 4;  // independent loads in flight per lane (>= 32 KiB per CU at full occupancy)
  const int64_t stride = (int64_t)gridDim.x * BT * UNROLL;
  for (int64_t i0 = (int64_t)blockIdx.x * BT * UNROLL + threadIdx.x; i0 < n; i0 += stride) {
    KeyT raw[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = i0 + (int64_t)u * BT;
      raw[u]          = (i < n) ? in[i] : KeyT(0);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = i0 + (int64_t)u * BT;
      if (i < n) {
        const KeyT k          = to_sortable<KeyT, KIND>(raw[u], desc_mask);
        const uint64_t active = ballot(true);
        const int leader      = __builtin_ctzll(active);
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
          const uint32_t d  = (uint32_t)(k >> (8 * p)) & 0xFFu;
          const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
          if (ballot(d == d0) == active) {  // whole wave hits one bin: one add instead of 64 conflicts
            if ((int)lane == leader) atomicAdd(&s_hist[p * BINS + d0], (uint32_t)__builtin_popcountll(active));
          } else {
            atomicAdd(&s_hist[p * BINS + d], 1u);
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NPASS * BINS; i += BT) {
    const uint32_t c = s_hist[i];
    if (c) atomicAdd(&plan->hist[i / BINS][i % BINS], c);
  }
}

// one block of 256 threads
__global__ void __launch_bounds__(BINS) k_plan(SortPlan* plan, int npass, int64_t n)
{
  __shared__ uint32_t s_tmp[BINS / GX_WAVE + 1];
  __shared__ int s_skip[MAX_PASSES];
  const int t = threadIdx.x;
  for (int p = 0; p < npass; ++p) {
    const uint32_t c = plan->hist[p][t];
    const int triv   = __syncthreads_or(c == (uint32_t)n);
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
  uint32_t* vbuf[3];  // vbuf[0] may be null: iota
  SortPlan* plan;
  unsigned long long* status;  // [ntiles][256] look-back granules (algorithm 0)
  const uint32_t* tile_off;    // [256][ntiles] absolute offsets (algorithm 1)
  uint32_t* batch_base;        // [MAX_PASSES][8][nbatch] first ticket of a pool batch, +1 (0 = unpublished)
  int64_t nbatch;
  int64_t n;
  int64_t ntiles;
  int pass;
  int order_mode;  // experiment knob for the LBW == 0 kernel: 0 XCD-swizzled blockIdx, 1 plain blockIdx, 2 ticket
  uint64_t desc_mask;
};

constexpr uint32_t SPIN_LIMIT = 1u << 22;  // ~seconds; only a broken forward-progress chain gets here

__device__ __forceinline__ unsigned long long pack_status(unsigned flag, unsigned epoch, uint32_t value)
{
  return ((unsigned long long)flag << 62) | ((unsigned long long)epoch << 32) | value;
}

// XCD id of the executing CU (HW_REG_XCC_ID, bits 3:0).  Used for L2 affinity only.
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u; }

typedef __attribute__((address_space(1))) unsigned int gu32_t;
__device__ __forceinline__ void store_agent_u32(uint32_t* p, uint32_t v)
{
  __hip_atomic_store((gu32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t load_agent_u32(const uint32_t* p)
{
  return __hip_atomic_load((gu32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int POOL_BATCH = 8;  // consecutive tickets an XCD pool reserves at a time

// Tile hand-out of the persistent kernel.  Tickets are still a single global sequence (so the
// look-back's forward-progress argument is the usual one: a tile is owned by a running workgroup
// before any later ticket is), but they are drawn in batches of POOL_BATCH by per-XCD pools:
// the k-th take of pool x (atomicAdd on pool_next[x]) is ticket base[x][k / B] + k % B, where the
// take with k % B == 0 reserves the batch from the global counter -- after the previous batch of the
// same pool has been published, so that a pool hands out increasing tickets -- and publishes its
// base.  Neighbouring tiles therefore run on the same XCD at about the same time, and the partial
// 128-B lines where their output runs meet are merged in that XCD's L2 instead of leaving two
// non-coherent L2s as masked partial writes (measured on the scatter alone: 3.5 ms per 1e9-key pass
// with XCD-contiguous tiles, 4.0 ms with arrival-order tickets, 5.7 ms with tiles dealt round-robin).
// A take never waits on tile processing: the first take of a batch reserves and publishes at once
// (it waits only for the previous batch's publication, which is equally prompt), and the others
// read the published base when they need the ticket.  A workgroup resolves its pending take after
// it has finished its current tile, so no workgroup ever holds an unprocessed tile while it waits.
__device__ __forceinline__ uint32_t pool_take(const PassArgs& a, SortPlan* plan, int pass, unsigned xcc)
{
  const uint32_t k = atomicAdd(&plan->pool_next[pass][xcc], 1u);
  if (k % POOL_BATCH == 0) {
    const uint32_t j = k / POOL_BATCH;
    uint32_t* slot   = a.batch_base + ((int64_t)pass * 8 + xcc) * a.nbatch + j;
    uint32_t spins   = 0;
    if (j > 0) {
      while (load_agent_u32(slot - 1) == 0u) {
        if (++spins > SPIN_LIMIT) { atomicExch(&plan->status, 2); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    store_agent_u32(slot, atomicAdd(&plan->tickets[pass], (uint32_t)POOL_BATCH) + 1u);
  }
  return k;
}
// the ticket of take k, or -1 when the sequence is exhausted (tickets only grow: once past the
// end, every later take of every pool is too)
__device__ __forceinline__ int64_t pool_resolve(const PassArgs& a, SortPlan* plan, int pass, unsigned xcc, uint32_t k)
{
  const uint32_t j = k / POOL_BATCH, wi = k % POOL_BATCH;
  const uint32_t* slot = a.batch_base + ((int64_t)pass * 8 + xcc) * a.nbatch + j;
  uint32_t base1, spins = 0;
  while ((base1 = load_agent_u32(slot)) == 0u) {
    if (++spins > SPIN_LIMIT) { atomicExch(&plan->status, 2); return -1; }
    __builtin_amdgcn_s_sleep(1);
  }
  const int64_t t = (int64_t)(base1 - 1u) + wi;
  return t < a.ntiles ? t : -1;
}

// LBW: look-back window (0 = no look-back: offsets were precomputed by algorithm 1).  A thread owns
// one bin and inspects LBW predecessor tiles per round with LBW independent loads in flight.
// PERSIST: workgroups loop over tiles handed out by pool_take/pool_resolve (2 workgroups per CU).
template <typename KeyT, int KIND, bool HAS_VAL, int KPT, int LBW, bool PERSIST>
__global__ void __launch_bounds__(BT, 4) k_radix_pass(PassArgs a)
{
  constexpr bool LOOKBACK = LBW > 0;
  static_assert(!PERSIST || LOOKBACK, "the persistent kernel is the look-back kernel");
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
  const int src_sel      = plan->pass_src[pass];
  const int dst_sel      = plan->pass_dst[pass];
  const KeyT* kin        = static_cast<const KeyT*>(a.kbuf[src_sel]);
  KeyT* kout             = static_cast<KeyT*>(a.kbuf[dst_sel]);
  const uint32_t* vin    = HAS_VAL ? a.vbuf[src_sel] : nullptr;
  uint32_t* vout         = HAS_VAL ? a.vbuf[dst_sel] : nullptr;
  const KeyT desc_mask   = (KeyT)a.desc_mask;
  const int shift        = pass * 8;
  const unsigned tid     = threadIdx.x;
  const unsigned lane    = lane_id();
  const unsigned w       = tid / GX_WAVE;
  const unsigned epoch   = (unsigned)pass + 1u;

  const unsigned xcc = PERSIST ? xcc_id() : 0u;
  uint32_t k_next    = 0;  // thread 0: pending pool take (the tile after the current one)
  int64_t tile;
  if (PERSIST) {
    if (tid == 0) {
      const uint32_t k0                 = pool_take(a, plan, pass, xcc);
      reinterpret_cast<int*>(s_misc)[0] = (int)pool_resolve(a, plan, pass, xcc, k0);
    }
    __syncthreads();
    tile = reinterpret_cast<int*>(s_misc)[0];
  } else if (LOOKBACK || a.order_mode == 2) {
    if (tid == 0) s_misc[0] = atomicAdd(&plan->tickets[pass], 1u);
    __syncthreads();
    tile = s_misc[0];
  } else {
    tile = a.order_mode == 1 ? (int64_t)blockIdx.x : xcd_swizzle(blockIdx.x, gridDim.x);
  }
  while (tile >= 0) {
  // the take for the next tile is issued now (its round trip overlaps this tile's key loads) and
  // resolved after this tile has been written out
  if (PERSIST && tid == 0) k_next = pool_take(a, plan, pass, xcc);
  const int64_t base = tile * TILE;
  const int nvalid   = (int)((a.n - base < (int64_t)TILE) ? (a.n - base) : (int64_t)TILE);

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
  const uint64_t lt = lanemask_lt();
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int idx = wbase + j * GX_WAVE;
    uint32_t d    = (uint32_t)(to_sortable<KeyT, KIND>(key[j], desc_mask) >> shift) & 0xFFu;
    if (idx >= nvalid) d = BINS - 1;  // padding sorts last (it is also last in input order)
    uint64_t m = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit   = (d >> b) & 1u;
      const uint64_t v = ballot(bit);
      m &= bit ? v : ~v;
    }
    const uint32_t lower = (uint32_t)__builtin_popcountll(m & lt);
    const uint32_t prev  = my_hist[d];
    if (lower == 0) my_hist[d] = prev + (uint32_t)__builtin_popcountll(m);
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
    store_agent_u64(&a.status[tile * BINS + tid], pack_status(tile == 0 ? 2u : 1u, epoch, pub_count));
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
              while ((x >> 62) == 0 || ((unsigned)(x >> 32) & 0xFFu) != epoch) {
                if (++spins > SPIN_LIMIT) {
                  atomicExch(&plan->status, 1);
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
        store_agent_u64(&a.status[tile * BINS + tid], pack_status(2u, epoch, prefix + pub_count));
      }
      gbase = plan->gbin[pass][tid] + prefix;
    } else {
      gbase = a.tile_off[(int64_t)tid * a.ntiles + tile];
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
  if (!PERSIST) break;
  __syncthreads();  // every read of this tile's LDS state is done
  if (tid == 0) reinterpret_cast<int*>(s_misc)[0] = (int)pool_resolve(a, plan, pass, xcc, k_next);
  __syncthreads();
  tile = reinterpret_cast<int*>(s_misc)[0];
  }  // while (tile >= 0)
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
  const KeyT* kin      = static_cast<const KeyT*>(a.kbuf[0]);
  KeyT* kout           = static_cast<KeyT*>(a.kbuf[1]);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
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

static int g_algorithm = 0;
static int g_order_mode = 0;
static int g_persist_blocks = 512;  // persistent look-back kernel: 2 workgroups x 256 CUs

// optional per-launch timing with HIP events on the caller's stream (bench.py's roofline leg)
struct Profile {
  bool enabled = false;
  bool created = false;
  int npass    = 0;
  hipEvent_t ev[2 * MAX_PASSES + 2];
};
static Profile g_prof;
static inline void prof_mark(int idx, hipStream_t s)
{
  if (g_prof.enabled) (void)hipEventRecord(g_prof.ev[idx], s);
}

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

template <typename KeyT, int KIND, bool HAS_VAL>
int sort_impl(const void* keys_in, void* keys_out, const int32_t* vals_in, int32_t* vals_out, int64_t n,
              int descending, bool radix_nan_rule, void* tmp, size_t* tmp_bytes, hipStream_t stream)
{
  constexpr int KPT   = kpt_for<KeyT>(HAS_VAL);
  constexpr int TILE  = BT * KPT;
  constexpr int NPASS = sizeof(KeyT);
  if (n < 0 || tmp_bytes == nullptr) return GX_EINVAL;
  const int64_t ntiles = n > 0 ? div_up(n, TILE) : 0;
  const int algo       = g_algorithm;

  Carver c(tmp);
  SortPlan* plan             = c.take<SortPlan>(1);
  unsigned long long* status = nullptr;
  uint32_t* tile_hist        = nullptr;
  uint32_t* partials         = nullptr;
  const int64_t pblocks = ntiles < g_persist_blocks ? ntiles : (int64_t)g_persist_blocks;
  const int64_t nbatch  = (ntiles + 2 * pblocks) / POOL_BATCH + 8;
  uint32_t* batch_base  = nullptr;
  if (algo != 1) {
    status     = c.take<unsigned long long>((size_t)ntiles * BINS);
    batch_base = c.take<uint32_t>((size_t)MAX_PASSES * 8 * nbatch);  // right behind status: one memset
  } else {
    tile_hist = c.take<uint32_t>((size_t)ntiles * BINS);
    partials  = c.take<uint32_t>(scan::partials_count(ntiles * BINS));
  }
  KeyT* kb_scratch = c.take<KeyT>((size_t)n);
  KeyT* ka_scratch = keys_out ? nullptr : c.take<KeyT>((size_t)n);
  uint32_t* vb     = HAS_VAL ? c.take<uint32_t>((size_t)n) : nullptr;
  if (tmp == nullptr) {
    *tmp_bytes = c.total();
    return 0;
  }
  if (*tmp_bytes < c.total()) return GX_ETMP;
  if (n > 0 && (keys_in == nullptr || (HAS_VAL && vals_out == nullptr))) return GX_EINVAL;
  if (n > 0 && keys_in == keys_out) return GX_EINVAL;

  GX_HIP_TRY(hipMemsetAsync(plan, 0, sizeof(SortPlan), stream));
  if (n == 0) return 0;
  if (algo != 1)
    GX_HIP_TRY(hipMemsetAsync(status, 0, (size_t)(reinterpret_cast<char*>(batch_base + (size_t)MAX_PASSES * 8 * nbatch) -
                                                   reinterpret_cast<char*>(status)), stream));

  const KeyT desc_mask = descending ? KeyT(~KeyT(0)) : KeyT(0);
  g_prof.npass = NPASS;
  prof_mark(0, stream);
  {
    int64_t blocks = div_up(n, (int64_t)BT * 8);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((k_hist_all<KeyT, KIND>), dim3((unsigned)blocks), dim3(BT), 0, stream,
                       static_cast<const KeyT*>(keys_in), n, desc_mask, plan);
    hipLaunchKernelGGL(k_plan, dim3(1), dim3(BINS), 0, stream, plan, NPASS, n);
  }
  prof_mark(1, stream);

  PassArgs a;
  a.kbuf[0]   = const_cast<void*>(keys_in);
  a.kbuf[1]   = keys_out ? keys_out : static_cast<void*>(ka_scratch);
  a.kbuf[2]   = kb_scratch;
  a.vbuf[0]   = reinterpret_cast<uint32_t*>(const_cast<int32_t*>(vals_in));
  a.vbuf[1]   = reinterpret_cast<uint32_t*>(vals_out);
  a.vbuf[2]   = vb;
  a.plan      = plan;
  a.status    = status;
  a.tile_off   = tile_hist;
  a.batch_base = batch_base;
  a.nbatch     = nbatch;
  a.n         = n;
  a.ntiles    = ntiles;
  a.desc_mask = (uint64_t)desc_mask;
  a.order_mode = g_order_mode;

  constexpr size_t lds = pass_lds_bytes<KeyT, HAS_VAL, KPT>();
  auto kern_lb         = (algo == 2) ? k_radix_pass<KeyT, KIND, HAS_VAL, KPT, 4, false> : k_radix_pass<KeyT, KIND, HAS_VAL, KPT, 4, true>;
  auto kern_pre        = k_radix_pass<KeyT, KIND, HAS_VAL, KPT, 0, false>;
  static bool attr_set = false;  // per template instantiation
  if (!attr_set) {
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_radix_pass<KeyT, KIND, HAS_VAL, KPT, 4, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_radix_pass<KeyT, KIND, HAS_VAL, KPT, 4, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern_pre),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  for (int pass = 0; pass < NPASS; ++pass) {
    a.pass = pass;
    prof_mark(2 + 2 * pass, stream);
    if (algo != 1) {
      hipLaunchKernelGGL(kern_lb, dim3((unsigned)(algo == 2 ? ntiles : pblocks)), dim3(BT), lds, stream, a);
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
    if (KIND == K_FLOAT && descending && radix_nan_rule && sizeof(KeyT) >= 4) {
      hipLaunchKernelGGL((k_reverse_nan_block<KeyT, HAS_VAL>), dim3(256), dim3(256), 0, stream,
                         static_cast<KeyT*>(a.kbuf[1]), a.vbuf[1], n);
    }
  }
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

}  // namespace sort
}  // namespace gx

extern "C" {

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

int gx_sorted_order(int dtype, const void* keys, const uint32_t* valid, int64_t n, int64_t null_count,
                    int descending, int nulls_before, int32_t* out_indices, void* tmp, size_t* tmp_bytes,
                    gx_stream_t stream)
{
  if (n < 0 || null_count < 0 || null_count > n || tmp_bytes == nullptr) return GX_EINVAL;
  if (valid == nullptr || null_count == 0) {
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
  auto& p = gx::sort::g_prof;
  if (enable && !p.created) {
    for (auto& e : p.ev) GX_HIP_TRY(hipEventCreate(&e));
    p.created = true;
  }
  p.enabled = enable != 0;
  return 0;
}

int gx_sort_profile_read(float* hist_ms, float* pass_ms, int* npass)
{
  auto& p = gx::sort::g_prof;
  if (!p.created || !hist_ms || !pass_ms || !npass) return GX_EINVAL;
  *npass = p.npass;
  GX_HIP_TRY(hipEventSynchronize(p.ev[3 + 2 * (p.npass - 1)]));
  GX_HIP_TRY(hipEventElapsedTime(hist_ms, p.ev[0], p.ev[1]));
  for (int i = 0; i < p.npass; ++i) GX_HIP_TRY(hipEventElapsedTime(&pass_ms[i], p.ev[2 + 2 * i], p.ev[3 + 2 * i]));
  return 0;
}

int gx_sort_status(const void* tmp, int* status_host, gx_stream_t stream)
{
  if (!tmp || !status_host) return GX_EINVAL;
  const auto* plan = static_cast<const gx::sort::SortPlan*>(tmp);
  GX_HIP_TRY(hipMemcpyAsync(status_host, &plan->status, sizeof(int), hipMemcpyDeviceToHost, stream));
  GX_HIP_TRY(hipStreamSynchronize(stream));
  return 0;
}

}  // extern "C"
