// xp_splitter_level0.hip -- micro-benchmark for DESIGN.md section 8 row 1 (the splitter sort's level 0).  Standalone: no dependency
// on the library, NOT part of the product build.
//
// STATUS (end of round 4): ran once, with the round's last GPU seconds (profiles/r4_run31_xp_splitter_level0.txt): correct on the
// first execution (no key outside its bucket's range, histogram totals = n); plan 0.16-0.19 ms, histogram through the search
// 6.6 ms, search + scatter 7.6-7.8 ms per 1e9 keys.  Runs 32 / 33 (profiles/r4_run32_xp_splitter_level0.txt): SEARCH 1 -- a LUT of
// 2048 key ranges + a scan of the sorted table -- histograms in 3.97 ms; the scatter below stays at 7.4 ms with either search: it
// is bound by ITS placement (one cursor per bucket, all XCDs into all regions), not by the search.  Next: put the LUT search into
// the product's k_hf_scatter<.., 0, ..> (per-(range, bin) slots) behind the plan's "uneven buckets" verdict.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/xp/xp_splitter_level0.hip -o /tmp/xp_split && /tmp/xp_split [n]
// The CPU model of the same plan (bucket balance, equality buckets, cells above capacity on eleven distributions) is
// scripts/xp/xp_splitter_model.py -> profiles/r4_model_splitter_sort.txt.
//
// What it measures, for n keys (default 1e9) of a chosen distribution (0 uniform, 1 bell-shaped around zero, 2 Zipf-like):
//   1. k_sample + k_sort_sample: 16384 sampled keys sorted by ONE workgroup in LDS (bitonic, 128 KiB), the 255 quantiles turned
//      into a splitter table with EQUALITY buckets [v, v + 1) for a value that fills two quantiles -- cost of planning;
//   2. k_level0<false>: histogram of bucket = upper_bound(splitters, key) by a branch-free search over an Eytzinger table in LDS
//      (9 steps) -- what the SEARCH costs on top of reading the keys;
//   3. k_level0<true>: the same search + LDS reorder + one returning atomic per (tile, bucket) on a cursor + coalesced write-out
//      into exact bucket regions: the proposed level 0, to be compared with k_hf_scatter<.., 0, 8>'s 3.6 ms per 1e9 keys;
//   4. k_check: every key of every region lies in [lo_b, hi_b) (violations must be 0); bucket balance and the equality share.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIP_TRY(x)                                                                                  \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) {                                                                         \
      std::fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));      \
      std::exit(1);                                                                                 \
    }                                                                                               \
  } while (0)

constexpr int WAVE   = 64;
constexpr int BT     = 512;        // threads per workgroup of the level-0 kernels
constexpr int KPT    = 16;         // keys per thread: 8192-key tiles, 64 KiB
constexpr int TILE   = BT * KPT;
constexpr int NSAMP  = 16384;      // sample size: sorted by one workgroup
constexpr int NLEAF  = 512;        // Eytzinger tree with 511 inner nodes -> up to 512 buckets
constexpr uint64_t U64MAX = ~0ull;
constexpr int NLUT   = 2048;       // SEARCH 1: first bucket of 2048 key ranges (uint16 each), then a short scan of the sorted table

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// sortable (unsigned, order-preserving) form of an int64 key
__device__ __forceinline__ uint64_t sortable(int64_t v) { return (uint64_t)v ^ (1ull << 63); }

__global__ void k_fill(uint64_t* keys, int64_t n, int dist, uint64_t seed)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t r = mix64((uint64_t)i * 0xD1342543DE82EF95ull + seed);
    int64_t v;
    if (dist == 0) {
      v = (int64_t)r;
    } else if (dist == 1) {  // bell-shaped around zero: sum of 12 uniforms (Irwin-Hall), sigma = 2^40
      double s = 0.0;
      uint64_t x = r;
      for (int k = 0; k < 12; ++k) {
        x = mix64(x);
        s += (double)(x >> 11) * (1.0 / 9007199254740992.0);
      }
      v = (int64_t)__builtin_round((s - 6.0) * 1099511627776.0);
    } else {  // Zipf-like: floor(u^-5), clipped to 2^31
      double u = (double)(r >> 11) * (1.0 / 9007199254740992.0);
      if (u < 1.1102230246251565e-16) u = 1.1102230246251565e-16;
      double z = 1.0 / (u * u * u * u * u);
      v = (int64_t)(z < 2147483648.0 ? z : 2147483648.0);
    }
    keys[i] = sortable(v);
  }
}

__global__ void k_sample(const uint64_t* __restrict__ keys, int64_t n, uint64_t* __restrict__ samp)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < NSAMP) samp[i] = keys[(int64_t)(((__int128)i * n) / NSAMP)];
}

// ONE workgroup of 1024 threads: bitonic sort of the 16384 sampled keys in LDS, then the splitter table.
// table[0 .. nsp) sorted distinct splitters, padded with U64MAX up to NLEAF - 1 entries; eyt[1 .. NLEAF) the same in BFS order.
struct Plan {
  uint32_t nsp;                 // splitters in use (<= 510); buckets = nsp + 1
  uint32_t pad;
  uint64_t table[NLEAF];        // sorted; table[NLEAF - 1] unused
  uint64_t eyt[NLEAF];          // eyt[0] unused
  unsigned long long hist[NLEAF];
  unsigned long long start[NLEAF + 1];
  unsigned int cursor[NLEAF];
  unsigned long long violations;
  // SEARCH 1 (run 32): lut[c] = number of splitters in cells below c, cells cut on rel = key - kmin either LINEARLY (rel >> lshift;
  // even or bell-shaped densities) or LOGARITHMICALLY (exponent and 5 mantissa bits of rel; power-law densities): the planner
  // takes the form whose fullest cell holds fewer splitters
  uint64_t kmin;
  uint32_t lut_log, lshift, lut_worst[2];
  uint16_t lut[NLUT];
};

__device__ __forceinline__ uint32_t lut_cell(uint64_t rel, uint32_t lut_log, uint32_t lshift)
{
  if (lut_log) {
    if (rel == 0) return 0;
    const int e = 63 - __clzll((long long)rel);
    const uint32_t m = e >= 5 ? (uint32_t)(rel >> (e - 5)) & 31u : (uint32_t)(rel << (5 - e)) & 31u;
    return (uint32_t)e * 32u + m;
  }
  const uint64_t c = rel >> lshift;
  return c < (uint64_t)(NLUT - 1) ? (uint32_t)c : (uint32_t)(NLUT - 1);
}

__global__ void __launch_bounds__(1024) k_sort_sample(uint64_t* __restrict__ samp, Plan* plan)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* s = reinterpret_cast<uint64_t*>(smem);  // [NSAMP]
  __shared__ uint32_t s_cnt[1024 / WAVE + 1];
  __shared__ uint32_t s_total;
  const int tid = threadIdx.x;
  for (int i = tid; i < NSAMP; i += 1024) s[i] = samp[i];
  __syncthreads();
  for (int k = 2; k <= NSAMP; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int p = tid; p < NSAMP / 2; p += 1024) {
        const int a = ((p & ~(j - 1)) << 1) | (p & (j - 1));  // index with bit j clear
        const int b = a | j;
        const bool up = (a & k) == 0;
        const uint64_t x = s[a], y = s[b];
        if ((x > y) == up) {
          s[a] = y;
          s[b] = x;
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < NSAMP; i += 1024) samp[i] = s[i];
  // quantile t (t = 0 .. 254) = s[(t + 1) * 64]; thread t emits q[t] when it starts a run of equal quantiles, and q[t] + 1 when it
  // ends a run of >= 2 (a value that holds more than 1/256 of the column: an EQUALITY bucket [v, v + 1)) unless the next distinct
  // quantile is v + 1 already
  uint64_t q = 0, qprev = 0, qnext = 0;
  bool first = false, eq_end = false;
  if (tid < 255) {
    q     = s[(tid + 1) * (NSAMP / 256)];
    qprev = tid > 0 ? s[tid * (NSAMP / 256)] : 0;
    qnext = tid < 254 ? s[(tid + 2) * (NSAMP / 256)] : 0;
    first = tid == 0 || q != qprev;
    const bool last = tid == 254 || q != qnext;
    const bool in_run = (tid > 0 && q == qprev) || (tid < 254 && q == qnext);
    eq_end = last && in_run && q != U64MAX && !(tid < 254 && qnext == q + 1);
  }
  const uint32_t mine = (first ? 1u : 0u) + (eq_end ? 1u : 0u);
  // block exclusive scan of `mine` over the first 256 threads (4 waves): wave scan by shuffles + wave totals in LDS
  uint32_t incl = mine;
  for (int d = 1; d < WAVE; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, WAVE);
    if ((tid & (WAVE - 1)) >= d) incl += o;
  }
  if ((tid & (WAVE - 1)) == WAVE - 1) s_cnt[tid / WAVE] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < tid / WAVE; ++w) base += s_cnt[w];
  const uint32_t excl = base + incl - mine;
  if (tid == 255) s_total = excl;  // (threads >= 255 contribute nothing: the exclusive prefix of thread 255 is the total)
  for (int i = tid; i < NLEAF; i += 1024) plan->table[i] = U64MAX;
  __syncthreads();
  if (tid < 255) {
    uint32_t o = excl;
    if (first) plan->table[o++] = q;
    if (eq_end) plan->table[o] = q + 1;
  }
  __syncthreads();
  if (tid == 0) plan->nsp = s_total;
  // BFS (Eytzinger) order: node k of level l (k in [2^l, 2^(l+1))), j = k - 2^l  ->  in-order rank ((2 j + 1) << (8 - l)) - 1
  if (tid >= 1 && tid < NLEAF) {
    const int l = 31 - __clz(tid);
    const int j = tid - (1 << l);
    plan->eyt[tid] = plan->table[(((2 * j + 1) << (8 - l)) - 1)];
  }
  // ---- SEARCH 1: the LUT.  kmin / kmax of the SAMPLE (keys outside still find their bucket: the scan is exact, the LUT only a start)
  __shared__ uint32_t s_idx[2][NLEAF];
  __shared__ uint32_t s_worst[2];
  const uint64_t kmin = s[0], kmax = s[NSAMP - 1];
  uint32_t lshift = 0;
  while (lshift < 63 && ((kmax - kmin) >> lshift) >= (uint64_t)NLUT) ++lshift;
  const uint32_t nsp = s_total;
  if (tid < 2) s_worst[tid] = 0;
  __syncthreads();
  for (int form = 0; form < 2; ++form) {
    for (uint32_t i = tid; i < NLEAF; i += 1024) s_idx[form][i] = i < nsp ? lut_cell(plan->table[i] - kmin, (uint32_t)form, lshift) : 0xFFFFFFFFu;
  }
  __syncthreads();
  uint32_t lo2[2][NLUT / 1024];
  for (int form = 0; form < 2; ++form) {
    for (int r = 0; r < NLUT / 1024; ++r) {
      const uint32_t c = (uint32_t)tid + 1024u * r;
      // splitters in cells below c = lower_bound(s_idx, c); in cell c: upper_bound - lower_bound
      uint32_t a = 0, b = nsp;
      while (a < b) { const uint32_t mid = (a + b) / 2; if (s_idx[form][mid] < c) a = mid + 1; else b = mid; }
      uint32_t a2 = a, b2 = nsp;
      while (a2 < b2) { const uint32_t mid = (a2 + b2) / 2; if (s_idx[form][mid] <= c) a2 = mid + 1; else b2 = mid; }
      lo2[form][r] = a | (a2 == a ? 0x8000u : 0u);  // bit 15: no splitter inside this cell -> the bucket is known without the table
      atomicMax(&s_worst[form], a2 - a);
    }
  }
  __syncthreads();
  const int pick = s_worst[1] < s_worst[0] ? 1 : 0;
  for (int r = 0; r < NLUT / 1024; ++r) plan->lut[tid + 1024 * r] = (uint16_t)lo2[pick][r];
  if (tid == 0) {
    plan->kmin = kmin;
    plan->lut_log = (uint32_t)pick;
    plan->lshift = lshift;
    plan->lut_worst[0] = s_worst[0];
    plan->lut_worst[1] = s_worst[1];
  }
}

// bucket = number of splitters <= key, by 9 branch-free steps over the BFS table in LDS (every level is contiguous: the lanes of a
// wave read at most 2^l neighbouring entries at step l)
__device__ __forceinline__ uint32_t bucket_of(const uint64_t* __restrict__ s_eyt, uint64_t key, uint32_t nsp)
{
  uint32_t k = 1;
#pragma unroll
  for (int l = 0; l < 9; ++l) k = 2 * k + (s_eyt[k] <= key ? 1u : 0u);
  const uint32_t b = k - NLEAF;  // keys equal to the U64MAX padding would pass it: clamp
  return b < nsp ? b : nsp;
}

// SEARCH 1: start at the LUT's bucket, walk the sorted table while its splitters are <= key (exact for every key)
__device__ __forceinline__ uint32_t bucket_lut(const uint64_t* __restrict__ s_tab, const uint16_t* __restrict__ s_lut, uint64_t key, uint32_t nsp,
                                               uint64_t kmin, uint32_t lut_log, uint32_t lshift)
{
  const uint64_t rel = key >= kmin ? key - kmin : 0;
  const uint32_t w   = s_lut[lut_cell(rel, lut_log, lshift)];
  uint32_t b         = w & 0x7FFFu;
  if (w & 0x8000u) return b;  // (run 33) a cell without a splitter: 88 % of the cells of a uniform column
  while (b < nsp && s_tab[b] <= key) ++b;
  return b;
}

template <bool SCATTER, int SEARCH>
__global__ void __launch_bounds__(BT, 4) k_level0(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int64_t n, Plan* plan)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* s_keys  = reinterpret_cast<uint64_t*>(smem);                                   // [TILE] (SCATTER)
  uint64_t* s_eyt   = reinterpret_cast<uint64_t*>(smem + (SCATTER ? (size_t)TILE * 8 : 0));  // [NLEAF]: BFS table (SEARCH 0) / sorted table (SEARCH 1)
  uint32_t* s_cnt   = reinterpret_cast<uint32_t*>(s_eyt + NLEAF);                          // [NLEAF] counts, then bin starts
  uint32_t* s_delta = s_cnt + NLEAF;                                                       // [NLEAF]
  uint16_t* s_lut   = reinterpret_cast<uint16_t*>(s_delta + NLEAF);                        // [NLUT] (SEARCH 1)
  __shared__ uint32_t s_wsum[BT / WAVE + 1];
  const unsigned tid = threadIdx.x;
  const uint32_t nsp = plan->nsp;
  const uint64_t kmin = plan->kmin;
  const uint32_t lut_log = plan->lut_log, lshift = plan->lshift;
  for (int i = tid; i < NLEAF; i += BT) {
    s_eyt[i] = SEARCH == 0 ? plan->eyt[i] : plan->table[i];
    s_cnt[i] = 0;
  }
  if (SEARCH == 1)
    for (int i = tid; i < NLUT; i += BT) s_lut[i] = plan->lut[i];
  const int64_t base = (int64_t)blockIdx.x * TILE;
  const int nvalid   = (int)(n - base < (int64_t)TILE ? n - base : (int64_t)TILE);
  uint64_t key[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int idx = j * BT + (int)tid;
    key[j]        = idx < nvalid ? in[base + idx] : U64MAX;
  }
  __syncthreads();
  uint32_t packed[KPT];  // bucket << 16 | rank inside (tile, bucket)
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const bool live  = j * BT + (int)tid < nvalid;
    const uint32_t b = SEARCH == 0 ? bucket_of(s_eyt, key[j], nsp) : bucket_lut(s_eyt, s_lut, key[j], nsp, kmin, lut_log, lshift);
    // (the product's lds_rank aggregates lanes that share a bucket; a plain returning atomic is enough for this measurement
    //  unless the column has heavy hitters -- distribution 2 will show that cost)
    const uint32_t r = live ? atomicAdd(&s_cnt[b], 1u) : 0u;
    packed[j]        = (b << 16) | r;
  }
  __syncthreads();
  if (!SCATTER) {
    for (int i = tid; i < NLEAF; i += BT)
      if (s_cnt[i]) atomicAdd(&plan->hist[i], (unsigned long long)s_cnt[i]);
    return;
  }
  // one returning atomic per non-empty bucket reserves the tile's run in the bucket's region; exclusive scan of the counts = bin
  // starts inside the tile (one bucket per thread: NLEAF == BT)
  const uint32_t c = s_cnt[tid];
  uint32_t g       = 0;
  if (c) g = atomicAdd(&plan->cursor[tid], c);
  uint32_t incl = c;
  for (int d = 1; d < WAVE; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, WAVE);
    if ((tid & (WAVE - 1)) >= d) incl += o;
  }
  if ((tid & (WAVE - 1)) == WAVE - 1) s_wsum[tid / WAVE] = incl;
  __syncthreads();
  uint32_t wbase = 0;
  for (unsigned w = 0; w < tid / WAVE; ++w) wbase += s_wsum[w];
  const uint32_t st = wbase + incl - c;
  s_cnt[tid]        = st;
  s_delta[tid]      = (uint32_t)plan->start[tid] + g - st;  // (bucket regions of < 2^32 keys: n < 2^32)
  __syncthreads();
#pragma unroll
  for (int j = 0; j < KPT; ++j)
    if (j * BT + (int)tid < nvalid) s_keys[s_cnt[packed[j] >> 16] + (packed[j] & 0xFFFFu)] = key[j];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int i = j * BT + (int)tid;
    if (i < nvalid) {
      const uint64_t k = s_keys[i];
      const uint32_t b = SEARCH == 0 ? bucket_of(s_eyt, k, nsp) : bucket_lut(s_eyt, s_lut, k, nsp, kmin, lut_log, lshift);
      out[s_delta[b] + (uint32_t)i] = k;
    }
  }
}

__global__ void k_starts(Plan* plan)
{
  if (threadIdx.x == 0) {
    unsigned long long run = 0;
    for (int b = 0; b < NLEAF; ++b) {
      plan->start[b]  = run;
      plan->cursor[b] = 0;
      run += plan->hist[b];
    }
    plan->start[NLEAF] = run;
  }
}

__global__ void k_check(const uint64_t* __restrict__ out, int64_t n, Plan* plan)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    // region of position i: the last bucket whose start is <= i
    int lo = 0, hi = NLEAF;
    while (hi - lo > 1) {
      const int mid = (lo + hi) / 2;
      if (plan->start[mid] <= (unsigned long long)i) lo = mid; else hi = mid;
    }
    const uint64_t k  = out[i];
    const uint64_t kl = lo == 0 ? 0 : plan->table[lo - 1];
    const bool below  = k < kl;
    const bool above  = (uint32_t)lo < plan->nsp && k >= plan->table[lo];
    if (below || above) ++bad;
  }
  if (bad) atomicAdd(&plan->violations, bad);
}

int main(int argc, char** argv)
{
  const int64_t n = argc > 1 ? (int64_t)std::atof(argv[1]) : 1000000000ll;
  if (n >= (1ll << 32)) {
    std::fprintf(stderr, "n must be below 2^32\n");
    return 1;
  }
  uint64_t *keys, *out, *samp;
  Plan* plan;
  HIP_TRY(hipMalloc(&keys, (size_t)n * 8));
  HIP_TRY(hipMalloc(&out, (size_t)n * 8));
  HIP_TRY(hipMalloc(&samp, (size_t)NSAMP * 8));
  HIP_TRY(hipMalloc(&plan, sizeof(Plan)));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort_sample), hipFuncAttributeMaxDynamicSharedMemorySize, NSAMP * 8));
  const size_t lds_hist = (size_t)NLEAF * 8 + 2 * NLEAF * 4 + NLUT * 2, lds_scat = (size_t)TILE * 8 + lds_hist;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_level0<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_scat));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_level0<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_scat));
  const unsigned tiles = (unsigned)((n + TILE - 1) / TILE);
  static const char* names[3] = {"uniform 64-bit", "bell-shaped around zero (sigma 2^40)", "Zipf-like floor(u^-5) <= 2^31"};
  for (int dist = 0; dist < 3; ++dist) {
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, keys, n, dist, 42ull);
    HIP_TRY(hipMemset(plan, 0, sizeof(Plan)));
    auto timed = [&](const char* what, auto&& launch, int reps) {
      launch();  // warm-up
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) launch();
      HIP_TRY(hipEventRecord(e1, 0));
      HIP_TRY(hipEventSynchronize(e1));
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
      std::printf("  %-52s %8.3f ms\n", what, ms / reps);
      return ms / reps;
    };
    std::printf("%s, n = %lld\n", names[dist], (long long)n);
    timed("sample + sort 16384 keys + splitter table", [&] {
      hipLaunchKernelGGL(k_sample, dim3(NSAMP / 256), dim3(256), 0, 0, keys, n, samp);
      hipLaunchKernelGGL(k_sort_sample, dim3(1), dim3(1024), NSAMP * 8, 0, samp, plan);
    }, 3);
    auto hist = [&] {
      HIP_TRY(hipMemsetAsync(plan->hist, 0, sizeof(plan->hist), 0));
      hipLaunchKernelGGL((k_level0<false, 0>), dim3(tiles), dim3(BT), lds_hist, 0, keys, out, n, plan);
    };
    const float t_hist = timed("histogram through the 9-step tree search (8 B/row)", hist, 3);
    auto hist1 = [&] {
      HIP_TRY(hipMemsetAsync(plan->hist, 0, sizeof(plan->hist), 0));
      hipLaunchKernelGGL((k_level0<false, 1>), dim3(tiles), dim3(BT), lds_hist, 0, keys, out, n, plan);
    };
    timed("histogram through LUT + scan (8 B/row)", hist1, 3);
    hipLaunchKernelGGL(k_starts, dim3(1), dim3(64), 0, 0, plan);
    auto scat = [&] {
      HIP_TRY(hipMemsetAsync(plan->cursor, 0, sizeof(plan->cursor), 0));
      hipLaunchKernelGGL((k_level0<true, 0>), dim3(tiles), dim3(BT), lds_scat, 0, keys, out, n, plan);
    };
    const float t_scat = timed("level 0, tree search + scatter (16 B/row)", scat, 3);
    auto scat1 = [&] {
      HIP_TRY(hipMemsetAsync(plan->cursor, 0, sizeof(plan->cursor), 0));
      hipLaunchKernelGGL((k_level0<true, 1>), dim3(tiles), dim3(BT), lds_scat, 0, keys, out, n, plan);
    };
    timed("level 0, LUT + scan + scatter (16 B/row)", scat1, 3);
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, out, n, plan);
    Plan h;
    HIP_TRY(hipMemcpy(&h, plan, sizeof(Plan), hipMemcpyDeviceToHost));
    unsigned long long mx = 0, eq = 0, total = 0;
    for (uint32_t b = 0; b <= h.nsp; ++b) {
      total += h.hist[b];
      const bool eqb = b > 0 && b <= h.nsp - 0 && b < h.nsp && h.table[b] == h.table[b - 1] + 1;  // [v, v + 1)
      if (eqb) eq += h.hist[b];
      else if (h.hist[b] > mx) mx = h.hist[b];
    }
    std::printf("  buckets %u, fullest ordinary bucket %.2f x n/256, %.1f %% of the keys in equality buckets, histogram total %llu (%s), "
                "keys outside their bucket's range: %llu\n",
                h.nsp + 1, (double)mx / ((double)n / 256.0), 100.0 * (double)eq / (double)n, total, total == (unsigned long long)n ? "= n" : "!= n",
                h.violations);
    std::printf("  LUT: %s cells, fullest cell holds %u splitters (linear form %u, logarithmic form %u)\n", h.lut_log ? "logarithmic" : "linear",
                h.lut_worst[h.lut_log], h.lut_worst[0], h.lut_worst[1]);
    std::printf("  => tree search: %.2f TB/s (histogram), %.2f TB/s (level 0); the check above is of the LUT form's output\n", 8.0 * n / t_hist * 1e-9,
                16.0 * n / t_scat * 1e-9);
  }
  return 0;
}
