// xp_scatter_ceiling.hip -- micro-benchmark (NOT product code): what can a BARE 256-way partition of 8-byte keys reach on this part?
//
// the three passes of the 1e9-row cudf::sort each move 16 GB in ~3.65 ms = 4.4 TB/s = 0.55 of the 8 TB/s spec, flat
// since round 3, and "at the ceiling" was asserted, not shown.  This file is the smallest kernel that does what level 0 must do -- read
// every key once, write every key once into one of 256 bins -- with nothing else in it: no sample plan, no exact-mask reduction, no slot
// verdict, no splitter forms, no key transform.  Variants, 16 B/row each, n keys (default 1e9, uniform 64-bit):
//   copy      : out[i] = in[i], 16-byte accesses, persistent grid                              -- the box's streaming rate
//   copy8     : the same with 8-byte accesses (what a key-granular kernel can issue)
//   scatter   : the product's structure, bare: 3584-key tiles (256 threads x 14), digit = top byte, rank through LDS atomics, one
//               returning atomic per (tile, non-empty bin) on the cursor of (XCD range, bin) -- padded slots of mean + 8 sigma --, keys
//               reordered through LDS so that a bin's keys leave as ONE run, XCD-contiguous tile assignment
//   scatter_nolds: the same without the LDS reorder: every key written straight to cursor + rank (8-byte scattered stores)
//   scatter_1cursor: one cursor per bin for the whole chip instead of per (XCD range, bin): lines shared by all eight L2s
// Output: one line per variant: ms, TB/s on 16 B/row, and (scatter*) a check that every key sits in its bin's slot range.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/xp/xp_scatter_ceiling.hip -o /tmp/xp_sc && /tmp/xp_sc [n]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) {                                                                     \
      std::fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));  \
      std::exit(1);                                                                             \
    }                                                                                           \
  } while (0)

constexpr int BT    = 256;
constexpr int KPT   = 14;
constexpr int TILE  = BT * KPT;  // 3584 keys = 28 KiB: three workgroups per CU like the product's level 0
constexpr int BINS  = 256;
constexpr int NR    = 8;         // XCD ranges

__global__ void k_fill(uint64_t* k, int64_t n)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = (uint64_t)i + 0x9E3779B97F4A7C15ull;
    x          = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x          = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    k[i]       = x ^ (x >> 31);
  }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_copy16(const u32x4* __restrict__ in, u32x4* __restrict__ out, int64_t n16)
{
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load(&in[i]), &out[i]);
}
__global__ void __launch_bounds__(256) k_copy8(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int64_t n)
{
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load(&in[i]), &out[i]);
}

// XCD x works on a contiguous eighth of the tiles (block b runs on XCD b % 8: observed, speed only)
__device__ __forceinline__ int64_t xcd_swizzle(int64_t bid, int64_t nblocks)
{
  const int64_t full = nblocks / NR * NR;
  if (bid >= full) return bid;
  const int64_t per = full / NR;
  return (bid % NR) * per + bid / NR;
}

struct alignas(128) Cursor {
  unsigned int v;
  unsigned int pad[31];
};

// MODE 0: LDS reorder + per-(range, bin) cursors; 1: no LDS reorder; 2: one cursor per bin
template <int MODE>
__global__ void __launch_bounds__(BT) k_scatter(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int64_t n, int64_t ntiles, Cursor* cur,
                                                unsigned int cap)
{
  __shared__ uint64_t s_k[MODE == 1 ? 1 : TILE];
  __shared__ unsigned int s_cnt[BINS], s_start[BINS], s_delta[BINS];
  __shared__ unsigned int s_w[BT / 64 + 1];
  const unsigned tid  = threadIdx.x;
  const int64_t tile  = xcd_swizzle(blockIdx.x, ntiles);
  const int64_t base  = tile * TILE;
  const int range     = MODE == 2 ? 0 : (int)((tile * NR) / ntiles);
  s_cnt[tid]          = 0;
  __syncthreads();
  uint64_t k[KPT];
  unsigned int rk[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int64_t i = base + j * BT + tid;
    k[j]            = i < n ? __builtin_nontemporal_load(&in[i]) : ~0ull;
  }
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int64_t i = base + j * BT + tid;
    rk[j]           = i < n ? atomicAdd(&s_cnt[k[j] >> 56], 1u) : 0u;
  }
  __syncthreads();
  // thread t owns bin t: reserve its run (one returning atomic per non-empty bin), exclusive scan of the counts for the LDS positions
  const unsigned int c = s_cnt[tid];
  unsigned int g       = 0;
  if (c) g = atomicAdd(&cur[range * BINS + tid].v, c);
  unsigned int inc = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned int u = __shfl_up(inc, o);
    if ((tid & 63) >= (unsigned)o) inc += u;
  }
  if ((tid & 63) == 63) s_w[tid >> 6] = inc;
  __syncthreads();
  unsigned int wb = 0;
  for (unsigned w = 0; w < (tid >> 6); ++w) wb += s_w[w];
  const unsigned int st = wb + inc - c;
  s_start[tid]          = st;
  s_delta[tid]          = (unsigned int)(range * BINS + tid) * cap + g - st;  // global position = delta + tile position
  __syncthreads();
  if (MODE == 1) {
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int64_t i = base + j * BT + tid;
      if (i < n) {
        const unsigned int b = (unsigned int)(k[j] >> 56);
        out[(size_t)s_delta[b] + s_start[b] + rk[j]] = k[j];
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int64_t i = base + j * BT + tid;
    if (i < n) s_k[s_start[k[j] >> 56] + rk[j]] = k[j];
  }
  __syncthreads();
  const int nv = (int)((n - base) < TILE ? (n - base) : TILE);
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int p = j * BT + tid;
    if (p < nv) {
      const uint64_t key = s_k[p];
      out[(size_t)s_delta[key >> 56] + p] = key;
    }
  }
}

__global__ void k_check(const uint64_t* out, const Cursor* cur, unsigned int cap, int nreg, unsigned long long* bad, unsigned long long* total)
{
  for (int r = blockIdx.x; r < nreg; r += gridDim.x) {
    const unsigned int c = cur[r].v;
    unsigned long long b = 0;
    for (unsigned int i = threadIdx.x; i < c && i < cap; i += blockDim.x)
      if ((out[(size_t)r * cap + i] >> 56) != (unsigned)(r % BINS)) ++b;
    if (c > cap) b += c - cap;
    if (b) atomicAdd(bad, b);
    if (threadIdx.x == 0) atomicAdd(total, (unsigned long long)c);
  }
}

int main(int argc, char** argv)
{
  const int64_t n = argc > 1 ? (int64_t)atof(argv[1]) : 1000000000ll;
  const int64_t ntiles = (n + TILE - 1) / TILE;
  const double mean = (double)n / NR / BINS;
  const unsigned int cap = (unsigned int)(mean + 8.0 * std::sqrt(mean) + 64.0 + TILE) / 32 * 32 + 32;  // (+ a tile: the range of a tile is approximate)
  const unsigned int cap1 = (unsigned int)((double)n / BINS + 8.0 * std::sqrt((double)n / BINS) + 64.0) / 32 * 32 + 32;
  uint64_t *in, *out;
  Cursor* cur;
  unsigned long long* res;
  const size_t out_rows = (size_t)BINS * cap1 > (size_t)NR * BINS * cap ? (size_t)BINS * cap1 : (size_t)NR * BINS * cap;
  CK(hipMalloc(&in, n * 8));
  CK(hipMalloc(&out, (out_rows > (size_t)n ? out_rows : (size_t)n) * 8));
  CK(hipMalloc(&cur, sizeof(Cursor) * NR * BINS));
  CK(hipMalloc(&res, 16));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, in, n);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int cus = 256;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  std::printf("# xp_scatter_ceiling: n = %lld uint64 keys, %d CUs, tile %d keys, slot cap %u rows per (range, bin)\n", (long long)n, cus, TILE, cap);
  auto time = [&](const char* name, auto launch, bool check, unsigned int ccap, int nreg) {
    float best = 1e9f, sum = 0;
    const int reps = 5;
    for (int r = 0; r < reps + 1; ++r) {
      CK(hipMemsetAsync(cur, 0, sizeof(Cursor) * NR * BINS, 0));
      CK(hipEventRecord(e0, 0));
      launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r == 0) continue;  // warm-up
      sum += ms;
      if (ms < best) best = ms;
    }
    unsigned long long h[2] = {0, 0};
    if (check) {
      CK(hipMemset(res, 0, 16));
      hipLaunchKernelGGL(k_check, dim3(2048), dim3(256), 0, 0, out, cur, ccap, nreg, res, res + 1);
      CK(hipMemcpy(h, res, 16, hipMemcpyDeviceToHost));
    }
    std::printf("%-16s avg %.3f ms  best %.3f ms  %.2f TB/s on 16 B/row (best)%s", name, sum / reps, best, 16.0 * n / (best * 1e-3) / 1e12, check ? "" : "\n");
    if (check) std::printf("  | keys outside their slot or dropped: %llu, keys placed: %llu of %lld\n", h[0], h[1], (long long)n);
  };
  time("copy (16 B)", [&] { hipLaunchKernelGGL(k_copy16, dim3(cus * 8), dim3(256), 0, 0, (const u32x4*)in, (u32x4*)out, n / 2); }, false, 0, 0);
  time("copy8 (8 B)", [&] { hipLaunchKernelGGL(k_copy8, dim3(cus * 8), dim3(256), 0, 0, in, out, n); }, false, 0, 0);
  time("scatter", [&] { hipLaunchKernelGGL((k_scatter<0>), dim3((unsigned)ntiles), dim3(BT), 0, 0, in, out, n, ntiles, cur, cap); }, true, cap, NR * BINS);
  time("scatter_nolds", [&] { hipLaunchKernelGGL((k_scatter<1>), dim3((unsigned)ntiles), dim3(BT), 0, 0, in, out, n, ntiles, cur, cap); }, true, cap, NR * BINS);
  time("scatter_1cursor", [&] { hipLaunchKernelGGL((k_scatter<2>), dim3((unsigned)ntiles), dim3(BT), 0, 0, in, out, n, ntiles, cur, cap1); }, true, cap1, BINS);
  return 0;
}
