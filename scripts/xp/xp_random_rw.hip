// xp_random_rw.hip -- micro-benchmark (NOT product code): what do 2^30 random 8-byte READS cost against 2^30 random 8-byte WRITES on this part?
// gx_gather of 8-byte rows through a random map runs at one 128-byte line per row (FETCH 140 B/row, 26-28 ms per 1e9 rows).  If random WRITES
// are cheaper (a partial line leaves the L2 as masked sectors instead of arriving as a whole line), a gather of a large column can be turned
// into: partition the (destination, source) pairs by SOURCE block (blocks the L2 holds) -> read the values from the L2-resident block ->
// write them to their random destinations.  Variants over n = 2^30 rows, index = a mixing bijection of [0, 2^30):
//   gather      out[i]      = in[idx(i)]
//   scatter     out[idx(i)] = in[i]
//   scatter_nt  the same with nontemporal stores
//   gather_blk  out[i] = in[base(b) + (idx(i) & (BLK - 1))], block b = 2 MiB of `in` walked by the workgroups of ONE XCD at a time: the
//               reads an L2-resident source block gives (upper bound of the proposed pass 2's read side)
//   scatter_blk out[idx(i)] = in[i] where the rows of a workgroup come from one 2-MiB source block (the proposed pass 2, without its records)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/xp/xp_random_rw.hip -o /tmp/xp_rw && /tmp/xp_rw
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      std::fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(1);                                                                            \
    }                                                                                          \
  } while (0)

constexpr int LOGN = 30;
constexpr uint32_t MASK = (1u << LOGN) - 1;
__device__ __forceinline__ uint32_t idx_of(uint32_t i)
{  // a bijection of [0, 2^30): xor-shifts and odd multipliers modulo 2^30 (not a lattice: a plain multiplicative map gives every wave a fixed stride)
  uint32_t x = i & MASK;
  x ^= x >> 15;
  x = (x * 0x2C1B3C6Du) & MASK;
  x ^= x >> 12;
  x = (x * 0x297A2D39u) & MASK;
  x ^= x >> 15;
  return x;
}

__global__ void k_fill(uint64_t* p, int64_t n)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (uint64_t)i * 0x9E3779B97F4A7C15ull;
}
template <int MODE>
__global__ void __launch_bounds__(256) k_rw(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int64_t n)
{
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += stride) {
    uint64_t v[4];
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = in[idx_of((uint32_t)(i0 + k))];
      *reinterpret_cast<ulonglong2*>(out + i0)     = make_ulonglong2(v[0], v[1]);
      *reinterpret_cast<ulonglong2*>(out + i0 + 2) = make_ulonglong2(v[2], v[3]);
    } else {
      const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(in + i0), b = *reinterpret_cast<const ulonglong2*>(in + i0 + 2);
      v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (MODE == 1) out[idx_of((uint32_t)(i0 + k))] = v[k];
        else __builtin_nontemporal_store(v[k], &out[idx_of((uint32_t)(i0 + k))]);
      }
    }
  }
}
// block forms: workgroup w of XCD x (= blockIdx % 8) walks the source blocks x, x + 8, ... ; BLKROWS rows of 8 bytes = 2 MiB
constexpr int BLKROWS = 1 << 18;
template <int MODE>
__global__ void __launch_bounds__(256) k_blk(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int64_t n)
{
  const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int64_t nblk = n / BLKROWS;
  for (int64_t b = xcd; b < nblk; b += 8) {
    const int64_t base = b * BLKROWS;
    for (int r0 = (w * 256 + (int)threadIdx.x) * 4; r0 < BLKROWS; r0 += per * 256 * 4) {
      uint64_t v[4];
      if (MODE == 0) {  // random reads INSIDE the block, streaming writes
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = in[base + (idx_of((uint32_t)(base + r0 + k)) & (BLKROWS - 1))];
        *reinterpret_cast<ulonglong2*>(out + base + r0)     = make_ulonglong2(v[0], v[1]);
        *reinterpret_cast<ulonglong2*>(out + base + r0 + 2) = make_ulonglong2(v[2], v[3]);
      } else {  // random reads inside the block, random writes over the whole output
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = in[base + (idx_of((uint32_t)(base + r0 + k)) & (BLKROWS - 1))];
#pragma unroll
        for (int k = 0; k < 4; ++k) out[idx_of((uint32_t)(base + r0 + k))] = v[k];
      }
    }
  }
}

int main()
{
  const int64_t n = 1ll << LOGN;
  uint64_t *in, *out;
  CK(hipMalloc(&in, n * 8));
  CK(hipMalloc(&out, n * 8));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, in, n);
  CK(hipMemset(out, 0, n * 8));
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto launch) {
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
      CK(hipEventRecord(e0, 0));
      launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r && ms < best) best = ms;
    }
    std::printf("%-12s best %.3f ms per 2^30 rows  (%.1f G rows/s)\n", name, best, n / (best * 1e-3) / 1e9);
  };
  std::printf("# xp_random_rw: 2^30 rows of 8 bytes, index = mixing bijection\n");
  time("gather", [&] { hipLaunchKernelGGL((k_rw<0>), dim3(8192), dim3(256), 0, 0, in, out, n); });
  time("scatter", [&] { hipLaunchKernelGGL((k_rw<1>), dim3(8192), dim3(256), 0, 0, in, out, n); });
  time("scatter_nt", [&] { hipLaunchKernelGGL((k_rw<2>), dim3(8192), dim3(256), 0, 0, in, out, n); });
  time("gather_blk", [&] { hipLaunchKernelGGL((k_blk<0>), dim3(2048), dim3(256), 0, 0, in, out, n); });
  time("scatter_blk", [&] { hipLaunchKernelGGL((k_blk<1>), dim3(2048), dim3(256), 0, 0, in, out, n); });
  return 0;
}
