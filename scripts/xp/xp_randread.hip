// Micro-benchmark (not product code): how fast can 1e9 coalesced 8-byte keys be turned into random 4-byte
// reads of a tag array of a given size?  Decides whether a pre-partition miss filter on the join's 4-bit tags
// (2^28 slots -> 128 MiB, Infinity-Cache sized) can pay for itself.  Usage: xp_randread <rows> <log2 slots>...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_fill(uint64_t* k, int64_t n)
{
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    uint64_t x = (uint64_t)i + 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    k[i] = x ^ (x >> 31);
  }
}

template <int U>
__global__ void __launch_bounds__(256) k_randread(const uint64_t* __restrict__ keys, int64_t n, const uint32_t* __restrict__ tags,
                                                  int lg, unsigned long long* __restrict__ bits)
{
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 * U + threadIdx.x; i0 < n; i0 += stride) {
    uint64_t k[U];
    uint32_t t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) k[u] = (i0 + u * 256 < n) ? __builtin_nontemporal_load(&keys[i0 + u * 256]) : 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t h = (k[u] * 0x9E3779B97F4A7C15ull) >> (64 - lg);
      t[u]             = tags[h >> 3];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t h   = (k[u] * 0x9E3779B97F4A7C15ull) >> (64 - lg);
      const bool hit     = ((t[u] >> ((h & 7) * 4)) & 15u) != 0;
      const uint64_t b   = __builtin_amdgcn_ballot_w64(hit);
      if ((threadIdx.x & 63) == 0 && i0 + u * 256 < n) bits[(i0 + u * 256) >> 6] = b;
    }
  }
}

int main(int argc, char** argv)
{
  const int64_t n = argc > 1 ? (int64_t)atof(argv[1]) : 1000000000ll;
  uint64_t* keys;
  unsigned long long* bits;
  CK(hipMalloc(&keys, n * 8));
  CK(hipMalloc(&bits, (n / 64 + 1) * 8));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, keys, n);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int a = 2; a < argc; ++a) {
    const int lg = atoi(argv[a]);
    const size_t tbytes = ((size_t)1 << lg) / 2;
    uint32_t* tags;
    CK(hipMalloc(&tags, tbytes));
    CK(hipMemset(tags, 0x5A, tbytes));
    for (int U : {4, 8}) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        if (U == 4) hipLaunchKernelGGL((k_randread<4>), dim3(256 * 16), dim3(256), 0, 0, keys, n, tags, lg, bits);
        else hipLaunchKernelGGL((k_randread<8>), dim3(256 * 8), dim3(256), 0, 0, keys, n, tags, lg, bits);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
      }
      printf("{\"xp\": \"randread\", \"rows\": %lld, \"log2_slots\": %d, \"tag_MiB\": %.1f, \"unroll\": %d, \"ms\": %.3f, \"Greads_per_s\": %.1f}\n",
             (long long)n, lg, tbytes / 1048576.0, U, best, n / best / 1e6);
    }
    CK(hipFree(tags));
  }
  return 0;
}
