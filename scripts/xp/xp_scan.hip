// Micro-benchmark (not product code): variants of the single-pass look-back scan of 1e9 uint64 (inclusive sum), to
// decide what bounds it: the ds_bpermute wave scans, the ticket atomic, the tile size, the look-back window, or the
// workgroup churn.  Usage: xp_scan [rows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;

__global__ void __launch_bounds__(256) k_fill(u64* k, int64_t n)
{
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    u64 x = (u64)i + 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    k[i] = x ^ (x >> 31);
  }
}
__global__ void __launch_bounds__(256) k_copy(const u64* __restrict__ in, u64* __restrict__ out, int64_t n)
{
  const int64_t stride = (int64_t)gridDim.x * 256 * 2;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2; i < n; i += stride) {
    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(in + i);
    *reinterpret_cast<ulonglong2*>(out + i) = v;
  }
}
// order-sensitive fingerprint of the output
__global__ void __launch_bounds__(256) k_digest(const u64* __restrict__ v, int64_t n, u64* out)
{
  u64 acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += v[i] * (u64)(2 * i + 1);
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

__device__ __forceinline__ unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

template <int CTRL, int RM>
__device__ __forceinline__ u64 dpp_take(u64 v)
{
  const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, CTRL, RM, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(v >> 32), (int)(uint32_t)(v >> 32), CTRL, RM, 0xF, false);
  return ((u64)(uint32_t)hi << 32) | (uint32_t)lo;
}
template <bool DPP>
__device__ __forceinline__ u64 wave_scan(u64 v)
{
  const unsigned l = lane_id();
  if (DPP) {
    { const u64 o = dpp_take<0x111, 0xF>(v); if ((l & 15) >= 1) v += o; }
    { const u64 o = dpp_take<0x112, 0xF>(v); if ((l & 15) >= 2) v += o; }
    { const u64 o = dpp_take<0x114, 0xF>(v); if ((l & 15) >= 4) v += o; }
    { const u64 o = dpp_take<0x118, 0xF>(v); if ((l & 15) >= 8) v += o; }
    { const u64 o = dpp_take<0x142, 0xA>(v); if (l & 16) v += o; }
    { const u64 o = dpp_take<0x143, 0xC>(v); if (l >= 32) v += o; }
  } else {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const u64 o = __shfl_up(v, d, 64);
      if (l >= (unsigned)d) v += o;
    }
  }
  return v;
}
template <bool DPP>
__device__ __forceinline__ u64 last_lane(u64 v)
{
  if (DPP) {
    const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, 63), hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
    return ((u64)hi << 32) | lo;
  }
  return __shfl(v, 63, 64);
}
__device__ __forceinline__ u64 wave_sum(u64 v)
{
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

struct LbStatus {
  u64 lo, hi;
};
__device__ __forceinline__ void st_agent(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_agent(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lb_publish(LbStatus* st, unsigned flag, u64 bits)
{
  st_agent(&st->lo, ((u64)flag << 62) | (bits & 0xFFFFFFFFull));
  st_agent(&st->hi, ((u64)flag << 62) | (bits >> 32));
}
__device__ __forceinline__ unsigned lb_read(const LbStatus* st, u64& v)
{
  u64 l, h;
  unsigned spins = 0;
  for (;;) {
    l = ld_agent(&st->lo);
    h = ld_agent(&st->hi);
    if ((l >> 62) != 0 && (l >> 62) == (h >> 62)) break;
    if (++spins > (1u << 22)) break;
    __builtin_amdgcn_s_sleep(2);
  }
  v = (l & 0xFFFFFFFFull) | (h << 32);
  return (unsigned)(l >> 62);
}

// wave 0 of a tile: publish the aggregate, fold the predecessors, publish the inclusive prefix; returns the exclusive prefix
template <int WIN>
__device__ __forceinline__ u64 look_back(LbStatus* status, int64_t tile, u64 agg)
{
  const unsigned l = lane_id();
  u64 ex = 0;
  if (tile == 0) {
    if (l == 0) lb_publish(&status[0], 2u, agg);
    return 0;
  }
  if (l == 0) lb_publish(&status[tile], 1u, agg);
  int64_t pos = tile - 1;
  for (;;) {
    const int64_t idx = pos - (int64_t)l;
    u64 v             = 0;
    unsigned flag     = 2u;
    if (l >= WIN) flag = 1u;  // lanes outside the window: an empty aggregate
    else if (idx >= 0) flag = lb_read(&status[idx], v);
    const u64 m2 = __builtin_amdgcn_ballot_w64(flag == 2u);
    if (m2 != 0) {
      const unsigned first = (unsigned)__builtin_ctzll(m2);
      ex += wave_sum(l <= first ? v : 0);
      break;
    }
    ex += wave_sum(v);
    pos -= WIN;
  }
  if (l == 0) lb_publish(&status[tile], 2u, ex + agg);
  return ex;
}

// MODE bits: 1 = blockIdx order instead of a ticket, 2 = DPP wave scans, 4 = 16-byte loads (two elements per lane)
template <int BT, int IPT, int MODE, int WIN>
__global__ void __launch_bounds__(BT) k_lb_scan(const u64* __restrict__ in, int64_t n, LbStatus* status, unsigned* ticket, u64* __restrict__ out)
{
  constexpr bool DPP = (MODE & 2) != 0;
  constexpr bool VEC = (MODE & 4) != 0;
  constexpr int NWV  = BT / 64;
  __shared__ u64 s_w[NWV];
  __shared__ u64 s_ex;
  __shared__ unsigned s_tile;
  const unsigned l = lane_id(), w = threadIdx.x / 64;
  int64_t tile;
  if (MODE & 1) tile = blockIdx.x;
  else {
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    tile = s_tile;
  }
  const int64_t wbase = tile * (int64_t)(BT * IPT) + (int64_t)w * (64 * IPT);
  u64 inc[IPT];
  u64 carry = 0;
  if (!VEC) {
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      const int64_t i = wbase + k * 64 + l;
      inc[k]          = i < n ? in[i] : 0;
    }
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      inc[k] = wave_scan<DPP>(inc[k]);
      carry += last_lane<DPP>(inc[k]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < IPT / 2; ++k) {
      const int64_t i = wbase + k * 128 + 2 * l;
      ulonglong2 v    = make_ulonglong2(0, 0);
      if (i + 1 < n) v = *reinterpret_cast<const ulonglong2*>(in + i);
      else if (i < n) v.x = in[i];
      inc[2 * k]     = v.x;
      inc[2 * k + 1] = v.y;
    }
#pragma unroll
    for (int k = 0; k < IPT / 2; ++k) {
      const u64 a = inc[2 * k], b = inc[2 * k + 1];
      const u64 s = wave_scan<DPP>(a + b);
      inc[2 * k + 1] = s;
      inc[2 * k]     = s - b;
      carry += last_lane<DPP>(s);
    }
  }
  if (l == 0) s_w[w] = carry;
  __syncthreads();
  if (w == 0) {
    u64 agg = 0;
#pragma unroll
    for (int k = 0; k < NWV; ++k) agg += s_w[k];
    const u64 ex = look_back<WIN>(status, tile, agg);
    if (l == 0) s_ex = ex;
  }
  __syncthreads();
  u64 run = s_ex;
  for (unsigned k = 0; k < w; ++k) run += s_w[k];
  if (!VEC) {
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      const int64_t i = wbase + k * 64 + l;
      if (i < n) out[i] = run + inc[k];
      run += last_lane<DPP>(inc[k]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < IPT / 2; ++k) {
      const int64_t i = wbase + k * 128 + 2 * l;
      if (i + 1 < n) *reinterpret_cast<ulonglong2*>(out + i) = make_ulonglong2(run + inc[2 * k], run + inc[2 * k + 1]);
      else if (i < n) out[i] = run + inc[2 * k];
      run += last_lane<DPP>(inc[2 * k + 1]);
    }
  }
}

// ---- two launches over a persistent grid: every workgroup owns one contiguous range; (1) its sum, (2) its scan with the
// next tile's loads in flight while the current one is scanned and stored
constexpr int PBT = 256, PIPT = 16, PTILE = PBT * PIPT;
__global__ void __launch_bounds__(PBT) k_range_sum(const u64* __restrict__ in, int64_t n, int64_t range, u64* partial)
{
  __shared__ u64 s_w[PBT / 64];
  const int64_t b = (int64_t)blockIdx.x * range, e = b + range < n ? b + range : n;
  u64 acc = 0;
  for (int64_t i0 = b + threadIdx.x * 2; i0 < e; i0 += PBT * 2 * 8) {
    ulonglong2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t i = i0 + (int64_t)u * PBT * 2;
      v[u]            = make_ulonglong2(0, 0);
      if (i + 1 < e) v[u] = *reinterpret_cast<const ulonglong2*>(in + i);
      else if (i < e) v[u].x = in[i];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y;
  }
  acc = wave_sum(acc);
  if (lane_id() == 0) s_w[threadIdx.x / 64] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 t = 0;
    for (int k = 0; k < PBT / 64; ++k) t += s_w[k];
    partial[blockIdx.x] = t;
  }
}
__device__ __forceinline__ void load_tile(const u64* __restrict__ in, int64_t wbase, int64_t e, unsigned l, ulonglong2 (&v)[PIPT / 2])
{
#pragma unroll
  for (int k = 0; k < PIPT / 2; ++k) {
    const int64_t i = wbase + k * 128 + 2 * l;
    v[k]            = make_ulonglong2(0, 0);
    if (i + 1 < e) v[k] = *reinterpret_cast<const ulonglong2*>(in + i);
    else if (i < e) v[k].x = in[i];
  }
}
__global__ void __launch_bounds__(PBT) k_range_scan(const u64* __restrict__ in, int64_t n, int64_t range, const u64* __restrict__ partial,
                                                    u64* __restrict__ out)
{
  constexpr int NWV = PBT / 64;
  __shared__ u64 s_w[2][NWV];
  __shared__ u64 s_pre[NWV];
  const unsigned l = lane_id(), w = threadIdx.x / 64;
  const int64_t b = (int64_t)blockIdx.x * range, e = b + range < n ? b + range : n;
  if (b >= e) return;
  // everything before this range
  u64 pre = 0;
  for (unsigned i = threadIdx.x; i < blockIdx.x; i += PBT) pre += partial[i];
  pre = wave_sum(pre);
  if (l == 0) s_pre[w] = pre;
  __syncthreads();
  u64 carry = 0;
  for (int k = 0; k < NWV; ++k) carry += s_pre[k];
  ulonglong2 cur[PIPT / 2], nxt[PIPT / 2];
  load_tile(in, b + (int64_t)w * (64 * PIPT), e, l, cur);
  int buf = 0;
  for (int64_t t = b; t < e; t += PTILE, buf ^= 1) {
    const int64_t wbase = t + (int64_t)w * (64 * PIPT);
    if (t + PTILE < e) load_tile(in, wbase + PTILE, e, l, nxt);
    u64 inc[PIPT];
    u64 wc = 0;
#pragma unroll
    for (int k = 0; k < PIPT / 2; ++k) {
      const u64 s    = wave_scan<true>(cur[k].x + cur[k].y);
      inc[2 * k + 1] = s;
      inc[2 * k]     = s - cur[k].y;
      wc += last_lane<true>(s);
    }
    if (l == 0) s_w[buf][w] = wc;
    __syncthreads();
    u64 run = carry, total = 0;
#pragma unroll
    for (int k = 0; k < NWV; ++k) {
      const u64 x = s_w[buf][k];
      if ((unsigned)k < w) run += x;
      total += x;
    }
    carry += total;
#pragma unroll
    for (int k = 0; k < PIPT / 2; ++k) {
      const int64_t i = wbase + k * 128 + 2 * l;
      if (i + 1 < e) *reinterpret_cast<ulonglong2*>(out + i) = make_ulonglong2(run + inc[2 * k], run + inc[2 * k + 1]);
      else if (i < e) out[i] = run + inc[2 * k];
      run += last_lane<true>(inc[2 * k + 1]);
    }
#pragma unroll
    for (int k = 0; k < PIPT / 2; ++k) cur[k] = nxt[k];
  }
}

struct Bench {
  u64 *in, *out, *dig, *partial;
  char* scratch;
  int64_t n;
  hipEvent_t e0, e1;
  u64 want = 0;
  template <typename F>
  void run(const char* name, F launch)
  {
    float best = 1e9f, sum = 0;
    const int reps = 6;
    for (int r = 0; r < reps + 1; ++r) {
      CK(hipMemsetAsync(out, 0xFF, 64, 0));
      CK(hipEventRecord(e0, 0));
      launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r == 0) continue;
      best = ms < best ? ms : best;
      sum += ms;
    }
    CK(hipMemset(dig, 0, 8));
    hipLaunchKernelGGL(k_digest, dim3(4096), dim3(256), 0, 0, out, n, dig);
    u64 d;
    CK(hipMemcpy(&d, dig, 8, hipMemcpyDeviceToHost));
    if (want == 0) want = d;
    printf("%-58s best %7.3f ms  mean %7.3f ms  %6.2f TB/s (16 B/row)  %s\n", name, best, sum / reps, 16.0 * n / best / 1e9,
           d == want ? "ok" : "DIGEST DIFFERS");
    fflush(stdout);
  }
  template <int BT, int IPT, int MODE, int WIN>
  void lb(const char* name)
  {
    const int64_t tiles = (n + BT * IPT - 1) / (BT * IPT);
    run(name, [&] {
      CK(hipMemsetAsync(scratch, 0, 256 + (size_t)(tiles + 1) * sizeof(LbStatus), 0));
      hipLaunchKernelGGL((k_lb_scan<BT, IPT, MODE, WIN>), dim3((unsigned)tiles), dim3(BT), 0, 0, in, n,
                         reinterpret_cast<LbStatus*>(scratch + 256), reinterpret_cast<unsigned*>(scratch), out);
    });
  }
};

int main(int argc, char** argv)
{
  Bench b;
  b.n = argc > 1 ? (int64_t)atof(argv[1]) : 1000000000ll;
  CK(hipMalloc(&b.in, b.n * 8 + 64));
  CK(hipMalloc(&b.out, b.n * 8 + 64));
  CK(hipMalloc(&b.dig, 8));
  CK(hipMalloc(&b.partial, 8 * 65536));
  CK(hipMalloc(&b.scratch, 256 + (size_t)(b.n / 2048 + 2) * sizeof(LbStatus)));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, b.in, b.n);
  CK(hipDeviceSynchronize());
  CK(hipEventCreate(&b.e0));
  CK(hipEventCreate(&b.e1));
  b.lb<256, 16, 0, 64>("look-back 256x16 ticket bpermute win64 (product today)");
  b.lb<256, 16, 2, 64>("look-back 256x16 ticket DPP win64");
  b.lb<256, 16, 6, 64>("look-back 256x16 ticket DPP vec2 win64");
  b.lb<256, 16, 6, 16>("look-back 256x16 ticket DPP vec2 win16");
  b.lb<512, 16, 6, 64>("look-back 512x16 ticket DPP vec2 win64");
  b.lb<1024, 16, 6, 64>("look-back 1024x16 ticket DPP vec2 win64");
  b.lb<256, 32, 6, 64>("look-back 256x32 ticket DPP vec2 win64");
  b.lb<512, 8, 6, 64>("look-back 512x8 ticket DPP vec2 win64");
  b.lb<256, 16, 7, 64>("look-back 256x16 blockIdx DPP vec2 win64");
  for (int g : {2048, 1536, 1024, 4096}) {
    int64_t range = (b.n + g - 1) / g;
    range         = (range + PTILE - 1) / PTILE * PTILE;
    char name[96];
    snprintf(name, sizeof name, "two-launch persistent ranges, %d workgroups", g);
    b.run(name, [&] {
      hipLaunchKernelGGL(k_range_sum, dim3(g), dim3(PBT), 0, 0, b.in, b.n, range, b.partial);
      hipLaunchKernelGGL(k_range_scan, dim3(g), dim3(PBT), 0, 0, b.in, b.n, range, b.partial, b.out);
    });
  }
  {
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(b.e0, 0));
      hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, b.in, b.out, b.n);
      CK(hipEventRecord(b.e1, 0));
      CK(hipEventSynchronize(b.e1));
      float ms;
      CK(hipEventElapsedTime(&ms, b.e0, b.e1));
      best = ms < best ? ms : best;
    }
    printf("%-58s best %7.3f ms  %6.2f TB/s\n", "copy (grid-stride, 16-byte loads)", best, 16.0 * b.n / best / 1e9);
  }
  return 0;
}
