// distributed.cpp -- the sharded operators of include/cudf_amd/gxd.h: host-side C++ over the HIP kernels (gx.h) and RCCL.
// One process per GPU.  Replaces, for this path, the shuffle of libcudf_streaming (cpp/libcudf_streaming/src/
// partition_utils.cpp:72-117, partition.cpp:56-80) + rapidsmpf and the collectives of cudf_polars' streaming executor
// (python/cudf_polars/cudf_polars/streaming/actor_graph/collectives/sort.py, streaming/join.py:58-135,
// streaming/groupby.py:411-437).
//
// Shape of every operator (counts all-gather, then an all-to-all as grouped ncclSend / ncclRecv pairs):
//   caller's stream : partition chunk 0 | partition chunk 1 | ... | partition chunk C-1 | local operator (chunks)
//   exchange stream :   wait P0, counts(0), send/recv(0) | wait P1, counts(1), send/recv(1) | ...
//   host            : enqueues ALL partition passes first, then per chunk waits only for that chunk's count matrix
//                     (a 64-entry D2H copy) -- the GPU is busy with the later chunks meanwhile.
// Receive buffers, partition buffers and scratch are slots of a grow-only arena kept by the communicator object.
#include <cstdio>
#include <cstdlib>
#include <cudf_amd/gxd.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int rc, std::string what)
{
  g_err = std::move(what);
  return rc;
}
#define GXD_HIP(expr)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return fail((int)e_, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define GXD_NCCL(expr)                                                                                 \
  do {                                                                                                 \
    ncclResult_t r_ = (expr);                                                                          \
    if (r_ != ncclSuccess) return fail(GX_EINTERNAL, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
  } while (0)
#define GXD_GX(expr)                                                                 \
  do {                                                                               \
    int g_ = (expr);                                                                 \
    if (g_ != 0) return fail(g_, std::string(#expr) + " returned " + std::to_string(g_)); \
  } while (0)

double now_ms()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// GXD_TRACE=1: print where an operator call spends its host time (each mark synchronises the device first)
struct Trace {
  bool on;
  double last;
  const char* op;
  explicit Trace(const char* o) : on(std::getenv("GXD_TRACE") != nullptr), last(0), op(o)
  {
    if (on) last = now_ms();
  }
  void mark(const char* what)
  {
    if (!on) return;
    (void)hipDeviceSynchronize();
    const double t = now_ms();
    std::fprintf(stderr, "[gxd trace] %-16s %-28s %9.3f ms\n", op, what, t - last);
    last = t;
  }
};

// grow-only device buffers, one per purpose, reused from call to call
struct Arena {
  enum Slot { PART_KEYS, PART_ROWS, RECV_KEYS, RECV_ROWS, TMP, TMP2, OFFS, ALLOFFS, SAMPLE, ALLSAMPLE, PAIR_L, PAIR_R, CURSOR, MISC_A,
              MISC_B, MISC_C, MISC_D, MISC_E, MISC_F, SEGTAB, EXACT_KEYS, EXACT_ROWS, EXACT_OFFS, TMP3, CNT, NSLOTS };
  void* p[NSLOTS]      = {};
  size_t cap[NSLOTS]   = {};
  // at least `bytes`; the contents are NOT kept
  int get(int s, size_t bytes, void** out)
  {
    if (bytes > cap[s]) {
      GXD_HIP(hipDeviceSynchronize());  // nothing may still be using the old block
      if (p[s]) GXD_HIP(hipFree(p[s]));
      p[s]   = nullptr;
      cap[s] = 0;
      size_t want = bytes + bytes / 8 + 4096;
      GXD_HIP(hipMalloc(&p[s], want));
      cap[s] = want;
    }
    *out = p[s];
    return 0;
  }
  // at least `bytes`, the first `keep` bytes preserved
  int grow(int s, size_t bytes, size_t keep, void** out)
  {
    if (bytes > cap[s]) {
      GXD_HIP(hipDeviceSynchronize());
      void* q     = nullptr;
      size_t want = bytes + bytes / 4 + 4096;
      GXD_HIP(hipMalloc(&q, want));
      if (p[s] && keep) GXD_HIP(hipMemcpy(q, p[s], keep, hipMemcpyDeviceToDevice));
      if (p[s]) GXD_HIP(hipFree(p[s]));
      p[s]   = q;
      cap[s] = want;
    }
    *out = p[s];
    return 0;
  }
  void release()
  {
    for (int s = 0; s < NSLOTS; ++s)
      if (p[s]) (void)hipFree(p[s]);
  }
};


// ---------------------------------------------------------------------------------------------------------------- transport
// The two collectives the sharded operators need ("ncclAllGather of the count matrix, then all-to-all as grouped
// ncclSend / ncclRecv pairs"), behind a seam so that the SAME operator code runs over
//   * RCCL (one process per GPU, xGMI) -- the product transport, and
//   * an in-process loopback fabric: W logical ranks = W communicators on ONE device, each driven by its own host thread,
//     device-to-device copies standing in for the links (SURVEY.md 8e's prescription when only one GPU can be reached;
//     tests/test_gpu_distributed_loopback.py).  Every r != rank branch, send offset and receive position of the operators
//     executes under it exactly as it would over RCCL.
// Semantics both keep: calls are stream-ordered on the stream passed in; a send buffer may be reused once the work behind the
// call has completed on that stream; sends and receives between one pair of ranks match in posting order and must agree in size.
struct Transport {
  virtual ~Transport() = default;
  virtual int allgather(const void* mine_dev, void* all_dev, size_t bytes_per_rank, hipStream_t s) = 0;
  virtual int group_start()                                                                      = 0;
  virtual int send(const void* p, size_t bytes, int peer, hipStream_t s)                          = 0;
  virtual int recv(void* p, size_t bytes, int peer, hipStream_t s)                                = 0;
  virtual int group_end(hipStream_t s)                                                            = 0;
  // called from ANOTHER host thread while an operator of this communicator may be blocked inside a collective: make that
  // call (and every later one) fail instead of waiting for a peer that will never come (gxd_comm_abort)
  virtual void abort() = 0;
};

struct RcclTransport final : Transport {
  ncclComm_t comm = nullptr;
  std::atomic<bool> aborted{false};
  ~RcclTransport() override
  {
    if (comm && !aborted.load()) (void)ncclCommDestroy(comm);
  }
  int dead() const { return aborted.load() ? fail(GX_EINTERNAL, "gxd: the communicator was aborted (gxd_comm_abort)") : 0; }
  int allgather(const void* mine, void* all, size_t bytes, hipStream_t s) override
  {
    GXD_GX(dead());
    GXD_NCCL(ncclAllGather(mine, all, bytes, ncclInt8, comm, s));
    return 0;
  }
  int group_start() override
  {
    GXD_GX(dead());
    GXD_NCCL(ncclGroupStart());
    return 0;
  }
  // (every entry checks the abort flag: a call racing gxd_comm_abort must not touch the aborted ncclComm -- ADVICE r5)
  int send(const void* p, size_t bytes, int peer, hipStream_t s) override
  {
    GXD_GX(dead());
    GXD_NCCL(ncclSend(p, bytes, ncclInt8, peer, comm, s));
    return 0;
  }
  int recv(void* p, size_t bytes, int peer, hipStream_t s) override
  {
    GXD_GX(dead());
    GXD_NCCL(ncclRecv(p, bytes, ncclInt8, peer, comm, s));
    return 0;
  }
  int group_end(hipStream_t) override
  {
    GXD_GX(dead());
    GXD_NCCL(ncclGroupEnd());
    return 0;
  }
  void abort() override
  {
    // ncclCommAbort: the kernels of in-flight collectives leave at their next poll of the abort flag, the stream drains, the
    // thread blocked in hipStreamSynchronize behind them returns; the communicator is gone afterwards (no ncclCommDestroy)
    if (!aborted.exchange(true) && comm) (void)ncclCommAbort(comm);
  }
};

// What the W loopback ranks share.  A collective is three host barriers: (1) every rank has published its source pointers and
// recorded `ready` on its stream -> peers make their streams wait for it and enqueue their copies, then record `done`;
// (2) every `done` is recorded -> a rank's stream waits for all of them (its send buffers are free again, as after an RCCL
// call); (3) every rank has issued those waits -> the events may be re-recorded by the next collective.  An event is always
// recorded (host side) before anyone waits for it, so the work enqueued at any moment depends only on work enqueued earlier:
// no cycle can form, whatever hardware queues the 2 W streams share.
struct Fabric {
  struct Msg {
    const void* p;
    size_t bytes;
    int peer;
  };
  int W;
  std::mutex m;
  std::condition_variable cv;
  int arrived       = 0;
  unsigned long gen = 0;
  bool broken       = false;  // a rank failed or timed out: every later barrier fails at once instead of hanging the others
  std::vector<const void*> ag_src;
  std::vector<std::vector<Msg>> sends;
  std::vector<hipEvent_t> ready, done;
  explicit Fabric(int w) : W(w), ag_src(w, nullptr), sends(w), ready(w, nullptr), done(w, nullptr) {}
  ~Fabric()
  {
    for (auto e : ready)
      if (e) (void)hipEventDestroy(e);
    for (auto e : done)
      if (e) (void)hipEventDestroy(e);
  }
  int init()
  {
    for (int r = 0; r < W; ++r) {
      GXD_HIP(hipEventCreateWithFlags(&ready[r], hipEventDisableTiming));
      GXD_HIP(hipEventCreateWithFlags(&done[r], hipEventDisableTiming));
    }
    return 0;
  }
  void abandon()
  {
    std::lock_guard<std::mutex> g(m);
    broken = true;
    cv.notify_all();
  }
  int barrier()
  {
    std::unique_lock<std::mutex> g(m);
    if (broken) return fail(GX_EINTERNAL, "gxd loopback: a peer rank failed");
    const unsigned long my = gen;
    if (++arrived == W) {
      arrived = 0;
      ++gen;
      cv.notify_all();
      return 0;
    }
    const bool ok = cv.wait_for(g, std::chrono::seconds(120), [&] { return gen != my || broken; });
    if (!ok || broken) {
      broken = true;
      cv.notify_all();
      return fail(GX_EINTERNAL, ok ? "gxd loopback: a peer rank failed" : "gxd loopback: barrier timed out (a rank did not join the collective)");
    }
    return 0;
  }
};

struct LoopbackTransport final : Transport {
  std::shared_ptr<Fabric> f;
  int rank = 0;
  bool grouping = false;
  std::vector<Fabric::Msg> my_recvs;
  struct Bail {  // leaving a collective early (an error) must not leave the peers waiting at the barrier
    Fabric* f;
    bool armed = true;
    ~Bail()
    {
      if (armed) f->abandon();
    }
  };
  // after the copies of a collective are enqueued on `s`: done/wait/barrier tail shared by both collectives
  int finish(hipStream_t s)
  {
    GXD_HIP(hipEventRecord(f->done[rank], s));
    GXD_GX(f->barrier());
    for (int r = 0; r < f->W; ++r)
      if (r != rank) GXD_HIP(hipStreamWaitEvent(s, f->done[r], 0));
    GXD_GX(f->barrier());
    return 0;
  }
  int allgather(const void* mine, void* all, size_t bytes, hipStream_t s) override
  {
    Bail bail{f.get()};
    f->ag_src[rank] = mine;
    GXD_HIP(hipEventRecord(f->ready[rank], s));
    GXD_GX(f->barrier());
    for (int r = 0; r < f->W; ++r) {
      if (r != rank) GXD_HIP(hipStreamWaitEvent(s, f->ready[r], 0));
      if (bytes) GXD_HIP(hipMemcpyAsync(static_cast<char*>(all) + (size_t)r * bytes, f->ag_src[r], bytes, hipMemcpyDeviceToDevice, s));
    }
    GXD_GX(finish(s));
    bail.armed = false;
    return 0;
  }
  int group_start() override
  {
    grouping = true;
    f->sends[rank].clear();
    my_recvs.clear();
    return 0;
  }
  int send(const void* p, size_t bytes, int peer, hipStream_t) override
  {
    if (!grouping || peer < 0 || peer >= f->W || peer == rank) return fail(GX_EINVAL, "gxd loopback: send outside a group / bad peer");
    f->sends[rank].push_back({p, bytes, peer});
    return 0;
  }
  int recv(void* p, size_t bytes, int peer, hipStream_t) override
  {
    if (!grouping || peer < 0 || peer >= f->W || peer == rank) return fail(GX_EINVAL, "gxd loopback: recv outside a group / bad peer");
    my_recvs.push_back({p, bytes, peer});
    return 0;
  }
  int group_end(hipStream_t s) override
  {
    Bail bail{f.get()};
    grouping = false;
    GXD_HIP(hipEventRecord(f->ready[rank], s));
    GXD_GX(f->barrier());
    std::vector<size_t> next(f->W, 0);  // per peer: its next unmatched send to me (posting order, as RCCL matches them)
    std::vector<char> waited(f->W, 0);
    for (const auto& rv : my_recvs) {
      const auto& ps = f->sends[rv.peer];
      size_t& i      = next[rv.peer];
      while (i < ps.size() && ps[i].peer != rank) ++i;
      if (i == ps.size()) return fail(GX_EINTERNAL, "gxd loopback: a receive has no matching send (rank " + std::to_string(rv.peer) + " -> " + std::to_string(rank) + ")");
      if (ps[i].bytes != rv.bytes)
        return fail(GX_EINTERNAL, "gxd loopback: send / receive sizes differ (rank " + std::to_string(rv.peer) + " -> " + std::to_string(rank) + ": " +
                                    std::to_string(ps[i].bytes) + " vs " + std::to_string(rv.bytes) + " bytes)");
      if (!waited[rv.peer]) {
        GXD_HIP(hipStreamWaitEvent(s, f->ready[rv.peer], 0));
        waited[rv.peer] = 1;
      }
      if (rv.bytes) GXD_HIP(hipMemcpyAsync(const_cast<void*>(rv.p), ps[i].p, rv.bytes, hipMemcpyDeviceToDevice, s));
      ++i;
    }
    for (int r = 0; r < f->W; ++r) {  // a send nobody received would hang an RCCL group: report it
      if (r == rank) continue;
      const auto& ps = f->sends[r];
      size_t i       = next[r];
      while (i < ps.size() && ps[i].peer != rank) ++i;
      if (i != ps.size()) return fail(GX_EINTERNAL, "gxd loopback: a send has no matching receive (rank " + std::to_string(r) + " -> " + std::to_string(rank) + ")");
    }
    GXD_GX(finish(s));
    bail.armed = false;
    return 0;
  }
  void abort() override { f->abandon(); }
};

}  // namespace

struct gxd_comm {
  std::unique_ptr<Transport> tp;  // RCCL, or the loopback fabric (world == 1 without a peer: a one-rank fabric)
  int rank = 0, world = 1;
  hipStream_t xs = nullptr;  // exchange stream
  std::vector<hipEvent_t> evP;  // partition of chunk c done (caller's stream)
  hipEvent_t evQ = nullptr, evX = nullptr;
  long long* pinned = nullptr;  // host staging of count matrices / small reads
  size_t pinned_elems = 0;
  Arena arena;
  double ms[3] = {0, 0, 0};
  // Buffers of destroyed join tables, kept for the next build: hipFree / hipMalloc of multi-gigabyte blocks are
  // device-synchronising and were seen to take SECONDS under a build / probe / destroy loop
  // (profiles/r3_run14_alloc_probe.txt: 1.87 s for one 4.3 GB hipMalloc).  Released with the communicator.
  std::vector<std::pair<void*, size_t>> pool;
  int pool_get(size_t bytes, void** out)
  {
    int best = -1;
    for (int i = 0; i < (int)pool.size(); ++i)
      if (pool[i].second >= bytes && pool[i].second <= bytes + bytes / 4 + (1u << 20) && (best < 0 || pool[i].second < pool[best].second)) best = i;
    if (best >= 0) {
      *out = pool[best].first;
      pool.erase(pool.begin() + best);
      return 0;
    }
    hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e != hipSuccess && !pool.empty()) {  // out of memory with cached buffers around: give them back and retry
      for (auto& b : pool) (void)hipFree(b.first);
      pool.clear();
      e = hipMalloc(out, bytes ? bytes : 1);
    }
    return (int)e;
  }
  void pool_put(void* p, size_t bytes)
  {
    if (p) pool.emplace_back(p, bytes);
  }
};

struct gxd_join {
  gxd_comm* comm = nullptr;
  int key_size = 8;
  bool single = false;  // world == 1 without forced exchange: local rows ARE global rows
  void* table = nullptr;
  size_t table_bytes = 0;
  void* keys_keep = nullptr;      // received build keys (the table stores rows, the keys live in its slots; kept for re-builds only)
  int32_t* rows = nullptr;        // received (int32 local row at the source), chunk-major / source-minor segments
  int64_t nrows = 0;
  std::vector<long long> seg_counts, seg_bases;  // per segment: rows, first global row of the source rank's shard
  void* segtab = nullptr;                          // device copy: [nseg + 1] starts | [nseg] bases
  size_t rows_bytes = 0, keys_bytes = 0;           // sizes of the pooled buffers (returned to the communicator's pool)
  int enc_shift = 0;                               // > 0: the table slots hold (source rank << enc_shift) | row at the source --
  void* bases_dev = nullptr;                       //      decoded by one streaming pass with these bases, no gather
};

// (rank << shift) | row fits an int32 >= 0 when shift = 31 - ceil(log2(world)) and row < 2^shift
static int g_row_bits = 0;  // gxd_test_set_row_bits
inline int code_shift(int world)
{
  int b = 0;
  while ((1 << b) < world) ++b;
  const int s = 31 - b;
  return g_row_bits > 0 && g_row_bits < s ? g_row_bits : s;
}

namespace {

constexpr int MAX_WORLD = 16;  // gx_partition_rows splits into <= 16 groups in one pass (PJ_MAX_SPLIT + 1)
double g_slot_scale     = 0.0;  // gxd_test_set_slot_scale

int elem_size(int dtype) { return gx_dtype_size(dtype); }

int ensure_events(gxd_comm* c, int chunks)
{
  while ((int)c->evP.size() < chunks) {
    hipEvent_t e;
    GXD_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->evP.push_back(e);
  }
  return 0;
}

int ensure_pinned(gxd_comm* c, size_t elems)
{
  if (elems > c->pinned_elems) {
    if (c->pinned) GXD_HIP(hipHostFree(c->pinned));
    c->pinned = nullptr;
    GXD_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->pinned), elems * sizeof(long long), hipHostMallocDefault));
    c->pinned_elems = elems;
  }
  return 0;
}

// all-gather of `count` int64 per rank (device -> device) followed by a copy to pinned host memory; waits for it.
int allgather_i64_host(gxd_comm* c, const long long* mine_dev, long long* all_dev, int count, long long* host)
{
  GXD_GX(c->tp->allgather(mine_dev, all_dev, sizeof(long long) * (size_t)count, c->xs));
  GXD_HIP(hipMemcpyAsync(host, all_dev, sizeof(long long) * count * c->world, hipMemcpyDeviceToHost, c->xs));
  GXD_HIP(hipEventRecord(c->evQ, c->xs));
  GXD_HIP(hipEventSynchronize(c->evQ));
  return 0;
}

// first global row of every rank's shard (exclusive scan of the shard sizes) and the largest shard: what chunk grids and row
// encodings are derived from, so that every rank walks the same number of exchange rounds (one tiny all-gather)
int shard_sizes(gxd_comm* c, int64_t n, std::vector<long long>& bases, int64_t* nmax);
int shard_bases(gxd_comm* c, int64_t n, std::vector<long long>& bases)
{
  int64_t nmax;
  return shard_sizes(c, n, bases, &nmax);
}
int shard_sizes(gxd_comm* c, int64_t n, std::vector<long long>& bases, int64_t* nmax)
{
  void *d1, *d2;
  GXD_GX(c->arena.get(Arena::MISC_E, sizeof(long long), &d1));
  GXD_GX(c->arena.get(Arena::MISC_F, sizeof(long long) * c->world, &d2));
  GXD_GX(ensure_pinned(c, (size_t)c->world * (c->world + 2) * 16));
  long long mine = n;
  GXD_HIP(hipMemcpyAsync(d1, &mine, sizeof(mine), hipMemcpyHostToDevice, c->xs));
  GXD_HIP(hipStreamSynchronize(c->xs));  // `mine` is a stack variable
  GXD_GX(allgather_i64_host(c, static_cast<long long*>(d1), static_cast<long long*>(d2), 1, c->pinned));
  bases.assign(c->world, 0);
  long long run = 0, mx = 0;
  for (int r = 0; r < c->world; ++r) {
    bases[r] = run;
    run += c->pinned[r];
    mx = std::max(mx, c->pinned[r]);
  }
  *nmax = mx;
  return 0;
}

struct Exchange {
  int64_t total = 0;                   // rows received
  int chunks    = 0;
  std::vector<long long> seg_counts;   // [chunks * world] rows received per (chunk, source rank), in buffer order
  std::vector<long long> chunk_start;  // [chunks + 1] first received row of each chunk
  std::vector<long long> send_off, send_cnt;   // [chunks * world] where the rows for rank r sit in chunk k's partition output, how many
  std::vector<const int32_t*> send_rows;       // [chunks] that chunk's row map (the permutation a payload column has to follow)
  int64_t crows = 0;                           // rows per chunk -- the SAME on every rank (from the largest shard)
};

// how a row is identified on the wire (the int32 that travels next to its key)
struct RowCode {
  int shift = 0;       // 0: the row number inside the sender's shard.  > 0: (sender rank << shift) | row ...
  bool in_chunk = false;  // ... counted inside the sender's CHUNK (probe side: chunk k of every rank starts at k * crows)
};


// Partition `keys` chunk by chunk into `world` destination groups (mode 0 hash / mode 1 range, gx_partition_rows_at), exchange
// every chunk as soon as it is partitioned, and call on_chunk(c, first received row, rows) when the chunk's rows have been
// POSTED on the exchange stream (evX is recorded behind them; the callee makes its stream wait for it).
int partition_exchange(gxd_comm* c, int dtype, const void* keys, int64_t n, int64_t nmax, int mode, const void* splitters_host,
                       bool want_rows, int chunks, int64_t max_chunk_rows, RowCode code, hipStream_t stream, Exchange* ex,
                       const std::function<int(int, int64_t, int64_t)>& on_chunk, int max_chunks = 0)
{
  const int W  = c->world;
  const int es = elem_size(dtype);
  // the chunk grid comes from the LARGEST shard, so it is the same on every rank (a rank with fewer rows has short or empty
  // chunks): the grouped send / recv rounds below are collective
  if (chunks <= 0) chunks = 8;
  if (nmax < (int64_t)chunks * (1 << 20)) chunks = (int)std::max<int64_t>(1, nmax >> 20);  // small shards: fewer, larger chunks
  if (max_chunk_rows > 0 && (nmax + chunks - 1) / chunks > max_chunk_rows - 16384) chunks = (int)((nmax + max_chunk_rows - 16385) / (max_chunk_rows - 16384));
  const int64_t crows = std::max<int64_t>(16384, ((nmax + chunks - 1) / chunks + 16383) / 16384 * 16384);  // whole scatter tiles per chunk
  chunks              = nmax > 0 ? (int)((nmax + crows - 1) / crows) : 1;
  // (checked BEFORE anything is enqueued or on_chunk writes per-chunk state; nmax is common to all ranks, so all ranks agree)
  if (max_chunks > 0 && chunks > max_chunks) return fail(GX_EINVAL, "gxd: the shard needs " + std::to_string(chunks) + " chunks, at most " + std::to_string(max_chunks) + " are supported");
  ex->chunks          = chunks;
  ex->crows           = crows;
  ex->seg_counts.assign((size_t)chunks * W, 0);
  ex->chunk_start.assign((size_t)chunks + 1, 0);
  ex->send_off.assign((size_t)chunks * W, 0);
  ex->send_cnt.assign((size_t)chunks * W, 0);
  ex->send_rows.assign((size_t)chunks, nullptr);
  GXD_GX(ensure_events(c, chunks));
  GXD_GX(ensure_pinned(c, (size_t)W * (W + 1) + 64));
  // Speculative partition passes (gx_partition_rows_spec_at): ONE read of the keys per chunk, group g of chunk k lands in the
  // fixed slot (k * W + g) * cap; a chunk whose keys are skewed beyond the slot margin is partitioned again, exactly.
  int64_t cap = W == 1 ? crows : crows / W + crows / (3 * W) + 4096;
  if (g_slot_scale > 0) cap = (int64_t)((double)cap * g_slot_scale);  // tests: slots too small on purpose -> the exact re-partition runs
  cap         = std::min<int64_t>((cap + 31) / 32 * 32, (crows + 31) / 32 * 32);
  if (cap < 32) cap = 32;
  const int64_t stride = cap * W;  // elements per chunk in the partition buffers
  void *pk, *prow = nullptr, *offs, *alloffs, *tmp, *cnt;
  GXD_GX(c->arena.get(Arena::PART_KEYS, (size_t)stride * chunks * es, &pk));
  if (want_rows) GXD_GX(c->arena.get(Arena::PART_ROWS, (size_t)stride * chunks * 4, &prow));
  GXD_GX(c->arena.get(Arena::OFFS, sizeof(long long) * (size_t)chunks * (W + 1), &offs));
  GXD_GX(c->arena.get(Arena::ALLOFFS, sizeof(long long) * (size_t)W * W, &alloffs));
  GXD_GX(c->arena.get(Arena::CNT, sizeof(long long) * (size_t)W, &cnt));
  size_t tb = 0;
  GXD_GX(gx_partition_rows_spec_at(dtype, keys, crows, 0, mode, W, splitters_host, cap, pk, static_cast<int32_t*>(prow),
                                   static_cast<int64_t*>(offs), nullptr, &tb, stream));
  GXD_GX(c->arena.get(Arena::TMP, tb ? tb : 1, &tmp));
  const double t0 = now_ms();
  // ---- every partition pass is enqueued before the first wait
  auto chunk_rows = [&](int k, int64_t& c0) -> int64_t {
    c0 = std::min<int64_t>((int64_t)k * crows, n);
    return std::min<int64_t>(crows, n - c0);
  };
  auto row_base = [&](int64_t c0) -> int32_t {
    if (code.shift == 0) return (int32_t)c0;
    return (int32_t)(((uint32_t)c->rank << code.shift) + (uint32_t)(code.in_chunk ? 0 : c0));
  };
  for (int k = 0; k < chunks; ++k) {
    int64_t c0;
    const int64_t nc = chunk_rows(k, c0);
    GXD_GX(gx_partition_rows_spec_at(dtype, static_cast<const char*>(keys) + c0 * es, nc, row_base(c0), mode, W, splitters_host, cap,
                                     static_cast<char*>(pk) + (size_t)k * stride * es,
                                     want_rows ? static_cast<int32_t*>(prow) + (size_t)k * stride : nullptr,
                                     static_cast<int64_t*>(offs) + (size_t)k * (W + 1), tmp, &tb, stream));
    GXD_HIP(hipEventRecord(c->evP[k], stream));
  }
  c->ms[0] = now_ms() - t0;
  // ---- exchange, chunk by chunk
  int64_t rpos      = 0;
  int64_t recv_cap  = 0;  // elements the receive buffers hold
  void *rk = nullptr, *rr = nullptr;
  {
    const int64_t guess = n + n / 4 + 65536;
    GXD_GX(c->arena.get(Arena::RECV_KEYS, (size_t)guess * es, &rk));
    if (want_rows) GXD_GX(c->arena.get(Arena::RECV_ROWS, (size_t)guess * 4, &rr));
    recv_cap = std::min<int64_t>((int64_t)(c->arena.cap[Arena::RECV_KEYS] / es), want_rows ? (int64_t)(c->arena.cap[Arena::RECV_ROWS] / 4) : INT64_MAX);
  }
  double waited = 0;
  std::vector<long long> soff(W), scnt(W);
  for (int k = 0; k < chunks; ++k) {
    int64_t c0;
    const int64_t nc = chunk_rows(k, c0);
    GXD_HIP(hipStreamWaitEvent(c->xs, c->evP[k], 0));
    const double w0 = now_ms();
    // my fill counts of this chunk (and the overflow flag)
    long long* H = c->pinned;
    GXD_HIP(hipMemcpyAsync(H, static_cast<long long*>(offs) + (size_t)k * (W + 1), sizeof(long long) * (W + 1), hipMemcpyDeviceToHost, c->xs));
    GXD_HIP(hipEventRecord(c->evQ, c->xs));
    GXD_HIP(hipEventSynchronize(c->evQ));
    const char* kbase    = static_cast<const char*>(pk) + (size_t)k * stride * es;
    const int32_t* rbase = want_rows ? static_cast<const int32_t*>(prow) + (size_t)k * stride : nullptr;
    if (H[W] != 0) {  // skewed beyond the margin: the exact two-pass partition of this chunk, on the exchange stream
      void *ek, *er = nullptr, *t3, *eo;
      GXD_GX(c->arena.get(Arena::EXACT_KEYS, (size_t)crows * es, &ek));
      if (want_rows) GXD_GX(c->arena.get(Arena::EXACT_ROWS, (size_t)crows * 4, &er));
      GXD_GX(c->arena.get(Arena::EXACT_OFFS, sizeof(long long) * (W + 1), &eo));  // (a slot of its own: the callers keep data in MISC_*)
      size_t tb3 = 0;
      GXD_GX(gx_partition_rows_at(dtype, keys, crows, 0, mode, W, splitters_host, ek, static_cast<int32_t*>(er), static_cast<int64_t*>(eo), nullptr,
                                  &tb3, reinterpret_cast<gx_stream_t>(c->xs)));
      GXD_GX(c->arena.get(Arena::TMP3, tb3 ? tb3 : 1, &t3));
      GXD_GX(gx_partition_rows_at(dtype, static_cast<const char*>(keys) + c0 * es, nc, row_base(c0), mode, W, splitters_host, ek,
                                  static_cast<int32_t*>(er), static_cast<int64_t*>(eo), t3, &tb3, reinterpret_cast<gx_stream_t>(c->xs)));
      GXD_HIP(hipMemcpyAsync(H, eo, sizeof(long long) * (W + 1), hipMemcpyDeviceToHost, c->xs));
      GXD_HIP(hipEventRecord(c->evQ, c->xs));
      GXD_HIP(hipEventSynchronize(c->evQ));
      for (int r = 0; r < W; ++r) {
        soff[r] = H[r];
        scnt[r] = H[r + 1] - H[r];
      }
      kbase = static_cast<const char*>(ek);
      rbase = static_cast<const int32_t*>(er);
    } else {
      for (int r = 0; r < W; ++r) {
        soff[r] = (long long)r * cap;
        scnt[r] = H[r];
      }
    }
    for (int r = 0; r < W; ++r) {
      ex->send_off[(size_t)k * W + r] = soff[r];
      ex->send_cnt[(size_t)k * W + r] = scnt[r];
    }
    ex->send_rows[k] = rbase;
    // everybody's counts: M[r * W + j] = rows rank r sends to rank j
    for (int r = 0; r < W; ++r) H[r] = scnt[r];
    GXD_HIP(hipMemcpyAsync(cnt, H, sizeof(long long) * W, hipMemcpyHostToDevice, c->xs));
    GXD_HIP(hipStreamSynchronize(c->xs));  // H is reused right below
    GXD_GX(allgather_i64_host(c, static_cast<long long*>(cnt), static_cast<long long*>(alloffs), W, c->pinned));
    waited += now_ms() - w0;
    const long long* M = c->pinned;
    int64_t rtotal     = 0;
    for (int r = 0; r < W; ++r) rtotal += M[r * W + c->rank];
    if (rpos + rtotal > recv_cap) {  // (rare: a skewed split) grow, keeping what has arrived
      GXD_HIP(hipStreamSynchronize(c->xs));
      const int64_t want = rpos + rtotal + (rpos + rtotal) / 4;
      GXD_GX(c->arena.grow(Arena::RECV_KEYS, (size_t)want * es, (size_t)rpos * es, &rk));
      if (want_rows) GXD_GX(c->arena.grow(Arena::RECV_ROWS, (size_t)want * 4, (size_t)rpos * 4, &rr));
      recv_cap = want;
    }
    ex->chunk_start[k] = rpos;
    GXD_GX(c->tp->group_start());
    for (int r = 0; r < W; ++r) {
      const int64_t so = soff[r], sc = scnt[r];  // what I send to r
      const int64_t rc = M[r * W + c->rank];     // what r sends to me
      ex->seg_counts[(size_t)k * W + r] = rc;
      const char* sk = kbase + so * es;
      char* dk       = static_cast<char*>(rk) + rpos * es;
      if (r == c->rank) {  // my own group: a device-local copy on the exchange stream
        // (a copy KERNEL: hipMemcpyAsync falls to the SDMA engines while other queues are busy)
        if (sc) GXD_GX(gx_copy_bytes(sk, dk, (size_t)sc * es, reinterpret_cast<gx_stream_t>(c->xs)));
        if (want_rows && sc)
          GXD_GX(gx_copy_bytes(rbase + so, static_cast<int32_t*>(rr) + rpos, (size_t)sc * 4, reinterpret_cast<gx_stream_t>(c->xs)));
      } else {
        if (sc) GXD_GX(c->tp->send(sk, (size_t)sc * es, r, c->xs));
        if (rc) GXD_GX(c->tp->recv(dk, (size_t)rc * es, r, c->xs));
        if (want_rows) {
          if (sc) GXD_GX(c->tp->send(rbase + so, (size_t)sc * 4, r, c->xs));
          if (rc) GXD_GX(c->tp->recv(static_cast<int32_t*>(rr) + rpos, (size_t)rc * 4, r, c->xs));
        }
      }
      rpos += rc;
    }
    GXD_GX(c->tp->group_end(c->xs));
    GXD_HIP(hipEventRecord(c->evX, c->xs));
    if (on_chunk) {
      int rc2 = on_chunk(k, ex->chunk_start[k], rpos - ex->chunk_start[k]);
      if (rc2) return rc2;
    }
  }
  ex->chunk_start[chunks] = rpos;
  ex->total               = rpos;
  c->ms[1]                = waited;
  return 0;
}

template <typename F>
int with_tmp(gxd_comm* c, int slot, F&& f);
template <typename F>
int with_tmp(gxd_comm* c, int slot, F&& f)
{
  size_t b = 0;
  GXD_GX(f(nullptr, &b));
  void* t;
  GXD_GX(c->arena.get(slot, b ? b : 1, &t));
  GXD_GX(f(t, &b));
  return 0;
}

// device table for gx_gather_global_rows_dev: [nseg + 1] int64 starts, then [nseg] int64 bases
int upload_segtab(gxd_comm* c, const std::vector<long long>& counts, const std::vector<long long>& bases, void* dst, hipStream_t s)
{
  const size_t nseg = counts.size();
  std::vector<long long> h(2 * nseg + 1);
  long long run = 0;
  for (size_t i = 0; i < nseg; ++i) {
    h[i] = run;
    run += counts[i];
  }
  h[nseg] = run;
  for (size_t i = 0; i < nseg; ++i) h[nseg + 1 + i] = bases[i];
  GXD_HIP(hipMemcpyAsync(dst, h.data(), h.size() * sizeof(long long), hipMemcpyHostToDevice, s));
  GXD_HIP(hipStreamSynchronize(s));  // `h` dies here
  (void)c;
  return 0;
}


// ------------------------------------------------------------------------------------------------------------ sort, fused
// The exchange BETWEEN the sort's two partition levels (gx.h gx_sortx_*).  Every rank runs level 0 on
// its shard -- 256 bins on digit positions all ranks agree on -- whole bins are dealt to ranks in contiguous runs balanced by
// the all-gathered level-0 histogram, a rank sends ONE span of its level-0 buffer per peer (the padded (bin, input range) slots
// of the peer's bins, as they lie), and the receiver's level 1 + cell sort run over the regions of what arrived.  Against the
// sample-sort path below this saves the range-partition pass on the sender (one full read + write of the shard) and level 0
// on the receiver.  Collective decisions use all-gathered values only, so every rank takes the same branch; a rank whose
// device-side checks fail reports it in the final status all-gather and ALL ranks take the sample-sort path instead.
// Returns 0 (sorted), 1 (not applicable / a check failed: the caller runs the sample-sort path), or an error.
int g_sort_mode = 0;  // gxd_test_set_sort_mode: 0 auto (fused for integer keys from 2^25 rows per rank), 1 never, 2 fused from 2^21 rows
constexpr int SX_BINS = 256, SX_RANGES = 8;

int sort_fused(gxd_comm* c, int dtype, const void* keys, int64_t n, gxd_alloc_fn alloc, void* actx, void** out_keys, int64_t* out_n,
               hipStream_t stream, Trace& tr, void** spare, size_t* spare_bytes)
{
  *spare       = nullptr;
  *spare_bytes = 0;
  const int W  = c->world;
  const int es = elem_size(dtype);
  if (g_sort_mode == 1 || !(dtype == GX_INT64 || dtype == GX_UINT64 || dtype == GX_INT32 || dtype == GX_UINT32)) return 1;
  gx_stream_t gstream = reinterpret_cast<gx_stream_t>(stream);
  std::vector<long long> bases;
  int64_t nmax = n;
  GXD_HIP(hipStreamSynchronize(stream));
  GXD_GX(shard_sizes(c, n, bases, &nmax));
  long long ntotal = 0;
  for (int r = 0; r < W; ++r) ntotal += c->pinned[r];
  const int64_t min_rows = g_sort_mode == 2 ? (1ll << 21) : (1ll << 25);
  if (ntotal / W < min_rows || nmax > 0x60000000ll) return 1;
  // a rank is prepared to receive 1.5 x the mean shard (level-0 bins are dealt by the exact histogram: the imbalance is at most
  // one bin); beyond that -- one bin holding a large share of all keys -- the sample-sort path splits inside the bin
  // (+ 2^22: the padding of small level-0 buffers -- two sample steps per slot -- is not proportional to the shard)
  const int64_t recv_max = std::min<int64_t>((int64_t)(ntotal / W) * 3 / 2 + (1 << 22), 0x7FFF0000ll);
  size_t tb = 0;
  GXD_GX(gx_sortx_sample(dtype, nullptr, n, recv_max, nullptr, &tb, gstream));
  void* tmp;
  GXD_GX(c->arena.get(Arena::TMP2, tb, &tmp));
  GXD_GX(gx_sortx_sample(dtype, keys, n, recv_max, tmp, &tb, gstream));
  // ---- digit positions: the OR of every rank's varying-bit masks
  constexpr int TAB = 2 * SX_RANGES * SX_BINS + 2;  // per rank: cur0 | slot0 | slot_total | state
  void *d_mine, *d_all;
  GXD_GX(c->arena.get(Arena::MISC_A, sizeof(long long) * TAB, &d_mine));
  GXD_GX(c->arena.get(Arena::MISC_B, sizeof(long long) * TAB * W, &d_all));
  GXD_GX(ensure_pinned(c, (size_t)TAB * W + 64));
  uint64_t masks[2] = {0, 0};
  if (n > 0) GXD_GX(gx_sortx_masks(tmp, masks, gstream));
  GXD_HIP(hipMemcpyAsync(d_mine, masks, sizeof(masks), hipMemcpyHostToDevice, c->xs));
  GXD_HIP(hipStreamSynchronize(c->xs));
  GXD_GX(allgather_i64_host(c, static_cast<long long*>(d_mine), static_cast<long long*>(d_all), 2, c->pinned));
  uint64_t gm[2] = {0, 0};
  for (int r = 0; r < W; ++r) {
    gm[0] |= (uint64_t)c->pinned[2 * r];
    gm[1] |= (uint64_t)c->pinned[2 * r + 1];
  }
  if ((gm[0] & gm[1]) == 0) return 1;  // every key of every rank is the same value: nothing to partition on
  tr.mark("sample + masks");
  // ---- level 0, then everybody's slot tables
  GXD_GX(gx_sortx_level0(dtype, keys, n, recv_max, gm, tmp, gstream));
  std::vector<uint32_t> cur0(SX_RANGES * SX_BINS, 0), slot0(SX_RANGES * SX_BINS, 0);
  uint32_t slot_total = 0;
  int32_t state       = 3;
  if (n > 0) GXD_GX(gx_sortx_tables(tmp, cur0.data(), slot0.data(), &slot_total, &state, gstream));
  tr.mark("level 0");
  {
    std::vector<long long> mine(TAB);
    for (int i = 0; i < SX_RANGES * SX_BINS; ++i) {
      mine[i]                       = cur0[i];
      mine[SX_RANGES * SX_BINS + i] = slot0[i];
    }
    mine[TAB - 2] = slot_total;
    mine[TAB - 1] = state;
    GXD_HIP(hipMemcpyAsync(d_mine, mine.data(), sizeof(long long) * TAB, hipMemcpyHostToDevice, c->xs));
    GXD_HIP(hipStreamSynchronize(c->xs));
  }
  GXD_GX(allgather_i64_host(c, static_cast<long long*>(d_mine), static_cast<long long*>(d_all), TAB, c->pinned));
  const long long* T = c->pinned;  // T[r * TAB + ...]
  auto CUR  = [&](int r, int rg, int b) { return (uint32_t)T[(size_t)r * TAB + rg * SX_BINS + b]; };
  auto SLOT = [&](int r, int rg, int b) { return (uint32_t)T[(size_t)r * TAB + SX_RANGES * SX_BINS + rg * SX_BINS + b]; };
  auto STOT = [&](int r) { return (uint32_t)T[(size_t)r * TAB + TAB - 2]; };
  for (int r = 0; r < W; ++r)
    if (T[(size_t)r * TAB + TAB - 1] != 3) return 1;  // some rank's level 0 did not hold (its sample missed an outlier, ...)
  // ---- whole bins to ranks, contiguous, balanced by the exact histogram
  std::vector<long long> hist(SX_BINS, 0);
  for (int b = 0; b < SX_BINS; ++b)
    for (int r = 0; r < W; ++r)
      for (int rg = 0; rg < SX_RANGES; ++rg) hist[b] += CUR(r, rg, b);
  std::vector<int> b0(W + 1, SX_BINS);
  {
    long long run = 0;
    int d         = 0;
    b0[0]         = 0;
    for (int b = 0; b < SX_BINS; ++b) {
      // bin b goes to the rank whose share of the key space its midpoint falls into
      const long long mid = run + hist[b] / 2;
      int want            = (int)std::min<long long>(W - 1, ntotal > 0 ? mid * W / ntotal : 0);
      while (d < want) b0[++d] = b;
      run += hist[b];
    }
    while (d < W) b0[++d] = SX_BINS;
  }
  // span of rank r's level-0 buffer that holds the bins of rank d (slots are bin-major: the slots of a bin are neighbours)
  auto span_lo = [&](int r, int d) { return b0[d] < SX_BINS ? SLOT(r, 0, b0[d]) : STOT(r); };
  auto span_hi = [&](int r, int d) { return b0[d + 1] < SX_BINS ? SLOT(r, 0, b0[d + 1]) : STOT(r); };
  for (int d = 0; d < W; ++d) {  // what rank d would have to receive: the same sum on every rank
    long long rv = 0, rows = 0;
    for (int r = 0; r < W; ++r) {
      if (r != d) rv += (long long)span_hi(r, d) - span_lo(r, d);
      for (int b = b0[d]; b < b0[d + 1]; ++b)
        for (int rg = 0; rg < SX_RANGES; ++rg) rows += CUR(r, rg, b);
    }
    if (rv > recv_max || rows > recv_max) return 1;  // (rows: what rank d ends up with, its own bins included)
  }
  const int me = c->rank;
  int64_t own_rows = 0, total_rows = 0;
  char* level0 = static_cast<char*>(gx_sortx_level0_buffer(dtype, tmp, n, recv_max, &own_rows, &total_rows));
  if (!level0) return fail(GX_EINTERNAL, "gxd_sort: gx_sortx_level0_buffer");
  // ---- the exchange: one span per peer, received behind this rank's own slots
  std::vector<long long> roff(W, 0);
  {
    long long off = own_rows;
    for (int r = 0; r < W; ++r) {
      roff[r] = off;
      if (r != me) off += (long long)span_hi(r, me) - span_lo(r, me);
    }
    if (off > total_rows) return fail(GX_EINTERNAL, "gxd_sort: receive area too small (internal bound)");
  }
  GXD_GX(c->tp->group_start());
  for (int r = 0; r < W; ++r) {
    if (r == me) continue;
    const long long slen = (long long)span_hi(me, r) - span_lo(me, r);
    const long long rlen = (long long)span_hi(r, me) - span_lo(r, me);
    if (slen > 0) GXD_GX(c->tp->send(level0 + (size_t)span_lo(me, r) * es, (size_t)slen * es, r, c->xs));
    if (rlen > 0) GXD_GX(c->tp->recv(level0 + (size_t)roff[r] * es, (size_t)rlen * es, r, c->xs));
  }
  GXD_GX(c->tp->group_end(c->xs));
  GXD_HIP(hipEventRecord(c->evX, c->xs));
  GXD_HIP(hipStreamWaitEvent(stream, c->evX, 0));
  tr.mark("exchange");
  // ---- regions of my bins: (bucket, source rank, input range), in bucket order
  std::vector<uint32_t> rs, rc, rb;
  long long nrecv = 0;
  for (int b = b0[me]; b < b0[me + 1]; ++b)
    for (int r = 0; r < W; ++r)
      for (int rg = 0; rg < SX_RANGES; ++rg) {
        const uint32_t cnt = CUR(r, rg, b);
        if (!cnt) continue;
        const long long st = r == me ? (long long)SLOT(r, rg, b) : roff[r] + ((long long)SLOT(r, rg, b) - span_lo(r, me));
        rs.push_back((uint32_t)st);
        rc.push_back(cnt);
        rb.push_back((uint32_t)b);
        nrecv += cnt;
      }
  // From here on the exchange has been posted: a rank that left on its own would leave its peers blocked in the status
  // all-gather below (for ever over RCCL, 120 s on the loopback fabric) -- ADVICE r4.  Every failure is therefore CARRIED to
  // that all-gather: `ok` 0 = a device-side check failed (all ranks take the sample-sort path), `err` != 0 = a hard error
  // (all ranks return an error together).
  int ok = 1, err = 0;
  std::string err_what;
  void* out = nullptr;
  if (nrecv > 0) {
    out = alloc((size_t)nrecv * es, actx);
    if (!out) {
      err      = GX_EINVAL;
      err_what = "gxd_sort: allocator returned NULL";
    } else {
      const int frc = gx_sortx_finish(dtype, n, recv_max, nrecv, gm, rs.data(), rc.data(), rb.data(), (int)rs.size(), out, tmp, gstream);
      if (frc == GX_EINVAL) ok = 0;  // (more regions than the table holds, ...): reported below, all ranks take the other path
      else if (frc) {
        err      = frc;
        err_what = "gx_sortx_finish returned " + std::to_string(frc);
      }
      if (ok && !err) {
        int32_t sok   = 0;
        const int src = gx_sortx_status(tmp, &sok, gstream);
        if (src) {
          err      = src;
          err_what = "gx_sortx_status returned " + std::to_string(src);
        }
        ok = sok;
      }
    }
  }
  tr.mark("level 1 + cells");
  // ---- everybody must have succeeded
  {
    long long mine = err ? -(long long)(err > 0 ? err : -err) - 1 : ok;
    if (hipStreamSynchronize(stream) != hipSuccess && !err) mine = -(long long)GX_EINTERNAL - 1;
    // (these two HIP calls must not return early on this rank alone -- the peers would sit in the all-gather below: a failure here
    //  is reported after the collective has been entered, with whatever `mine` made it to the device -- ADVICE r5)
    const hipError_t e1 = hipMemcpyAsync(d_mine, &mine, sizeof(mine), hipMemcpyHostToDevice, c->xs);
    const hipError_t e2 = hipStreamSynchronize(c->xs);
    GXD_GX(allgather_i64_host(c, static_cast<long long*>(d_mine), static_cast<long long*>(d_all), 1, c->pinned));
    if (e1 != hipSuccess || e2 != hipSuccess) return fail(GX_EINTERNAL, "gxd_sort: publishing this rank's status failed (HIP error)");
    for (int r = 0; r < W; ++r)
      if (c->pinned[r] < 0)
        return fail(err ? err : GX_EINTERNAL, err ? err_what : "gxd_sort: the fused path failed on rank " + std::to_string(r));
    for (int r = 0; r < W; ++r)
      if (c->pinned[r] != 1) {
        // the result buffer the allocator has already handed out is not dropped: the sample-sort path below uses it when its own
        // result fits (same shard sizes: it usually does); otherwise it stays the caller's to reclaim like every allocator buffer
        *spare       = out;
        *spare_bytes = (size_t)nrecv * es;
        return 1;
      }
  }
  *out_keys = out;
  *out_n    = nrecv;
  return 0;
}

int comm_resources(gxd_comm* c)
{
  GXD_HIP(hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking));
  GXD_HIP(hipEventCreateWithFlags(&c->evQ, hipEventDisableTiming));
  GXD_HIP(hipEventCreateWithFlags(&c->evX, hipEventDisableTiming));
  return 0;
}

}  // namespace

extern "C" {

const char* gxd_last_error(void) { return g_err.c_str(); }

int gxd_unique_id(void* id128_host)
{
  if (!id128_host) return GX_EINVAL;
  static_assert(sizeof(ncclUniqueId) == 128, "gxd_unique_id hands out 128 bytes");
  ncclUniqueId id;
  GXD_NCCL(ncclGetUniqueId(&id));
  std::memcpy(id128_host, &id, sizeof(id));
  return 0;
}

int gxd_comm_create(const void* id128_host, int world, int rank, gxd_comm** out)
{
  if (!id128_host || !out || world < 1 || rank < 0 || rank >= world) return GX_EINVAL;
  if (world > MAX_WORLD) return fail(GX_EINVAL, "gxd: at most 16 ranks (one partition pass splits into <= 16 groups)");
  auto c   = std::make_unique<gxd_comm>();
  c->rank  = rank;
  c->world = world;
  if (world > 1) {
    ncclUniqueId id;
    std::memcpy(&id, id128_host, sizeof(id));
    auto tp        = std::make_unique<RcclTransport>();
    ncclResult_t r = ncclCommInitRank(&tp->comm, world, id, rank);
    if (r != ncclSuccess) return fail(GX_EINTERNAL, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    c->tp = std::move(tp);
  } else {  // one rank: the collectives are device-local copies (a one-rank fabric) -- no RCCL communicator is needed
    auto f = std::make_shared<Fabric>(1);
    GXD_GX(f->init());
    auto tp  = std::make_unique<LoopbackTransport>();
    tp->f    = std::move(f);
    tp->rank = 0;
    c->tp    = std::move(tp);
  }
  GXD_GX(comm_resources(c.get()));
  *out = c.release();
  return 0;
}

int gxd_comm_create_loopback(int world, gxd_comm** out)
{
  if (!out || world < 1) return GX_EINVAL;
  if (world > MAX_WORLD) return fail(GX_EINVAL, "gxd: at most 16 ranks (one partition pass splits into <= 16 groups)");
  auto f = std::make_shared<Fabric>(world);
  GXD_GX(f->init());
  std::vector<std::unique_ptr<gxd_comm>> cs;
  for (int r = 0; r < world; ++r) {
    auto c   = std::make_unique<gxd_comm>();
    c->rank  = r;
    c->world = world;
    auto tp  = std::make_unique<LoopbackTransport>();
    tp->f    = f;
    tp->rank = r;
    c->tp    = std::move(tp);
    GXD_GX(comm_resources(c.get()));
    cs.push_back(std::move(c));
  }
  for (int r = 0; r < world; ++r) out[r] = cs[r].release();
  return 0;
}

int gxd_comm_destroy(gxd_comm* c)
{
  if (!c) return 0;
  (void)hipDeviceSynchronize();
  c->arena.release();
  for (auto& b : c->pool) (void)hipFree(b.first);
  c->pool.clear();
  for (auto e : c->evP) (void)hipEventDestroy(e);
  if (c->evQ) (void)hipEventDestroy(c->evQ);
  if (c->evX) (void)hipEventDestroy(c->evX);
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->xs) (void)hipStreamDestroy(c->xs);
  delete c;  // (the transport goes with it: ncclCommDestroy / the last rank's reference to the loopback fabric)
  return 0;
}

int gxd_comm_abort(gxd_comm* c)
{
  if (!c || !c->tp) return GX_EINVAL;
  c->tp->abort();
  return 0;
}

void gxd_test_set_slot_scale(double scale) { g_slot_scale = scale; }
void gxd_test_set_row_bits(int bits) { g_row_bits = bits; }
void gxd_test_set_sort_mode(int mode) { g_sort_mode = mode; }
int gxd_comm_rank(const gxd_comm* c) { return c ? c->rank : -1; }
int gxd_comm_world(const gxd_comm* c) { return c ? c->world : -1; }
int gxd_last_timing(const gxd_comm* c, double* ms3_host)
{
  if (!c || !ms3_host) return GX_EINVAL;
  for (int i = 0; i < 3; ++i) ms3_host[i] = c->ms[i];
  return 0;
}

// ------------------------------------------------------------------------------------------------------------ sort
int gxd_sort(gxd_comm* c, int dtype, const void* keys, int64_t n, int chunks, int force_exchange, gxd_alloc_fn alloc, void* actx,
             void** out_keys, int64_t* out_n, gx_stream_t gstream)
{
  if (!c || !alloc || !out_keys || !out_n || n < 0 || (n > 0 && !keys)) return GX_EINVAL;
  const int es = elem_size(dtype);
  if (es != 4 && es != 8) return GX_EDTYPE;
  hipStream_t stream = reinterpret_cast<hipStream_t>(gstream);
  const double t0    = now_ms();
  Trace tr("gxd_sort");
  const int W        = c->world;
  *out_keys          = nullptr;
  *out_n             = 0;
  if (W == 1 && !force_exchange) {
    if (n == 0) return 0;
    void* out = alloc((size_t)n * es, actx);
    if (!out) return fail(GX_EINVAL, "gxd_sort: allocator returned NULL");
    GXD_GX(with_tmp(c, Arena::TMP2, [&](void* t, size_t* b) { return gx_sort_keys(dtype, keys, out, n, 0, t, b, gstream); }));
    GXD_HIP(hipStreamSynchronize(stream));
    *out_keys = out;
    *out_n    = n;
    c->ms[2]  = now_ms() - t0;
    return 0;
  }
  void* spare        = nullptr;  // a result buffer the fused attempt had already asked the allocator for (see sort_fused)
  size_t spare_bytes = 0;
  {  // the exchange between the sort's own two partition levels, where it applies (integer keys, large shards)
    const int frc = sort_fused(c, dtype, keys, n, alloc, actx, out_keys, out_n, stream, tr, &spare, &spare_bytes);
    if (frc == 0) {
      GXD_HIP(hipStreamSynchronize(stream));
      c->ms[2] = now_ms() - t0;
      c->ms[0] = -1.0;  // (marks the fused path in gxd_last_timing: no separate partition pass)
      return 0;
    }
    if (frc != 1) return frc;
    *out_keys = nullptr;
    *out_n    = 0;
  }
  // ---- splitters from an evenly strided sample of every shard (collectives/sort.py: sample -> allgather -> boundaries)
  constexpr int S = 1024;
  void *samp, *allsamp, *sorted;
  GXD_GX(c->arena.get(Arena::SAMPLE, (size_t)S * es, &samp));
  GXD_GX(c->arena.get(Arena::ALLSAMPLE, (size_t)S * es * W, &allsamp));
  GXD_GX(c->arena.get(Arena::MISC_A, (size_t)S * es * W, &sorted));
  GXD_HIP(hipStreamSynchronize(stream));  // the caller's keys are ready from here on for the exchange stream as well
  if (n >= S) {
    const int64_t stride = (n - 1) / (S - 1);
    GXD_HIP(hipMemcpy2DAsync(samp, (size_t)es, keys, (size_t)stride * es, (size_t)es, S, hipMemcpyDeviceToDevice, c->xs));
  } else {  // tiny or empty shard: cyclic copies of what there is; an empty shard contributes all-ones bit patterns
    std::vector<char> h((size_t)S * es, (char)0xFF), mine((size_t)std::max<int64_t>(n, 1) * es);
    if (n > 0) {
      GXD_HIP(hipMemcpy(mine.data(), keys, (size_t)n * es, hipMemcpyDeviceToHost));
      for (int i = 0; i < S; ++i) std::memcpy(&h[(size_t)i * es], &mine[(size_t)((int64_t)i * n / S) * es], es);
    } else if (dtype == GX_INT32 || dtype == GX_INT64 || dtype == GX_FLOAT32 || dtype == GX_FLOAT64) {
      for (int i = 0; i < S; ++i) h[(size_t)i * es + es - 1] = (char)0x7F;  // INT_MAX / a NaN: sorts last either way
    }
    GXD_HIP(hipMemcpy(samp, h.data(), h.size(), hipMemcpyHostToDevice));
  }
  GXD_GX(c->tp->allgather(samp, allsamp, (size_t)S * es, c->xs));
  GXD_GX(with_tmp(c, Arena::TMP2, [&](void* t, size_t* b) {
    return gx_sort_keys(dtype, allsamp, sorted, (int64_t)S * W, 0, t, b, reinterpret_cast<gx_stream_t>(c->xs));
  }));
  std::vector<char> hs((size_t)S * es * W), split((size_t)MAX_WORLD * 8, 0);
  GXD_HIP(hipMemcpyAsync(hs.data(), sorted, hs.size(), hipMemcpyDeviceToHost, c->xs));
  GXD_HIP(hipStreamSynchronize(c->xs));
  for (int r = 1; r < W; ++r) std::memcpy(&split[(size_t)(r - 1) * es], &hs[(size_t)r * S * es], es);
  tr.mark("splitters");
  // ---- one range-partition pass per chunk, exchange, one local sort
  Exchange ex;
  std::vector<long long> bases;
  int64_t nmax = n;
  GXD_GX(shard_sizes(c, n, bases, &nmax));
  GXD_GX(partition_exchange(c, dtype, keys, n, nmax, 1, split.data(), false, chunks, 0, RowCode{}, stream, &ex, nullptr));
  GXD_HIP(hipStreamWaitEvent(stream, c->evX, 0));
  tr.mark("partition + exchange");
  if (ex.total > 0) {
    void* out = (spare && (size_t)ex.total * es <= spare_bytes) ? spare : alloc((size_t)ex.total * es, actx);
    if (!out) return fail(GX_EINVAL, "gxd_sort: allocator returned NULL");
    tr.mark("result allocation");
    void* rk = c->arena.p[Arena::RECV_KEYS];
    GXD_GX(with_tmp(c, Arena::TMP2, [&](void* t, size_t* b) { return gx_sort_keys(dtype, rk, out, ex.total, 0, t, b, gstream); }));
    *out_keys = out;
  }
  GXD_HIP(hipStreamSynchronize(stream));
  tr.mark("local sort");
  *out_n   = ex.total;
  c->ms[2] = now_ms() - t0;
  return 0;
}

// ------------------------------------------------------------------------------------------------------------ join
int gxd_join_build(gxd_comm* c, int key_dtype, const void* build_keys, int64_t n, int force_exchange, gx_stream_t gstream, gxd_join** out)
{
  if (!c || !out || n < 0 || (n > 0 && !build_keys)) return GX_EINVAL;
  const int ks = elem_size(key_dtype);
  if (ks != 4 && ks != 8) return GX_EDTYPE;
  hipStream_t stream = reinterpret_cast<hipStream_t>(gstream);
  const double t0    = now_ms();
  Trace tr("gxd_join_build");
  auto* j            = new gxd_join;
  j->comm            = c;
  j->key_size        = ks;
  j->single          = c->world == 1 && !force_exchange;
  const void* tkeys  = build_keys;
  const int32_t* tpl = nullptr;  // payload of the table slots (NULL: the row number)
  int64_t tn         = n;
  if (!j->single) {
    std::vector<long long> bases;
    int64_t nmax = n;
    GXD_HIP(hipStreamSynchronize(stream));
    GXD_GX(shard_sizes(c, n, bases, &nmax));
    // Rows travel -- and sit in the table slots -- as (source rank << shift) | row at the source whenever a shard fits the
    // row field (2^28 rows at 8 ranks): the probe's pairs then decode in one streaming pass.  Larger build shards keep the
    // position in the receive buffer in the slot and are translated by a gather (gx_gather_global_rows_dev).
    const int sh = code_shift(c->world);
    RowCode code;
    if (nmax <= (1ll << sh)) code.shift = sh;
    Exchange ex;
    GXD_GX(partition_exchange(c, key_dtype, build_keys, n, nmax, 0, nullptr, true, 4, 0, code, stream, &ex, nullptr));
    GXD_HIP(hipStreamWaitEvent(stream, c->evX, 0));
    GXD_HIP(hipStreamSynchronize(stream));
    tr.mark("partition + exchange");
    // the received rows stay with the table (the arena's receive buffers are reused by every probe)
    j->nrows = ex.total;
    j->rows_bytes = (size_t)std::max<int64_t>(ex.total, 1) * 4;
    j->keys_bytes = (size_t)std::max<int64_t>(ex.total, 1) * ks;
    GXD_HIP((hipError_t)c->pool_get(j->rows_bytes, reinterpret_cast<void**>(&j->rows)));
    GXD_HIP((hipError_t)c->pool_get(j->keys_bytes, &j->keys_keep));
    if (ex.total > 0) {  // (copy kernels on the caller's stream: nothing here touches the null stream)
      GXD_GX(gx_copy_bytes(c->arena.p[Arena::RECV_ROWS], j->rows, (size_t)ex.total * 4, gstream));
      GXD_GX(gx_copy_bytes(c->arena.p[Arena::RECV_KEYS], j->keys_keep, (size_t)ex.total * ks, gstream));
    }
    j->seg_counts = ex.seg_counts;
    j->seg_bases.resize(ex.seg_counts.size());
    for (size_t i = 0; i < ex.seg_counts.size(); ++i) j->seg_bases[i] = bases[i % c->world];
    GXD_HIP(hipMalloc(&j->segtab, (2 * j->seg_counts.size() + 1) * sizeof(long long)));
    GXD_GX(upload_segtab(c, j->seg_counts, j->seg_bases, j->segtab, stream));
    GXD_HIP(hipMalloc(&j->bases_dev, sizeof(long long) * c->world));
    GXD_HIP(hipMemcpy(j->bases_dev, bases.data(), sizeof(long long) * c->world, hipMemcpyHostToDevice));
    j->enc_shift = code.shift;
    tkeys = j->keys_keep;
    tpl   = code.shift ? j->rows : nullptr;
    tn    = ex.total;
  }
  tr.mark("keep rows");
  j->table_bytes = gx_join_table_bytes(ks, tn, 0.5);
  GXD_HIP((hipError_t)c->pool_get(j->table_bytes, &j->table));
  tr.mark("table allocation");
  if (tn >= (1 << 20) && gx_join_partition_bits(ks, j->table_bytes) > 0) {
    GXD_GX(with_tmp(c, Arena::TMP2, [&](void* t, size_t* b) {
      return gx_join_build_partitioned_pl(ks, tkeys, tpl, tn, j->table, j->table_bytes, 0.5, t, b, gstream);
    }));
  } else {
    GXD_GX(gx_join_build_pl(ks, tkeys, tpl, nullptr, tn, j->table, j->table_bytes, 0.5, gstream));
  }
  GXD_HIP(hipStreamSynchronize(stream));
  tr.mark("table build");
  c->ms[2] = now_ms() - t0;
  *out     = j;
  return 0;
}

int gxd_join_destroy(gxd_join* j)
{
  if (!j) return 0;
  (void)hipDeviceSynchronize();
  j->comm->pool_put(j->table, j->table_bytes);
  j->comm->pool_put(j->rows, j->rows_bytes);
  j->comm->pool_put(j->keys_keep, j->keys_bytes);
  if (j->segtab) (void)hipFree(j->segtab);
  if (j->bases_dev) (void)hipFree(j->bases_dev);
  delete j;
  return 0;
}

int gxd_join_probe(gxd_join* j, const void* probe_keys, int64_t n, int chunks, gxd_alloc_fn alloc, void* actx, int64_t** out_probe_rows,
                   int64_t** out_build_rows, int64_t* out_pairs, gx_stream_t gstream)
{
  if (!j || !alloc || !out_probe_rows || !out_build_rows || !out_pairs || n < 0 || (n > 0 && !probe_keys)) return GX_EINVAL;
  gxd_comm* c        = j->comm;
  hipStream_t stream = reinterpret_cast<hipStream_t>(gstream);
  const int ks       = j->key_size;
  const int W        = c->world;
  const double t0    = now_ms();
  *out_probe_rows = *out_build_rows = nullptr;
  *out_pairs                        = 0;
  void* cur;
  GXD_GX(c->arena.get(Arena::CURSOR, 256, &cur));
  const int key_dtype    = ks == 8 ? GX_INT64 : GX_INT32;  // bit patterns are hashed / compared
  const bool partitioned = gx_join_partition_bits(ks, j->table_bytes) > 0;

  auto probe_into = [&](const void* keys, int64_t rows, int32_t row_base, void* pl, void* pr, int64_t cap) -> int {
    if (rows == 0) return 0;
    if (partitioned)
      return with_tmp(c, Arena::TMP2, [&](void* t, size_t* b) {
        return gx_join_probe_partitioned_at(ks, keys, rows, row_base, j->table, j->table_bytes, 0, static_cast<int32_t*>(pl),
                                            static_cast<int32_t*>(pr), cap, static_cast<int64_t*>(cur), t, b, gstream);
      });
    return gx_join_probe(ks, keys, nullptr, rows, j->table, j->table_bytes, 0, static_cast<int32_t*>(pl), static_cast<int32_t*>(pr), cap,
                         static_cast<int64_t*>(cur), gstream);
  };
  auto read_cursor = [&](long long* v) -> int {
    GXD_HIP(hipMemcpyAsync(v, cur, sizeof(long long), hipMemcpyDeviceToHost, stream));
    GXD_HIP(hipStreamSynchronize(stream));
    return 0;
  };

  if (j->single) {  // one rank, no exchange: local rows are global rows
    int64_t cap = std::max<int64_t>(n, 1);
    long long pairs = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
      void *pl, *pr;
      GXD_GX(c->arena.get(Arena::PAIR_L, (size_t)cap * 4, &pl));
      GXD_GX(c->arena.get(Arena::PAIR_R, (size_t)cap * 4, &pr));
      GXD_HIP(hipMemsetAsync(cur, 0, 8, stream));
      GXD_GX(probe_into(probe_keys, n, 0, pl, pr, cap));
      GXD_GX(read_cursor(&pairs));
      if (pairs <= cap) break;
      cap = pairs;  // duplicate build keys blew the guess: the size is now known exactly
    }
    if (pairs > 0) {
      auto* ol = static_cast<int64_t*>(alloc((size_t)pairs * 8, actx));
      auto* orr = static_cast<int64_t*>(alloc((size_t)pairs * 8, actx));
      if (!ol || !orr) return fail(GX_EINVAL, "gxd_join_probe: allocator returned NULL");
      GXD_GX(gx_widen_i32_i64(static_cast<const int32_t*>(c->arena.p[Arena::PAIR_L]), pairs, ol, gstream));
      GXD_GX(gx_widen_i32_i64(static_cast<const int32_t*>(c->arena.p[Arena::PAIR_R]), pairs, orr, gstream));
      *out_probe_rows = ol;
      *out_build_rows = orr;
    }
    GXD_HIP(hipStreamSynchronize(stream));
    *out_pairs = pairs;
    c->ms[2]   = now_ms() - t0;
    return 0;
  }

  std::vector<long long> bases;
  int64_t nmax = n;
  GXD_HIP(hipStreamSynchronize(stream));
  GXD_GX(shard_sizes(c, n, bases, &nmax));
  // Probe rows travel as (source rank << shift) | row inside the sender's CHUNK; chunks are cut so that a chunk fits the row
  // field.  The local partition pass carries that int32 along (payload form), the probe writes it into the pair array, and one
  // streaming pass decodes it with the pair positions at which each received chunk's probe started -- no gather.  Tables too
  // small for the partitioned probe take the plain probe + gather path (small inputs).
  const int sh = code_shift(W);
  RowCode code;
  if (partitioned) {
    code.shift    = sh;
    code.in_chunk = true;
  }
  Exchange ex;
  int64_t pair_cap = 0;
  void *pl = nullptr, *pr = nullptr, *snap = nullptr;
  GXD_GX(c->arena.get(Arena::MISC_C, sizeof(long long) * 1024, &snap));
  GXD_HIP(hipMemsetAsync(cur, 0, 8, stream));
  // a chunk is probed as soon as it has been posted: its probe overlaps the exchange of the later chunks
  auto on_chunk = [&](int k, int64_t first, int64_t rows) -> int {
    if (!partitioned) return 0;  // small tables: one direct probe of everything at the end
    const int64_t need = first + rows;
    if (need > pair_cap) {  // pairs <= probe rows unless build keys repeat (checked at the end)
      const int64_t want = std::max<int64_t>(need + need / 4, n + n / 4 + 65536);
      GXD_GX(c->arena.grow(Arena::PAIR_L, (size_t)want * 4, (size_t)pair_cap * 4, &pl));
      GXD_GX(c->arena.grow(Arena::PAIR_R, (size_t)want * 4, (size_t)pair_cap * 4, &pr));
      pair_cap = want;
    }
    GXD_HIP(hipStreamWaitEvent(stream, c->evX, 0));
    GXD_HIP(hipMemcpyAsync(static_cast<long long*>(snap) + k, cur, 8, hipMemcpyDeviceToDevice, stream));  // pairs before this chunk
    if (rows == 0) return 0;
    const char* rk    = static_cast<const char*>(c->arena.p[Arena::RECV_KEYS]);
    const int32_t* rr = static_cast<const int32_t*>(c->arena.p[Arena::RECV_ROWS]);
    return with_tmp(c, Arena::TMP2, [&](void* t, size_t* b) {
      return gx_join_probe_partitioned_pl(ks, rk + first * ks, rr + first, rows, 0, j->table, j->table_bytes, 0, static_cast<int32_t*>(pl),
                                          static_cast<int32_t*>(pr), pair_cap, static_cast<int64_t*>(cur), t, b, gstream);
    });
  };
  GXD_GX(partition_exchange(c, key_dtype, probe_keys, n, nmax, 0, nullptr, true, chunks, partitioned ? (1ll << sh) : 0, code, stream, &ex, on_chunk,
                            1023 /* snap[] holds 1024 pair positions */));
  GXD_HIP(hipStreamWaitEvent(stream, c->evX, 0));
  long long pairs = 0;
  if (!partitioned) {
    pair_cap = std::max<int64_t>(ex.total, 1);
    GXD_GX(c->arena.get(Arena::PAIR_L, (size_t)pair_cap * 4, &pl));
    GXD_GX(c->arena.get(Arena::PAIR_R, (size_t)pair_cap * 4, &pr));
    GXD_GX(probe_into(c->arena.p[Arena::RECV_KEYS], ex.total, 0, pl, pr, pair_cap));
  }
  GXD_GX(read_cursor(&pairs));
  if (pairs > pair_cap) {  // duplicate build keys: probe again into buffers of the exact size
    pair_cap = pairs;
    GXD_GX(c->arena.get(Arena::PAIR_L, (size_t)pair_cap * 4, &pl));
    GXD_GX(c->arena.get(Arena::PAIR_R, (size_t)pair_cap * 4, &pr));
    GXD_HIP(hipMemsetAsync(cur, 0, 8, stream));
    const char* rk    = static_cast<const char*>(c->arena.p[Arena::RECV_KEYS]);
    const int32_t* rr = static_cast<const int32_t*>(c->arena.p[Arena::RECV_ROWS]);
    if (partitioned) {
      for (int k = 0; k < ex.chunks; ++k) {
        const int64_t first = ex.chunk_start[k], rows = ex.chunk_start[k + 1] - ex.chunk_start[k];
        GXD_HIP(hipMemcpyAsync(static_cast<long long*>(snap) + k, cur, 8, hipMemcpyDeviceToDevice, stream));
        if (rows == 0) continue;
        GXD_GX(with_tmp(c, Arena::TMP2, [&](void* t, size_t* b) {
          return gx_join_probe_partitioned_pl(ks, rk + first * ks, rr + first, rows, 0, j->table, j->table_bytes, 0, static_cast<int32_t*>(pl),
                                              static_cast<int32_t*>(pr), pair_cap, static_cast<int64_t*>(cur), t, b, gstream);
        }));
      }
    } else {
      GXD_GX(probe_into(rk, ex.total, 0, pl, pr, pair_cap));
    }
    GXD_GX(read_cursor(&pairs));
  }
  if (pairs > 0) {
    auto* ol  = static_cast<int64_t*>(alloc((size_t)pairs * 8, actx));
    auto* orr = static_cast<int64_t*>(alloc((size_t)pairs * 8, actx));
    if (!ol || !orr) return fail(GX_EINVAL, "gxd_join_probe: allocator returned NULL");
    if (partitioned) {
      // probe side: (rank << sh | row in the sender's chunk) + chunk * crows + the sender's first global row
      GXD_HIP(hipMemcpyAsync(static_cast<long long*>(snap) + ex.chunks, cur, 8, hipMemcpyDeviceToDevice, stream));
      void *bd, *cr;
      GXD_GX(c->arena.get(Arena::MISC_A, sizeof(long long) * W, &bd));
      GXD_GX(c->arena.get(Arena::MISC_B, sizeof(long long) * W, &cr));
      std::vector<long long> crv(W, ex.crows);
      GXD_HIP(hipMemcpyAsync(bd, bases.data(), sizeof(long long) * W, hipMemcpyHostToDevice, stream));
      GXD_HIP(hipMemcpyAsync(cr, crv.data(), sizeof(long long) * W, hipMemcpyHostToDevice, stream));
      GXD_HIP(hipStreamSynchronize(stream));  // host vectors
      GXD_GX(gx_decode_global_rows(static_cast<const int32_t*>(pl), pairs, sh, static_cast<const int64_t*>(bd), static_cast<const int64_t*>(cr),
                                   static_cast<const int64_t*>(snap), ex.chunks, ol, gstream));
    } else {
      // (position in the receive buffer) -> (int32 local row at its source) + (first global row of that source's shard)
      std::vector<long long> segb(ex.seg_counts.size());
      for (size_t i = 0; i < segb.size(); ++i) segb[i] = bases[i % W];
      void* st;
      GXD_GX(c->arena.get(Arena::SEGTAB, (2 * segb.size() + 1) * sizeof(long long), &st));
      GXD_GX(upload_segtab(c, ex.seg_counts, segb, st, stream));
      GXD_GX(gx_gather_global_rows_dev(static_cast<const int32_t*>(c->arena.p[Arena::RECV_ROWS]), ex.total, static_cast<const int32_t*>(pl), pairs,
                                       (int)segb.size(), static_cast<const int64_t*>(st), ol, gstream));
    }
    if (j->enc_shift)  // build side: the slots hold (rank << shift) | row at the source
      GXD_GX(gx_decode_global_rows(static_cast<const int32_t*>(pr), pairs, j->enc_shift, static_cast<const int64_t*>(j->bases_dev), nullptr, nullptr, 0,
                                   orr, gstream));
    else
      GXD_GX(gx_gather_global_rows_dev(j->rows, j->nrows, static_cast<const int32_t*>(pr), pairs, (int)j->seg_counts.size(),
                                       static_cast<const int64_t*>(j->segtab), orr, gstream));
    *out_probe_rows = ol;
    *out_build_rows = orr;
  }
  GXD_HIP(hipStreamSynchronize(stream));
  *out_pairs = pairs;
  c->ms[2]   = now_ms() - t0;
  return 0;
}

// ------------------------------------------------------------------------------------------------------------ groupby
int gxd_groupby_sum_count(gxd_comm* c, int key_dtype, const void* keys, int val_dtype, const void* vals, int64_t n, int64_t max_groups,
                          int force_exchange, gxd_alloc_fn alloc, void* actx, void** out_keys, void** out_sums, int64_t** out_counts,
                          int64_t* out_groups, gx_stream_t gstream)
{
  if (!c || !alloc || !out_keys || !out_sums || !out_counts || !out_groups || n < 0 || (n > 0 && (!keys || !vals))) return GX_EINVAL;
  const int ks = elem_size(key_dtype);
  if ((key_dtype != GX_INT32 && key_dtype != GX_INT64)) return GX_EDTYPE;
  const bool fsum    = val_dtype == GX_FLOAT32 || val_dtype == GX_FLOAT64;
  const int sum_type = fsum ? GX_FLOAT64 : GX_INT64;
  if (!fsum && val_dtype != GX_INT32 && val_dtype != GX_INT64) return GX_EDTYPE;
  hipStream_t stream = reinterpret_cast<hipStream_t>(gstream);
  const double t0    = now_ms();
  const int W        = c->world;
  *out_keys = *out_sums = nullptr;
  *out_counts           = nullptr;
  *out_groups           = 0;
  if (max_groups <= 0) max_groups = std::max<int64_t>(std::min<int64_t>(n, 1 << 20), 1);
  // ---- local aggregate: at most #groups partial rows leave this rank
  void *pk, *ps, *pc, *ng;
  GXD_GX(c->arena.get(Arena::MISC_A, (size_t)max_groups * ks, &pk));
  GXD_GX(c->arena.get(Arena::MISC_B, (size_t)max_groups * 8, &ps));
  GXD_GX(c->arena.get(Arena::MISC_C, (size_t)max_groups * 4, &pc));
  GXD_GX(c->arena.get(Arena::CURSOR, 256, &ng));
  long long g = 0;
  for (;;) {
    GXD_HIP(hipMemsetAsync(ng, 0, 8, stream));
    if (n > 0)
      GXD_GX(with_tmp(c, Arena::TMP2, [&](void* t, size_t* b) {
        return gx_groupby_sum_count(key_dtype, keys, nullptr, val_dtype, vals, nullptr, n, max_groups, pk, ps, static_cast<int32_t*>(pc), nullptr,
                                    static_cast<int64_t*>(ng), t, b, gstream);
      }));
    GXD_HIP(hipMemcpyAsync(&g, ng, 8, hipMemcpyDeviceToHost, stream));
    GXD_HIP(hipStreamSynchronize(stream));
    if (g >= 0 && g <= max_groups) break;
    if (max_groups >= n) return fail(GX_EOVERFLOW, "gxd_groupby_sum_count: group table overflow");
    max_groups = std::min<int64_t>(n, max_groups * 8);  // more groups than the bound: the table reports overflow, not a count
    GXD_GX(c->arena.get(Arena::MISC_A, (size_t)max_groups * ks, &pk));
    GXD_GX(c->arena.get(Arena::MISC_B, (size_t)max_groups * 8, &ps));
    GXD_GX(c->arena.get(Arena::MISC_C, (size_t)max_groups * 4, &pc));
  }
  // counts as int64 partials (they are summed again after the exchange)
  void* pc64;
  GXD_GX(c->arena.get(Arena::MISC_D, (size_t)std::max<long long>(g, 1) * 8, &pc64));
  if (g > 0) GXD_GX(gx_widen_i32_i64(static_cast<const int32_t*>(pc), g, static_cast<int64_t*>(pc64), gstream));
  const void *mk = pk, *msum = ps, *mcnt = pc64;
  int64_t mrows = g;
  void *rs = nullptr, *rc = nullptr;
  if (!(W == 1 && !force_exchange)) {
    // ---- hash-partition the partial rows, exchange keys + (sum, count) gathered into the same order
    GXD_HIP(hipStreamSynchronize(stream));
    Exchange ex;
    std::vector<long long> gb;
    int64_t gmax = g;
    GXD_GX(shard_sizes(c, g, gb, &gmax));
    GXD_GX(partition_exchange(c, key_dtype, pk, g, std::max<int64_t>(gmax, 1 << 20) /* one chunk */, 0, nullptr, true, 1, 0, RowCode{}, stream, &ex,
                              nullptr));
    // the (sum, count) payload rides the SAME split: per destination, gather by the partition's row map into a contiguous
    // staging area, then one more grouped exchange with the counts of the key exchange
    void *gs, *gc;
    GXD_GX(c->arena.get(Arena::MISC_E, (size_t)std::max<long long>(g, 1) * 8, &gs));
    GXD_GX(c->arena.get(Arena::MISC_F, (size_t)std::max<long long>(g, 1) * 8, &gc));
    std::vector<long long> stage_off(W, 0);
    {
      long long run = 0;
      for (int r = 0; r < W; ++r) {
        stage_off[r]        = run;
        const long long cnt = ex.send_cnt[r];
        if (cnt > 0) {
          const int32_t* map = ex.send_rows[0] + ex.send_off[r];
          GXD_GX(gx_gather(8, ps, nullptr, g, map, cnt, 0, static_cast<char*>(gs) + run * 8, nullptr, gstream));
          GXD_GX(gx_gather(8, pc64, nullptr, g, map, cnt, 0, static_cast<char*>(gc) + run * 8, nullptr, gstream));
        }
        run += cnt;
      }
    }
    GXD_HIP(hipEventRecord(c->evP[0], stream));
    GXD_HIP(hipStreamWaitEvent(c->xs, c->evP[0], 0));
    GXD_GX(c->arena.get(Arena::PAIR_L, (size_t)std::max<int64_t>(ex.total, 1) * 8, &rs));
    GXD_GX(c->arena.get(Arena::PAIR_R, (size_t)std::max<int64_t>(ex.total, 1) * 8, &rc));
    int64_t rpos = 0;
    GXD_GX(c->tp->group_start());
    for (int r = 0; r < W; ++r) {
      const int64_t so = stage_off[r], sc = ex.send_cnt[r];
      const int64_t rcv = ex.seg_counts[r];
      if (r == c->rank) {
        if (sc) {
          GXD_GX(gx_copy_bytes(static_cast<const char*>(gs) + so * 8, static_cast<char*>(rs) + rpos * 8, (size_t)sc * 8, reinterpret_cast<gx_stream_t>(c->xs)));
          GXD_GX(gx_copy_bytes(static_cast<const char*>(gc) + so * 8, static_cast<char*>(rc) + rpos * 8, (size_t)sc * 8, reinterpret_cast<gx_stream_t>(c->xs)));
        }
      } else {
        if (sc) {
          GXD_GX(c->tp->send(static_cast<const char*>(gs) + so * 8, (size_t)sc * 8, r, c->xs));
          GXD_GX(c->tp->send(static_cast<const char*>(gc) + so * 8, (size_t)sc * 8, r, c->xs));
        }
        if (rcv) {
          GXD_GX(c->tp->recv(static_cast<char*>(rs) + rpos * 8, (size_t)rcv * 8, r, c->xs));
          GXD_GX(c->tp->recv(static_cast<char*>(rc) + rpos * 8, (size_t)rcv * 8, r, c->xs));
        }
      }
      rpos += rcv;
    }
    GXD_GX(c->tp->group_end(c->xs));
    GXD_HIP(hipEventRecord(c->evX, c->xs));
    GXD_HIP(hipStreamWaitEvent(stream, c->evX, 0));
    mk    = c->arena.p[Arena::RECV_KEYS];
    msum  = rs;
    mcnt  = rc;
    mrows = ex.total;
  } else if (g == 0) {
    c->ms[2] = now_ms() - t0;
    return 0;
  }
  if (mrows == 0) {
    GXD_HIP(hipStreamSynchronize(stream));
    c->ms[2] = now_ms() - t0;
    return 0;
  }
  // ---- merge: ONE grouping of the received partials (sorted order, run heads, labels) carries the sum and the count
  void *order, *sk, *ss, *sc2, *heads, *labels, *offsets;
  GXD_GX(c->arena.get(Arena::PART_KEYS, (size_t)mrows * 4, &order));
  GXD_GX(with_tmp(c, Arena::TMP2, [&](void* t, size_t* b) {
    return gx_sorted_order(key_dtype, mk, nullptr, mrows, 0, 0, 1, static_cast<int32_t*>(order), t, b, gstream);
  }));
  GXD_GX(c->arena.get(Arena::SAMPLE, (size_t)mrows * ks, &sk));
  GXD_GX(c->arena.get(Arena::ALLSAMPLE, (size_t)mrows * 8, &ss));
  GXD_GX(c->arena.get(Arena::OFFS, (size_t)mrows * 8, &sc2));
  GXD_GX(gx_gather(ks, mk, nullptr, mrows, static_cast<const int32_t*>(order), mrows, 0, sk, nullptr, gstream));
  GXD_GX(gx_gather(8, msum, nullptr, mrows, static_cast<const int32_t*>(order), mrows, 0, ss, nullptr, gstream));
  GXD_GX(gx_gather(8, mcnt, nullptr, mrows, static_cast<const int32_t*>(order), mrows, 0, sc2, nullptr, gstream));
  GXD_GX(c->arena.get(Arena::ALLOFFS, (size_t)mrows, &heads));
  GXD_GX(c->arena.get(Arena::PART_ROWS, (size_t)mrows * 4, &labels));
  GXD_GX(c->arena.get(Arena::SEGTAB, ((size_t)mrows + 1) * 4, &offsets));
  GXD_GX(gx_group_heads(key_dtype, sk, nullptr, nullptr, mrows, 0, static_cast<uint8_t*>(heads), gstream));
  GXD_HIP(hipMemsetAsync(ng, 0, 8, stream));
  GXD_GX(with_tmp(c, Arena::TMP2, [&](void* t, size_t* b) {
    return gx_group_offsets(static_cast<const uint8_t*>(heads), mrows, static_cast<int32_t*>(labels), static_cast<int32_t*>(offsets), nullptr,
                            static_cast<int64_t*>(ng), t, b, gstream);
  }));
  long long G = 0;
  GXD_HIP(hipMemcpyAsync(&G, ng, 8, hipMemcpyDeviceToHost, stream));
  GXD_HIP(hipStreamSynchronize(stream));
  void* ok = alloc((size_t)G * ks, actx);
  void* os = alloc((size_t)G * 8, actx);
  auto* oc = static_cast<int64_t*>(alloc((size_t)G * 8, actx));
  if (!ok || !os || !oc) return fail(GX_EINVAL, "gxd_groupby_sum_count: allocator returned NULL");
  GXD_GX(with_tmp(c, Arena::TMP2, [&](void* t, size_t* b) {
    return gx_segmented_reduce(sum_type, ss, nullptr, static_cast<const uint8_t*>(heads), static_cast<const int32_t*>(labels), mrows, GX_OP_SUM,
                               os, nullptr, t, b, gstream);
  }));
  GXD_GX(with_tmp(c, Arena::TMP2, [&](void* t, size_t* b) {
    return gx_segmented_reduce(GX_INT64, sc2, nullptr, static_cast<const uint8_t*>(heads), static_cast<const int32_t*>(labels), mrows, GX_OP_SUM,
                               oc, nullptr, t, b, gstream);
  }));
  GXD_GX(gx_gather(ks, sk, nullptr, mrows, static_cast<const int32_t*>(offsets), G, 0, ok, nullptr, gstream));
  GXD_HIP(hipStreamSynchronize(stream));
  *out_keys   = ok;
  *out_sums   = os;
  *out_counts = oc;
  *out_groups = G;
  c->ms[2]    = now_ms() - t0;
  return 0;
}

}  // extern "C"
