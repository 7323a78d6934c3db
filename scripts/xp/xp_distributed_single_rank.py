"""Measurement (not product code): the per-rank cost of the sharded operators' local steps on ONE MI355X, with the
partition + all-to-all path forced on a 1-rank RCCL group (the all-to-all is then a device-local copy).  Gives the
partition-pass costs the multi-GPU scaling model of DESIGN.md section 6 uses.  Usage: xp_distributed_single_rank.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
import cudf_amd
from cudf_amd import ops, distributed as D

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
D._FORCE_EXCHANGE = True
local = D.HipLocalOps()


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def as_t(col, dt):
    return col.data[: col.size * col.dtype.itemsize].view(dt)


keys = as_t(ops.random_column(np.int64, n, seed=1), torch.int64)
print(f"rows {n:.1e}")
print(f"local sort                                  {timed(lambda: local.sort(keys)):8.2f} ms")
sp = torch.sort(keys[:: max(1, n // 4096)])[0]
splitters = sp[[len(sp) // 8 * i for i in range(1, 8)]].cpu().tolist()
print(f"range partition into 8 (gx_partition_rows)  {timed(lambda: local.range_partition(keys, splitters)):8.2f} ms")
print(f"hash partition into 8 of (key,row)          {timed(lambda: local.hash_partition_rows(keys, 8)):8.2f} ms")
print(f"distributed_sort (round-2 python path)      {timed(lambda: D.distributed_sort(keys, local=local)):8.2f} ms")
torch.cuda.empty_cache()
comm = D._gxd_comm(None)
for ch in (1, 4, 8, 16):
    print(f"gxd_sort forced, {ch:2d} chunks                   {timed(lambda: comm.sort(keys, chunks=ch, force_exchange=True)):8.2f} ms   (enqueue, count waits, total) = "
          + ", ".join(f"{x:.2f}" for x in comm.last_timing()))
del keys
nb = n // 10
bk = torch.randperm(nb, device="cuda") * 3 + 1
pk = as_t(ops.random_column(np.int64, n, seed=2, lo=0, hi=int(nb / 0.3)), torch.int64) * 3 + 1
t0 = time.perf_counter()
hj = D.DistributedHashJoin(bk, local=local)
torch.cuda.synchronize()
print(f"DistributedHashJoin build (round-2 python)  {(time.perf_counter() - t0) * 1e3:8.2f} ms")
print(f"DistributedHashJoin.inner_join (r2 python)  {timed(lambda: hj.inner_join(pk)):8.2f} ms")
del hj
torch.cuda.empty_cache()
from cudf_amd import gxd
# the FIRST build of a shape pays for its buffers: the communicator's arena grows (hipDeviceSynchronize + hipFree + hipMalloc per
# slot: the partition / receive buffers sized for the 1e9-row sort above are released and re-made) and the 4.3 GB table, the kept
# rows and keys come from hipMalloc -- the 1269.6 ms of profiles/r3_run36_single_rank_steps.txt was this one-off (a single
# multi-gigabyte hipMalloc was measured at up to 1.87 s on these boxes: profiles/r3_run14_alloc_probe.txt).  Destroyed tables go
# to the communicator's pool, so every later build of the shape allocates nothing: that is the number a pipeline sees.
for label in ("first call (allocations)", "second call (pooled)    ", "third call (pooled)     "):
    t0 = time.perf_counter()
    gj = gxd.HashJoin(comm, bk, force_exchange=True)
    torch.cuda.synchronize()
    print(f"gxd_join_build forced (1e8), {label} {(time.perf_counter() - t0) * 1e3:8.2f} ms")
    if not label.startswith("third"):
        gj.close()
for ch in (1, 4, 8, 16):
    print(f"gxd_join_probe forced, {ch:2d} chunks             {timed(lambda: gj.inner_join(pk, chunks=ch)):8.2f} ms   (enqueue, count waits, total) = "
          + ", ".join(f"{x:.2f}" for x in comm.last_timing()))
gj.close()
del bk, pk
gk = as_t(ops.random_column(np.int32, n, seed=3, lo=0, hi=1_000_000), torch.int32)
gv = as_t(ops.random_column(np.float64, n, seed=4), torch.float64)
print(f"distributed_groupby_sum_count (r2 python)   {timed(lambda: D.distributed_groupby_sum_count(gk, gv, local=local)):8.2f} ms")
print(f"gxd_groupby_sum_count forced                {timed(lambda: comm.groupby_sum_count(gk, gv, max_groups=1 << 20, force_exchange=True)):8.2f} ms")
D.close_communicators()
dist.destroy_process_group()
