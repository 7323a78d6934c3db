"""Measurement (not product code): per-rank cost of the sharded operators' steps on ONE MI355X -- a 1-rank communicator with the
exchange path forced (a device-local copy stands in for the links).  Usage: xp_gxd_steps.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import cudf_amd  # noqa: F401
from cudf_amd import ops, gxd

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000


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


comm = gxd.Communicator()
keys = as_t(ops.random_column(np.int64, n, seed=1), torch.int64)
print(f"rows {n:.1e}")
single = timed(lambda: comm.sort(keys))
print(f"gx_sort_keys (one GPU, no exchange)                        {single:8.2f} ms")
for mode, name in ((0, "fused: level 0 | exchange | level 1 + cells"), (1, "sample sort: range partition | exchange | sort")):
    gxd.set_sort_mode(mode)
    ms = timed(lambda: comm.sort(keys, chunks=4, force_exchange=True))
    t = comm.last_timing()
    print(f"gxd_sort forced, {name:48s} {ms:8.2f} ms   (fused path taken: {t[0] == -1.0})")
gxd.set_sort_mode(0)
os.environ["GXD_TRACE"] = "1"
comm.sort(keys, force_exchange=True)
del os.environ["GXD_TRACE"]
del keys
nb = n // 10
bk = torch.randperm(nb, device="cuda") * 3 + 1
pk = as_t(ops.random_column(np.int64, n, seed=2, lo=0, hi=int(nb / 0.3)), torch.int64) * 3 + 1
for label in ("first call (allocations)", "second call (pooled)    ", "third call (pooled)     "):
    t0 = time.perf_counter()
    gj = gxd.HashJoin(comm, bk, force_exchange=True)
    torch.cuda.synchronize()
    print(f"gxd_join_build forced ({nb:.0e}), {label}          {(time.perf_counter() - t0) * 1e3:8.2f} ms")
    if not label.startswith("third"):
        gj.close()
for ch in (1, 4):
    print(f"gxd_join_probe forced, {ch:2d} chunks                               {timed(lambda: gj.inner_join(pk, chunks=ch)):8.2f} ms")
gj.close()
del bk, pk
gk = as_t(ops.random_column(np.int32, n, seed=3, lo=0, hi=1_000_000), torch.int32)
gv = as_t(ops.random_column(np.float64, n, seed=4), torch.float64)
print(f"gxd_groupby_sum_count forced                                {timed(lambda: comm.groupby_sum_count(gk, gv, max_groups=1 << 20, force_exchange=True)):8.2f} ms")
comm.close()
