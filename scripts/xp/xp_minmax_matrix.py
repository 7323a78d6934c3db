"""Micro-benchmark (not product code): which ingredient makes k_part_minmax<uint64 key, int32 value> 10x slower than
<uint32 key, float64 value>?  Times gx_groupby_min_max over key width x value type x value pattern x nsplit."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import cudf_amd
from cudf_amd import Column, ops, _lib as L

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 250_000_000
G = 1_000_000


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


k32 = ops.random_column(np.int32, n, seed=1, lo=0, hi=G)
k64 = ops.random_column(np.int64, n, seed=1, lo=0, hi=G)
kh = ops.hash_rows64([k64])                      # uniform 64-bit keys, G distinct values
vf = ops.random_column(np.float64, n, seed=2)
vi_rand = ops.random_column(np.int32, n, seed=3, lo=0, hi=1 << 30)
vi_iota = Column.empty(np.int32, n)
L.check(L.lib.gx_sequence_i32(vi_iota.data_ptr, n, 0, None), "seq")
v64 = ops.random_column(np.int64, n, seed=4)
for nsplit in (1, 2, 4):
    L.lib.gx_groupby_set_algorithm(0, nsplit)
    for kname, k in (("int32", k32), ("int64", k64), ("hash64", kh)):
        for vname, v in (("f64", vf), ("i32 random", vi_rand), ("i32 iota", vi_iota), ("i64 random", v64)):
            ms = timed(lambda: ops.groupby_min_max(k, v, max_groups_hint=1 << 20))
            print(f"nsplit {nsplit}  key {kname:7s} value {vname:11s} {ms:8.2f} ms  ({n:.1e} rows)", flush=True)
    ms = timed(lambda: ops.groupby_sum_count(k64, vf, max_groups_hint=1 << 20))
    print(f"nsplit {nsplit}  sum_count key int64 value f64 {ms:8.2f} ms", flush=True)
    ms = timed(lambda: ops.groupby_sum_count(kh, vf, max_groups_hint=1 << 20))
    print(f"nsplit {nsplit}  sum_count key hash64 value f64 {ms:8.2f} ms", flush=True)
