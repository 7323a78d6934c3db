"""gx_sorted_order of n random int32 keys: the packed-word path ((sortable key << 32) | row through the 64-bit keys-only cursor sort)
against the stable LSD pair passes (gx_sort_set_cursor_path(0)); whole-call times from HIP events; the order is checked by
gathering the keys (sortedness on the device) and, between the two paths, by checksums of the permutation."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import cudf_amd
from cudf_amd import Column, ops, _lib as L

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
arg = sys.argv[2] if len(sys.argv) > 2 else ""
sparse = arg.startswith("sparse")  # "sparse1e6": 1e6 distinct keys spread over the whole 32-bit range (x 2^12 + a constant)
lo, hi = (0, int(float(arg.replace("sparse", "")))) if arg else (0, 0)  # optional key range [0, hi): ties
sp = ops.stream_ptr()
keys = ops.random_column(np.int32, n, seed=7, lo=lo, hi=hi)
if sparse:
    t = keys.data[: n * 4].view(torch.int32)
    t.mul_(4093).add_(-2000000000)  # ids 0 .. hi-1 -> hi distinct values 4093 apart: sparse in a 32-bit range
out = Column.empty(np.int32, n)
sums = {}
for cursor in (0, 1, 0, 1):
    L.lib.gx_sort_set_cursor_path(cursor, 0.0)
    nb = ctypes.c_size_t(0)
    L.check(L.lib.gx_sorted_order(keys.gx, keys.data_ptr, None, n, 0, 0, 1, out.data_ptr, None, ctypes.byref(nb), sp), "query")
    tmp = ops.device_bytes(nb.value)
    call = lambda: L.check(L.lib.gx_sorted_order(keys.gx, keys.data_ptr, None, n, 0, 0, 1, out.data_ptr, ops.ptr(tmp), ctypes.byref(nb), sp), "order")
    call(); call()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(5): call()
    e.record(); torch.cuda.synchronize()
    total = s.elapsed_time(e) / 5
    srt = ops.gather(keys, out)
    assert ops.checksum(srt)[2] == 0
    sums[cursor] = ops.checksum(out)[:2]
    print(f"sorted_order int32 n={n:.1e} keys in [{lo}, {hi}){' x 4093 (sparse)' if sparse else ''} cursor_path={cursor} total {total:7.3f} ms = {n / total / 1e6:6.1f} G rows/s  scratch {nb.value / 1e9:.1f} GB", flush=True)
    del tmp, srt
assert sums[0] == sums[1], sums
L.lib.gx_sort_set_cursor_path(1, 0.0)
print("ok")
