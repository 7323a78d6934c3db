"""gx_sort_keys of 1e9 keys drawn from a range that is not a power of two -- int64 in [0, 1e12), int32 in [0, 1.5e9) -- with the cursor
path (whose verdict kernel takes one more level-1 bit when the exact level-0 histogram shows fuller buckets) and without it
(gx_sort_set_cursor_path(0): the look-back path with the level-1 bits n alone suggests; its cells overflow -> LSD passes)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import cudf_amd
from cudf_amd import Column, ops, _lib as L

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
sp = ops.stream_ptr()
for dt, hi in ((np.int64, 10**12), (np.int32, 1_500_000_000)):
    keys = ops.random_column(dt, n, seed=11, lo=0, hi=hi)
    out = Column.empty(dt, n)
    ref = ops.checksum(keys)
    for cursor in (0, 1):
        L.lib.gx_sort_set_cursor_path(cursor, 0.0)
        nb = ctypes.c_size_t(0)
        L.check(L.lib.gx_sort_keys(keys.gx, keys.data_ptr, out.data_ptr, n, 0, None, ctypes.byref(nb), sp), "query")
        tmp = ops.device_bytes(nb.value)
        call = lambda: L.check(L.lib.gx_sort_keys(keys.gx, keys.data_ptr, out.data_ptr, n, 0, ops.ptr(tmp), ctypes.byref(nb), sp), "sort")
        call(); call()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(4): call()
        e.record(); torch.cuda.synchronize()
        total = s.elapsed_time(e) / 4
        st, todo = ctypes.c_int32(-1), ctypes.c_int32(-1)
        L.lib.gx_sort_cursor_state(ops.ptr(tmp), ctypes.byref(st), sp)
        L.lib.gx_sort_place_info(ops.ptr(tmp), ctypes.byref(todo), sp)
        info = (ctypes.c_int32 * 8)()
        L.lib.gx_sort_info(ops.ptr(tmp), info, sp)
        cs = ops.checksum(out)
        assert cs[2] == 0 and cs[:2] == ref[:2], (cs, ref)
        print(f"sort_keys {np.dtype(dt).name} n={n:.1e} keys in [0, {hi:.2e}) cursor_path={cursor} total {total:7.3f} ms | state {st.value} hybrid_used {info[1]} "
              f"level-1 bits {info[4]} largest cell {info[6]} lsd_passes {info[7]} crowded {todo.value}  scratch {nb.value / 1e9:.1f} GB", flush=True)
        del tmp
    del keys, out
L.lib.gx_sort_set_cursor_path(1, 0.0)
print("ok")
