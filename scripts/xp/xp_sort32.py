"""gx_sort_keys of n random int32 keys: the cursor path (two atomic-cursor partition levels + k_local_place on 32-bit words) against
the LSD passes (gx_sort_set_cursor_path(0)); whole-call times from HIP events, outputs checked (device checksums)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import cudf_amd
from cudf_amd import Column, ops, _lib as L

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
sp = ops.stream_ptr()
keys = ops.random_column(np.int32, n, seed=7)
out = Column.empty(np.int32, n)
ref = ops.checksum(keys)
for cursor in (0, 1, 0, 1):
    L.lib.gx_sort_set_cursor_path(cursor, 0.0)
    nb = ctypes.c_size_t(0)
    L.check(L.lib.gx_sort_keys(keys.gx, keys.data_ptr, out.data_ptr, n, 0, None, ctypes.byref(nb), sp), "query")
    tmp = ops.device_bytes(nb.value)
    call = lambda: L.check(L.lib.gx_sort_keys(keys.gx, keys.data_ptr, out.data_ptr, n, 0, ops.ptr(tmp), ctypes.byref(nb), sp), "sort")
    call(); call()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(5): call()
    e.record(); torch.cuda.synchronize()
    total = s.elapsed_time(e) / 5
    L.lib.gx_sort_profile(1)
    call()
    h4 = (ctypes.c_float * 4)()
    ok = L.lib.gx_sort_profile_read_hybrid(h4) == 0
    L.lib.gx_sort_profile(0)
    st, todo = ctypes.c_int32(-1), ctypes.c_int32(-1)
    L.lib.gx_sort_cursor_state(ops.ptr(tmp), ctypes.byref(st), sp)
    L.lib.gx_sort_place_info(ops.ptr(tmp), ctypes.byref(todo), sp)
    info = (ctypes.c_int32 * 8)()
    L.lib.gx_sort_info(ops.ptr(tmp), info, sp)
    cs = ops.checksum(out)
    assert cs[2] == 0 and cs[:2] == ref[:2], (cs, ref)
    stages = f"level0 {h4[0]:5.2f} level1 {h4[1]:5.2f} local stage {h4[3]:5.2f} ms" if (ok and cursor) else ""
    print(f"sort_keys int32 n={n:.1e} cursor_path={cursor} total {total:7.3f} ms = {n / total / 1e6:6.1f} G rows/s | state {st.value} hybrid_used {info[1]} "
          f"lsd_passes {info[7]} crowded {todo.value} | {stages}  scratch {nb.value / 1e9:.1f} GB", flush=True)
    del tmp
L.lib.gx_sort_set_cursor_path(1, 0.0)
print("ok")
