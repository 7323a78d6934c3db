"""A/B of the hybrid sort's local stage at size: k_local_place (counting placement + per-thread window networks; default) against
k_local_sort's sub-bucket path for every cell (gx_sort_set_experiment(32)).  gx_sort_keys and gx_sorted_order of n random int64
keys; whole-call times from HIP events on the launch stream, the stages from gx_sort_profile_read_hybrid (marks 3 -> 4 =
k_local_place + k_local_sort).  The two variants' outputs must agree (device checksums; keys: no sortedness violation)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import cudf_amd
from cudf_amd import Column, ops, _lib as L

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
which = sys.argv[2] if len(sys.argv) > 2 else "both"
kind = sys.argv[3] if len(sys.argv) > 3 else "i64"  # f64: random BIT PATTERNS as float64 keys (16384-key cells, packed words)
sp = ops.stream_ptr()
PLACE_GRID = int(os.environ.get("PLACE_GRID", "0"))  # workgroups of k_local_place (0 = one per cell)
L.lib.gx_sort_set_place_grid(PLACE_GRID)
if kind == "f64":
    keys = Column.empty(np.float64, n)
    L.check(L.lib.gx_fill_random(Column.empty(np.int64, 0).gx, keys.data_ptr, n, 42, 0, 0, sp), "fill")
else:
    keys = ops.random_column(np.int64, n, seed=42)
KDT = np.float64 if kind == "f64" else np.int64


def measure(name, call, tmp, out):
    sums = {}
    for bits in (32, 0, 32, 0):
        L.lib.gx_sort_set_experiment(bits)
        L.lib.gx_sort_profile(0)
        call(); call()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(5): call()
        e.record(); torch.cuda.synchronize()
        total = s.elapsed_time(e) / 5
        L.lib.gx_sort_profile(1)
        st = []
        for _ in range(4):
            call()
            h4 = (ctypes.c_float * 4)()
            if L.lib.gx_sort_profile_read_hybrid(h4) == 0: st.append(list(h4))
        L.lib.gx_sort_profile(0)
        todo = ctypes.c_int32(-1)
        L.lib.gx_sort_place_info(ops.ptr(tmp), ctypes.byref(todo), sp)
        state = ctypes.c_int32(-1)
        L.lib.gx_sort_cursor_state(ops.ptr(tmp), ctypes.byref(state), sp)
        sums[bits] = ops.checksum(out)
        m = np.mean(np.array(st), axis=0) if st else [float("nan")] * 4
        print(f"{name:16s} place_grid={PLACE_GRID} n={n:.1e} exp={bits:2d} total {total:7.3f} ms | level0 {m[0]:5.2f} level1 {m[1]:5.2f} plan2 {m[2]:5.2f} local stage {m[3]:5.2f} ms | "
              f"cursor state {state.value} cells left to k_local_sort {todo.value}", flush=True)
    L.lib.gx_sort_set_experiment(0)
    return sums


if which in ("keys", "both"):
    out = Column.empty(KDT, n)
    nb = ctypes.c_size_t(0)
    L.check(L.lib.gx_sort_keys(keys.gx, keys.data_ptr, out.data_ptr, n, 0, None, ctypes.byref(nb), sp), "query")
    tmp = ops.device_bytes(nb.value)
    sums = measure("sort_keys " + kind, lambda: L.check(L.lib.gx_sort_keys(keys.gx, keys.data_ptr, out.data_ptr, n, 0, ops.ptr(tmp), ctypes.byref(nb), sp), "sort"), tmp, out)
    assert sums[0] == sums[32] and sums[0][2] == 0 and sums[0][:2] == ops.checksum(keys)[:2], sums
    del tmp, out
if which in ("order", "both"):
    out = Column.empty(np.int32, n)
    nb = ctypes.c_size_t(0)
    L.check(L.lib.gx_sorted_order(keys.gx, keys.data_ptr, None, n, 0, 0, 1, out.data_ptr, None, ctypes.byref(nb), sp), "query")
    tmp = ops.device_bytes(nb.value)
    sums = measure("sorted_order " + kind, lambda: L.check(L.lib.gx_sorted_order(keys.gx, keys.data_ptr, None, n, 0, 0, 1, out.data_ptr, ops.ptr(tmp), ctypes.byref(nb), sp), "order"), tmp, out)
    assert sums[0][:2] == sums[32][:2], sums
    srt = ops.gather(keys, out)
    assert ops.checksum(srt)[2] == 0
    del tmp, out
print("ok")
