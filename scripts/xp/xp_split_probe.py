"""Measurement helper (not product code): which path the sort takes on a few value distributions at a given size, with the result
checked against numpy.  Usage: xp_split_probe.py [rows]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import cudf_amd  # noqa: F401
from cudf_amd import Column, ops, _lib as L

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_sort_splitters import _keys  # the test's distributions

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 40_000_003
for kind in ("normal", "normal_tail", "lognormal", "zipf", "clusters", "normal_hot", "steps"):
    rng = np.random.default_rng(5)
    v = _keys(kind, rng, n)
    col = Column.from_numpy(v)
    out = Column.empty(v.dtype, v.size)
    tmp = ops._run(L.lib.gx_sort_keys, col.gx, col.data_ptr, out.data_ptr, col.size, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tmp = ops._run(L.lib.gx_sort_keys, col.gx, col.data_ptr, out.data_ptr, col.size, 0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    st = ctypes.c_int32(-1)
    L.lib.gx_sort_cursor_state(ops.ptr(tmp), ctypes.byref(st), ops.stream_ptr())
    info = (ctypes.c_int32 * 4)()
    L.lib.gx_sort_split_info(ops.ptr(tmp), info, ops.stream_ptr())
    big = (ctypes.c_int64 * 3)()
    L.lib.gx_sort_big_info(ops.ptr(tmp), big, ops.stream_ptr())
    ok = out.to_numpy().tobytes() == np.sort(v).tobytes()
    print(f"{kind:12s} n={n} state={st.value} split={list(info)} big={list(big)} {ms:7.2f} ms (incl. scratch alloc) correct={ok}", flush=True)
