"""Measurement: where do k_local_sort's 5.5 ms go?  gx_sort_keys of 1e9 random int64 keys under the kernel's ablation bits
(gx_sort_set_experiment; outputs are NOT sorted under them), per-kernel times from gx_sort_profile_read_hybrid."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import cudf_amd
from cudf_amd import Column, ops, _lib as L

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
keys = ops.random_column(np.int64, n, seed=42)
out = Column.empty(np.int64, n)
nb = ctypes.c_size_t(0)
L.check(L.lib.gx_sort_keys(keys.gx, keys.data_ptr, out.data_ptr, n, 0, None, ctypes.byref(nb), ops.stream_ptr()), "query")
tmp = ops.device_bytes(nb.value)
L.lib.gx_sort_profile(1)
for bits, what in [(0, "production"), (4, "no sub-bucket sorts"), (8, "no LDS atomics in the split"), (12, "neither: load, scatter through LDS, store"),
                   (16, "networks instead of the counting split")]:
    L.lib.gx_sort_set_experiment(bits)
    ms = []
    for it in range(4):
        L.check(L.lib.gx_sort_keys(keys.gx, keys.data_ptr, out.data_ptr, n, 0, ops.ptr(tmp), ctypes.byref(nb), ops.stream_ptr()), "sort")
        h4 = (ctypes.c_float * 4)()
        if L.lib.gx_sort_profile_read_hybrid(h4) == 0 and it > 0:
            ms.append(h4[3])
    print(f"exp={bits:2d} {what:45s} k_local_sort {sum(ms) / len(ms):6.3f} ms", flush=True)
L.lib.gx_sort_set_experiment(0)
