#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out
L=$O/run19.log
: > $L
for e in 0 4 8 12; do
  echo "== GX_EXP $e" >> $L
  GX_EXP=$e python - >> $L 2>&1 <<'PY'
import numpy as np, torch, time, ctypes
from cudf_amd import Column, ops, _lib as L
from cudf_amd.column import device_bytes, ptr, stream_ptr
n = 1_000_000_000
col = ops.random_column(np.int64, n, seed=42)
out = Column.empty(np.int64, n)
nb = ctypes.c_size_t(0)
args = (col.gx, col.data_ptr, out.data_ptr, n, 0)
L.check(L.lib.gx_sort_keys(*args, None, ctypes.byref(nb), stream_ptr()), "q")
tmp = device_bytes(nb.value)
L.lib.gx_sort_profile(1)
for i in range(3):
    L.check(L.lib.gx_sort_keys(*args, ptr(tmp), ctypes.byref(nb), stream_ptr()), "s")
    h4 = (ctypes.c_float * 4)(); L.lib.gx_sort_profile_read_hybrid(h4)
print("   local sort ms", round(h4[3], 2), "msd", round(h4[0], 2), round(h4[2], 2))
PY
done
grep -h "==\|local sort" $L
