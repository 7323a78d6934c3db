#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out
L=$O/run16.log
: > $L
timeout 900 python -m pytest tests/test_gpu_sort.py -m gpu -x -q > $O/pytest_gpu16.log 2>&1
echo "pytest sort exit $?" | tee -a $L
tail -4 $O/pytest_gpu16.log | tee -a $L
python - >> $L 2>&1 <<'PY'
import numpy as np, torch, time, ctypes
from cudf_amd import Column, ops, _lib as L
from cudf_amd.column import device_bytes, ptr, stream_ptr
n = 1_000_000_000
# float64 keys with uniformly random bit patterns in the finite range: random doubles via int bits, NaNs removed
k = ops.random_column(np.int64, n, seed=5)
t = k.data[: n * 8].view(torch.int64)
t.bitwise_and_(-1 ^ (1 << 62))   # clear the top exponent bit -> finite values, both signs
col = Column(k.data, np.float64, n)
out = Column.empty(np.float64, n)
nb = ctypes.c_size_t(0)
args = (col.gx, col.data_ptr, out.data_ptr, n, 0)
L.check(L.lib.gx_sort_keys(*args, None, ctypes.byref(nb), stream_ptr()), "q")
tmp = device_bytes(nb.value)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    L.check(L.lib.gx_sort_keys(*args, ptr(tmp), ctypes.byref(nb), stream_ptr()), "s")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
info = (ctypes.c_int32 * 8)(); L.lib.gx_sort_info(ptr(tmp), info, stream_ptr())
print("float64 1e9 sort ms", round(dt * 1e3, 2), "Grows/s", round(n / dt / 1e9, 2), list(info), ops.checksum(out)[2])
PY
tail -2 $L
