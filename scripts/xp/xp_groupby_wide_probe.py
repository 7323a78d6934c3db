"""Measurement / diagnosis: gx_groupby_sum_count_wide at 1e8 and 1e9 rows, 2 x int64 keys in [0, 1000)^2: raw *ngroups, time."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import cudf_amd
from cudf_amd import Column, ops, _lib as L

for n in [int(float(x)) for x in sys.argv[1:]] or [100_000_000]:
    k0 = ops.random_column(np.int64, n, seed=21, lo=0, hi=1000)
    k1 = ops.random_column(np.int64, n, seed=22, lo=0, hi=1000)
    gv = ops.random_column(np.float64, n, seed=23)
    mg = 1 << 20
    outs = [Column.empty(np.int64, mg) for _ in range(2)]
    osum, ocv = Column.empty(np.float64, mg), Column.empty(np.int32, mg)
    kp = (ctypes.c_void_p * 2)(k0.data_ptr.value, k1.data_ptr.value)
    op = (ctypes.c_void_p * 2)(outs[0].data_ptr.value, outs[1].data_ptr.value)
    ng = torch.zeros(1, dtype=torch.int64, device="cuda")
    nb = ctypes.c_size_t(0)
    L.check(L.lib.gx_groupby_sum_count_wide(2, kp, gv.gx, gv.data_ptr, n, mg, op, osum.data_ptr, ocv.data_ptr, ops.ptr(ng), None, ctypes.byref(nb), ops.stream_ptr()), "query")
    tmp = ops.device_bytes(nb.value)
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        L.check(L.lib.gx_groupby_sum_count_wide(2, kp, gv.gx, gv.data_ptr, n, mg, op, osum.data_ptr, ocv.data_ptr, ops.ptr(ng), ops.ptr(tmp), ctypes.byref(nb), ops.stream_ptr()), "run")
        torch.cuda.synchronize()
        print(f"n={n:.0e} it={it}: {(time.perf_counter() - t0) * 1e3:.2f} ms, ngroups={int(ng.item())}, tmp={nb.value / 1e9:.2f} GB", flush=True)
    g = int(ng.item())
    if g > 0:
        cnt = ocv.data[: g * 4].view(torch.int32).to(torch.int64).sum().item()
        print("  sum(count) =", cnt, "rows", n)
    del tmp
