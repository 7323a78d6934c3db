"""Diagnosis (not product code): which kernel of a chunked gxd_sort is slow.  Run under rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import cudf_amd
from cudf_amd import ops, gxd

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
col = ops.random_column(np.int64, n, seed=1)
keys = col.data[: n * 8].view(torch.int64)
comm = gxd.Communicator()
for rep in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = comm.sort(keys, chunks=chunks, force_exchange=True)
    torch.cuda.synchronize()
    print(f"chunks {chunks} rep {rep}: {(time.perf_counter() - t) * 1e3:.2f} ms  timing {comm.last_timing()}", flush=True)
    del out
comm.close()
