"""Diagnosis (not product code): per-kernel cost of a forced gxd_join_probe step.  Run under rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import cudf_amd
from cudf_amd import ops, gxd

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nb = n // 10
bk = torch.randperm(nb, device="cuda") * 3 + 1
pcol = ops.random_column(np.int64, n, seed=2, lo=0, hi=int(nb / 0.3))
pk = pcol.data[: n * 8].view(torch.int64)
pk.mul_(3).add_(1)
comm = gxd.Communicator()
hj = gxd.HashJoin(comm, bk, force_exchange=True)
for rep in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter()
    l, r = hj.inner_join(pk, chunks=chunks)
    torch.cuda.synchronize()
    print(f"chunks {chunks} rep {rep}: {(time.perf_counter() - t) * 1e3:.2f} ms  pairs {l.numel()}  timing {comm.last_timing()}", flush=True)
    del l, r
hj.close()
comm.close()
