"""Measurement: where do 225 ms per gxd_sort call go in xp_distributed_single_rank.py (result allocation through the
torch callback)?  Times torch.empty(8 GB) in a loop, then gxd sorts with / without the python-path warm-up before."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
dist.init_process_group("nccl", rank=0, world_size=1)
import cudf_amd
from cudf_amd import ops, distributed as D, gxd

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000


def t_empty(tag):
    for i in range(4):
        t0 = time.perf_counter()
        x = torch.empty(n * 8, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        print(f"{tag}: torch.empty(8 GB) #{i}: {(time.perf_counter() - t0) * 1e3:.2f} ms; reserved {torch.cuda.memory_reserved() / 1e9:.1f} GB", flush=True)
        del x


t_empty("fresh")
col = ops.random_column(np.int64, n, seed=1)
keys = col.data[: n * 8].view(torch.int64)
comm = D._gxd_comm(None)
for i in range(4):
    t0 = time.perf_counter()
    r = comm.sort(keys, chunks=4, force_exchange=True)
    torch.cuda.synchronize()
    print(f"gxd_sort #{i}: {(time.perf_counter() - t0) * 1e3:.2f} ms  timing {comm.last_timing()}; reserved {torch.cuda.memory_reserved() / 1e9:.1f} GB; free {torch.cuda.mem_get_info()[0] / 1e9:.1f} GB", flush=True)
    del r
t_empty("after gxd sorts")
local = D.HipLocalOps()
D._FORCE_EXCHANGE = True
r = D.distributed_sort(keys, local=local)
torch.cuda.synchronize()
del r
print(f"after python path: reserved {torch.cuda.memory_reserved() / 1e9:.1f} GB; free {torch.cuda.mem_get_info()[0] / 1e9:.1f} GB", flush=True)
torch.cuda.empty_cache()
print(f"after empty_cache: reserved {torch.cuda.memory_reserved() / 1e9:.1f} GB; free {torch.cuda.mem_get_info()[0] / 1e9:.1f} GB", flush=True)
for i in range(4):
    t0 = time.perf_counter()
    r = comm.sort(keys, chunks=4, force_exchange=True)
    torch.cuda.synchronize()
    print(f"gxd_sort after python path #{i}: {(time.perf_counter() - t0) * 1e3:.2f} ms; reserved {torch.cuda.memory_reserved() / 1e9:.1f} GB", flush=True)
    del r
nb = n // 10
bk = torch.randperm(nb, device="cuda") * 3 + 1
for i in range(3):
    t0 = time.perf_counter()
    gj = gxd.HashJoin(comm, bk, force_exchange=True)
    torch.cuda.synchronize()
    print(f"gxd_join_build #{i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
    gj.close()
D.close_communicators()
dist.destroy_process_group()
