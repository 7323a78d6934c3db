"""GPU: the distributed operators with the HIP LocalOps on a 1-rank RCCL group, forced through the
partition + all_to_all_single path (the multi-rank exchange logic itself is covered on CPU with
gloo in tests/test_distributed_cpu.py; the 8-GPU run is the driver's)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cudf_oracle as orc


@pytest.fixture(scope="module")
def pg():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from cudf_amd import distributed as D
    D._FORCE_EXCHANGE = True
    yield D
    D._FORCE_EXCHANGE = False
    dist.destroy_process_group()


def test_distributed_ops_single_rank_rccl(pg):
    import torch
    D = pg
    rng = np.random.default_rng(3)
    keys = rng.integers(-2**62, 2**62, 300_000, dtype=np.int64)
    out = D.distributed_sort(torch.from_numpy(keys).cuda())
    np.testing.assert_array_equal(out.cpu().numpy(), np.sort(keys))
    left = rng.integers(0, 50_000, 200_000).astype(np.int64)
    right = rng.permutation(60_000)[:30_000].astype(np.int64)
    gl, gr = D.distributed_inner_join(torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda())
    el, er = orc.inner_join(left, right)
    cl, cr = orc.canonical_pairs(gl.cpu().numpy(), gr.cpu().numpy())
    np.testing.assert_array_equal(cl, el)
    np.testing.assert_array_equal(cr, er)
    gk = rng.integers(0, 5000, 400_000).astype(np.int32)
    gv = rng.integers(0, 100, 400_000).astype(np.float64)
    k, s, c = D.distributed_groupby_sum_count(torch.from_numpy(gk).cuda(), torch.from_numpy(gv).cuda())
    uk = np.unique(gk)
    np.testing.assert_array_equal(k.cpu().numpy(), uk)
    np.testing.assert_array_equal(s.cpu().numpy(), np.bincount(gk, weights=gv)[uk])
    np.testing.assert_array_equal(c.cpu().numpy(), np.bincount(gk)[uk])


def test_distributed_reduce_scan_single_rank_rccl(pg):
    """reduce = all-gather of one partial per GPU, scan = exclusive prefix of the shard
    totals; here with the HIP LocalOps on the 1-rank RCCL group (the multi-rank folding runs under gloo)."""
    import torch
    D = pg
    rng = np.random.default_rng(4)
    v = rng.integers(-1000, 1000, 300_001).astype(np.int64)
    t = torch.from_numpy(v).cuda()
    assert D.distributed_reduce(t, "sum") == int(v.sum())
    assert D.distributed_reduce(t, "min") == int(v.min()) and D.distributed_reduce(t, "max") == int(v.max())
    f = (v / 4).astype(np.float64)
    assert D.distributed_reduce(torch.from_numpy(f).cuda(), "sum") == float(f.sum())   # quarters: exact
    np.testing.assert_array_equal(D.distributed_scan(t, "sum", True).cpu().numpy(), np.cumsum(v))
    ex = D.distributed_scan(t, "max", False).cpu().numpy()
    np.testing.assert_array_equal(ex[1:], np.maximum.accumulate(v)[:-1])
    assert D.distributed_reduce(t[:0], "sum") == 0
