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
    D.close_communicators()
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
    hj = D.DistributedHashJoin(torch.from_numpy(right).cuda())           # build once ...
    for sl in (slice(None), slice(0, len(left) // 3)):                   # ... probe twice
        hl, hr = hj.inner_join(torch.from_numpy(left[sl].copy()).cuda())
        el_, er_ = orc.inner_join(left[sl], right)
        a, b = orc.canonical_pairs(hl.cpu().numpy(), hr.cpu().numpy())
        np.testing.assert_array_equal(a, el_)
        np.testing.assert_array_equal(b, er_)
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


@pytest.mark.parametrize("dtype", ["int64", "uint64", "float64", "int32"])
def test_partition_rows_range_and_hash(dtype):
    """gx_partition_rows, the one-pass partition a rank runs before the all-to-all: range mode sends a key to
    (number of splitters <= key) in cudf sort order; hash mode groups equal keys, returns a permutation and uses
    bits that are independent of the join table's slot bits."""
    import torch
    import cudf_amd  # noqa: F401
    from cudf_amd import Column, ops
    rng = np.random.default_rng(8)
    for n in (0, 1, 1000, 300_007, 5_000_011):
        if dtype == "float64":
            v = rng.standard_normal(n) * 1e6
            if n > 10:
                v[::97] = -0.0
                v[5::101] = np.inf
        elif dtype == "int32":
            v = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
        else:
            v = rng.integers(-2**62 if dtype == "int64" else 0, 2**62, n).astype(dtype)
        col = Column.from_numpy(v)
        for nparts in (1, 2, 3, 8, 16):
            sp = np.sort(rng.choice(v, nparts - 1)) if n >= nparts and nparts > 1 else np.zeros(nparts - 1, v.dtype)
            pk, rows, offs = ops.partition_rows(col, nparts, splitters=sp.tolist(), want_rows=True)
            got, r = pk.to_numpy(), rows.to_numpy()
            assert offs[0] == 0 and offs[-1] == n and all(a <= b for a, b in zip(offs, offs[1:]))
            assert np.array_equal(np.sort(r), np.arange(n))            # a permutation ...
            assert got.tobytes() == v[r].tobytes()                       # ... that carries the keys
            dest = np.searchsorted(sp, v, side="right")
            for j in range(nparts):
                assert np.all(dest[r[offs[j]:offs[j + 1]]] == j), (dtype, n, nparts, j)
        if np.dtype(dtype).itemsize == 8 or dtype == "int32":
            for nparts in (1, 2, 8):
                pk, rows, offs = ops.partition_rows(col, nparts)
                got, r = pk.to_numpy(), rows.to_numpy()
                assert offs[-1] == n and np.array_equal(np.sort(r), np.arange(n)) and got.tobytes() == v[r].tobytes()
                if n >= 1000 and dtype != "float64":
                    seen = {}
                    for j in range(nparts):                               # equal keys land in one group
                        for k in np.unique(got[offs[j]:offs[j + 1]][:2000]):
                            assert seen.setdefault(int(k), j) == j
                    if nparts == 8 and n > 100_000:
                        sizes = np.diff(offs)
                        assert sizes.min() > 0.8 * n / 8 and sizes.max() < 1.2 * n / 8, sizes   # balanced


def test_merge_sum_count_matches_numpy():
    import cudf_amd  # noqa: F401
    from cudf_amd import Column, ops
    rng = np.random.default_rng(9)
    k = rng.integers(0, 5000, 40_000).astype(np.int32)
    s = rng.integers(0, 1000, 40_000).astype(np.float64)
    c = rng.integers(1, 50, 40_000).astype(np.int64)
    mk, ms, mc = ops.merge_sum_count(Column.from_numpy(k), Column.from_numpy(s), Column.from_numpy(c))
    uk = np.unique(k)
    np.testing.assert_array_equal(mk.to_numpy(), uk)
    np.testing.assert_array_equal(ms.to_numpy(), np.bincount(k, weights=s)[uk])
    np.testing.assert_array_equal(mc.to_numpy(), np.bincount(k, weights=c)[uk].astype(np.int64))


def test_gxd_sharded_operators_chunked_forced_exchange(pg):
    """The C++/RCCL product path (include/cudf_amd/gxd.h) at sizes where it CHUNKS: 6e6-row shards in several chunks, the
    chunked partitioned probe against a 2^21-slot table, duplicate build keys that blow the first output-size guess, float and
    32-bit sorts, chunk-relative row ids (gx_partition_rows_at / gx_join_probe_partitioned_at) and the chunk x rank segment table
    (gx_gather_global_rows_dev)."""
    import torch
    from cudf_amd import gxd
    D = pg
    comm = D._gxd_comm(None)
    rng = np.random.default_rng(77)
    # slots of the speculative partition passes shrunk below the chunk size: every chunk overflows and is re-partitioned exactly
    gxd.set_slot_scale(0.4)
    try:
        v = rng.integers(-2**62, 2**62, 3_000_017, dtype=np.int64)
        out = comm.sort(torch.from_numpy(v).cuda(), chunks=3, force_exchange=True)
        assert out.cpu().numpy().tobytes() == np.sort(v).tobytes()
        b2 = rng.permutation(900_000)[:400_000].astype(np.int64)
        p2 = rng.integers(0, 1_200_000, 2_500_000).astype(np.int64)
        hj2 = gxd.HashJoin(comm, torch.from_numpy(b2).cuda(), force_exchange=True)
        l2, r2 = hj2.inner_join(torch.from_numpy(p2).cuda(), chunks=2)
        a2, c2 = orc.canonical_pairs(l2.cpu().numpy(), r2.cpu().numpy())
        e2l, e2r = orc.inner_join(p2, b2)
        np.testing.assert_array_equal(a2, e2l)
        np.testing.assert_array_equal(c2, e2r)
        hj2.close()
    finally:
        gxd.set_slot_scale(0.0)
    for v in (rng.integers(-2**62, 2**62, 6_000_011, dtype=np.int64), (rng.standard_normal(3_000_001) * 1e3).astype(np.float64),
              rng.integers(-2**31, 2**31 - 1, 2_500_000).astype(np.int32), np.zeros(0, np.int64), np.arange(5, dtype=np.int64)[::-1].copy()):
        out = comm.sort(torch.from_numpy(v).cuda(), chunks=4, force_exchange=True)
        assert out.cpu().numpy().tobytes() == np.sort(v).tobytes()
    nb, npr = 700_000, 6_000_013
    build = rng.permutation(3 * nb)[:nb].astype(np.int64)
    build[:90_000] = np.tile(build[100_000:130_000], 3)          # 30 000 build keys occur four times
    probe = rng.integers(0, 4 * nb, npr).astype(np.int64)
    probe[::3] = build[rng.integers(0, 30_000, len(probe[::3]))]  # a third of the probe rows hit them: pairs > rows
    hj = gxd.HashJoin(comm, torch.from_numpy(build).cuda(), force_exchange=True)
    el, er = orc.inner_join(probe, build)
    assert len(el) > npr                                          # the first capacity guess (rows) is too small: the re-probe runs
    for chunks in (5, 1):
        l, r = hj.inner_join(torch.from_numpy(probe).cuda(), chunks=chunks)
        a, b = orc.canonical_pairs(l.cpu().numpy(), r.cpu().numpy())
        np.testing.assert_array_equal(a, el)
        np.testing.assert_array_equal(b, er)
    hj.close()
    gk = rng.integers(0, 200_000, 4_000_000).astype(np.int64)
    gv = rng.integers(0, 100, 4_000_000).astype(np.int32)
    k, s, c = comm.groupby_sum_count(torch.from_numpy(gk).cuda(), torch.from_numpy(gv).cuda(), max_groups=1000, force_exchange=True)
    uk = np.unique(gk)
    np.testing.assert_array_equal(k.cpu().numpy(), uk)
    np.testing.assert_array_equal(s.cpu().numpy(), np.bincount(gk, weights=gv)[uk].astype(np.int64))
    np.testing.assert_array_equal(c.cpu().numpy(), np.bincount(gk)[uk])


@pytest.mark.parametrize("mode", ["hash", "range"])
def test_partition_rows_speculative_form(mode):
    """gx_partition_rows_spec_at: the one-read partition pass of the sharded operators -- group g in the fixed slot
    [g, g + 1) * cap, rows per group instead of starts, the overflow flag when a group outgrows its slot."""
    import ctypes
    import torch
    import cudf_amd  # noqa: F401
    from cudf_amd import Column, _lib as L
    from cudf_amd.column import device_bytes, ptr, stream_ptr
    rng = np.random.default_rng(12)
    n, W = 3_000_017, 8
    v = rng.integers(-2**62, 2**62, n, dtype=np.int64)
    col = Column.from_numpy(v)
    sp = np.sort(rng.choice(v, W - 1)).astype(np.int64)
    spp = sp.ctypes.data_as(ctypes.c_void_p) if mode == "range" else None
    # (random splitters make uneven groups: only a slot of n rows can never overflow in range mode)
    for cap, expect_overflow in (((n + 31) // 32 * 32 if mode == "range" else n // W * 2, False), (n // W // 2 // 32 * 32, True)):
        ok = Column.empty(np.int64, W * cap)
        orow = Column.empty(np.int32, W * cap)
        offs = torch.zeros(W + 1, dtype=torch.int64, device="cuda")
        nb = ctypes.c_size_t(0)
        args = lambda t: (col.gx, col.data_ptr, n, 1000, 1 if mode == "range" else 0, W, spp, cap, ok.data_ptr, orow.data_ptr, ptr(offs), t,
                          ctypes.byref(nb), stream_ptr())
        L.check(L.lib.gx_partition_rows_spec_at(*args(None)), "query")
        tmp = device_bytes(nb.value)
        L.check(L.lib.gx_partition_rows_spec_at(*args(ptr(tmp))), "partition")
        f = offs.cpu().numpy()
        assert bool(f[W]) == expect_overflow
        if expect_overflow:
            assert np.all(f[:W] <= cap)
            continue
        assert not bool(f[W])
        assert int(f[:W].sum()) == n and np.all(f[:W] <= cap)
        keys_out, rows_out = ok.to_numpy(), orow.to_numpy()
        dest = np.searchsorted(sp, v, side="right") if mode == "range" else None
        seen = np.zeros(n, bool)
        for g in range(W):
            r = rows_out[g * cap: g * cap + f[g]] - 1000         # row_base was 1000
            assert np.all((r >= 0) & (r < n)) and not seen[r].any()
            seen[r] = True
            assert keys_out[g * cap: g * cap + f[g]].tobytes() == v[r].tobytes()
            if mode == "range":
                assert np.all(dest[r] == g)
        assert seen.all()
