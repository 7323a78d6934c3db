"""GPU parity: reduce / scan / murmur3 / hash_partition / bitmask utilities vs the CPU oracle."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cudf_oracle as orc
from tests.golden import reference_vectors as gv


@pytest.fixture(scope="module")
def gx():
    import torch
    assert torch.cuda.is_available()
    import cudf_amd
    from cudf_amd import Column, ops
    return Column, ops


DTYPES = ["int8", "int16", "int32", "int64", "uint8", "uint32", "uint64", "float32", "float64"]


def _vals(dtype, n, rng):
    dt = np.dtype(dtype)
    if dt.kind == "f":
        return (rng.random(n) * 2000 - 1000).astype(dt)
    info = np.iinfo(dt)
    return rng.integers(info.min, info.max, n, dtype=dt, endpoint=True)


@pytest.mark.parametrize("dtype", DTYPES)
def test_scan_matches_oracle(gx, dtype):
    Column, ops = gx
    rng = np.random.default_rng(5)
    for n in [0, 1, 63, 4096, 4097, 300_007]:
        v = _vals(dtype, n, rng)
        valid = rng.random(n) > 0.15
        for op in ("sum", "min", "max"):
            for inclusive in (True, False):
                for m in (None, valid):
                    for include in ((False, True) if m is not None else (False,)):
                        out = ops.scan(Column.from_numpy(v, m), op, inclusive, include)
                        ev, em = orc.scan(v, op, inclusive, m, include)
                        if m is not None:
                            np.testing.assert_array_equal(out.valid_numpy(), em)
                        got = out.to_numpy()
                        assert got.dtype == v.dtype
                        sel = em if m is not None else np.ones(n, bool)
                        if np.dtype(dtype).kind == "f" and op == "sum":
                            # tolerance: the oracle is a sequential np.cumsum in the column dtype; ours is a
                            # compensated (double-double) prefix rounded once -- compare against a long-double cumsum
                            x = np.where(sel if m is not None else True, v, 0).astype(np.longdouble)
                            ref = np.cumsum(x)
                            if not inclusive:
                                ref = np.concatenate([[0], ref[:-1]])
                            tol = 1 if dtype == "float64" else None
                            if tol:
                                assert np.all(orc.ulp_diff(got[sel], ref[sel].astype(np.float64)) <= 1)
                            else:
                                np.testing.assert_allclose(got[sel], ref[sel].astype(np.float32), rtol=3e-7, atol=1e-3)
                        else:
                            np.testing.assert_array_equal(got[sel], ev[sel], err_msg=f"{dtype} {n} {op} {inclusive}")


@pytest.mark.parametrize("dtype", ["int8", "int32", "int64", "uint32", "float32", "float64"])
@pytest.mark.parametrize("case", gv.SCAN, ids=lambda c: c["name"])
def test_reference_golden_scan(gx, case, dtype):
    Column, ops = gx
    vals, mask = gv.col(case["values"], dtype, case["valid"])
    out = ops.scan(Column.from_numpy(vals, mask), case["op"], case["inclusive"], case["null_include"])
    ev = np.array(case["expect_valid"], bool)
    if mask is not None:
        gm = out.valid_numpy()
        np.testing.assert_array_equal(np.ones(len(vals), bool) if gm is None else gm, ev)
    np.testing.assert_array_equal(out.to_numpy()[ev], np.array(case["expect"])[ev].astype(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_reduce_matches_oracle(gx, dtype):
    Column, ops = gx
    rng = np.random.default_rng(6)
    for n in [0, 1, 1000, 500_003]:
        v = _vals(dtype, n, rng)
        valid = rng.random(n) > 0.3
        for m in (None, valid, np.zeros(n, bool)):
            for op in ("sum", "min", "max"):
                got, ok = ops.reduce(Column.from_numpy(v, m), op)
                out_dt = None
                if op == "sum":
                    out_dt = np.float64 if np.dtype(dtype).kind == "f" else (np.uint64 if np.dtype(dtype).kind == "u" else np.int64)
                exp, eok = orc.reduce(v, op, m, out_dt)
                assert ok == eok, (dtype, n, op)
                if not ok:
                    continue
                if np.dtype(dtype).kind == "f" and op == "sum":
                    # tolerance 1 ulp of the exact (fsum) result (north_star)
                    assert orc.ulp_diff(np.array([got]), np.array([exp]))[0] <= 1
                else:
                    assert got == exp, (dtype, n, op, got, exp)


@pytest.mark.parametrize("dtype", ["int8", "int32", "int64", "float32", "float64"])
@pytest.mark.parametrize("case", gv.REDUCE, ids=lambda c: c["name"])
def test_reference_golden_reduce(gx, case, dtype):
    Column, ops = gx
    vals, mask = gv.col(case["values"], dtype, case["valid"])
    got, ok = ops.reduce(Column.from_numpy(vals, mask), case["op"])
    assert ok == case["expect_valid"]
    if ok:
        assert got == case["expect"]


@pytest.mark.parametrize("dtype", ["int8", "uint8", "int16", "int32", "uint32", "int64", "uint64", "float32", "float64", "bool"])
def test_murmur3_matches_oracle(gx, dtype):
    Column, ops = gx
    rng = np.random.default_rng(7)
    n = 10_007
    if dtype == "bool":
        v = rng.integers(0, 2, n).astype(bool)
    elif np.dtype(dtype).kind == "f":
        v = rng.standard_normal(n).astype(dtype)
        v[:4] = [0.0, -0.0, np.nan, -np.nan]
    else:
        v = _vals(dtype, n, rng)
    valid = rng.random(n) > 0.2
    for seed in (0, 619):
        got = ops.murmurhash3_x86_32([Column.from_numpy(v, valid)], seed).to_numpy()
        np.testing.assert_array_equal(got, orc.murmur3_32(v, seed, valid))
    # multi-column row hash (primitive_row_operators.cuh:247-268)
    w = rng.integers(-5, 5, n).astype(np.int64)
    got = ops.murmurhash3_x86_32([Column.from_numpy(v), Column.from_numpy(w)]).to_numpy()
    np.testing.assert_array_equal(got, orc.row_hash([v, w]))


@pytest.mark.parametrize("dtype", ["int8", "uint8", "int16", "uint16", "int32", "uint32", "int64", "uint64", "float32", "float64", "bool"])
def test_identity_hash_matches_oracle(gx, dtype):
    """gx_identity_hash_32 = IdentityHash<T> (partitioning.cu:852-872): the cast to uint32, bit-exact incl. what the device conversion
    does outside C++'s defined range; the column fold and the null value of every row hasher"""
    Column, ops = gx
    rng = np.random.default_rng(17)
    n = 10_007
    if dtype == "bool":
        v = rng.integers(0, 2, n).astype(bool)
    elif np.dtype(dtype).kind == "f":
        v = (rng.standard_normal(n) * 10.0 ** rng.integers(0, 12, n)).astype(dtype)
        v[:10] = np.array([0.0, -0.0, np.nan, -np.nan, np.inf, -np.inf, 4294967296.0, 4294967040.0, -0.99, 0.99], dtype)
    else:
        v = _vals(dtype, n, rng)
    valid = rng.random(n) > 0.2
    got = ops.identity_hash([Column.from_numpy(v, valid)]).to_numpy()
    np.testing.assert_array_equal(got, orc.identity_hash32(v, valid))
    w = rng.integers(-2**40, 2**40, n).astype(np.int64)
    got = ops.identity_hash([Column.from_numpy(v), Column.from_numpy(w, valid)]).to_numpy()
    np.testing.assert_array_equal(got, orc.row_hash_identity([v, w], [None, valid]))
    m, offs = ops.hash_partition_map([Column.from_numpy(v), Column.from_numpy(w)], 37, hash_function="identity")
    eo, eoffs = orc.hash_partition([v, w], 37, hash_function="identity")
    np.testing.assert_array_equal(offs, eoffs)
    np.testing.assert_array_equal(m.to_numpy(), eo)


@pytest.mark.parametrize("nparts", [1, 2, 3, 8, 200, 1000])
def test_hash_partition_matches_oracle(gx, nparts):
    Column, ops = gx
    rng = np.random.default_rng(8)
    for n in [0, 1, 5000, 250_001]:
        k = rng.integers(-10**6, 10**6, n).astype(np.int64)
        m, offs = ops.hash_partition_map([Column.from_numpy(k)], nparts)
        eo, eoffs = orc.hash_partition([k], nparts)
        np.testing.assert_array_equal(offs, eoffs)
        np.testing.assert_array_equal(m.to_numpy(), eo)


def test_bitmask_utilities(gx):
    import torch
    from cudf_amd import _lib as L
    from cudf_amd.column import pack_mask, ptr, stream_ptr, unpack_mask
    Column, ops = gx
    rng = np.random.default_rng(9)
    n = 100_003
    a = rng.random(n) > 0.4
    b = rng.random(n) > 0.1
    ma = torch.from_numpy(pack_mask(a).view(np.int32)).cuda()
    mb = torch.from_numpy(pack_mask(b).view(np.int32)).cuda()
    assert ops.bitmask_count(ma, n) == int(a.sum())
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    L.check(L.lib.gx_bitmask_count(ptr(ma), 37, 90_001, ptr(cnt), stream_ptr()))
    assert int(cnt.item()) == int(a[37:90_001].sum())
    out = torch.zeros_like(ma)
    import ctypes
    arr = (ctypes.c_void_p * 3)(ma.data_ptr(), None, mb.data_ptr())
    L.check(L.lib.gx_bitmask_and(arr, 3, n, ptr(out), ptr(cnt), stream_ptr()))
    assert int(cnt.item()) == int((a & b).sum())
    np.testing.assert_array_equal(unpack_mask(out.cpu().numpy().view(np.uint32), n), a & b)
    L.check(L.lib.gx_bitmask_set(ptr(out), 5, 70_001, 1, stream_ptr()))
    L.check(L.lib.gx_bitmask_set(ptr(out), 100, 200, 0, stream_ptr()))
    exp = (a & b).copy()
    exp[5:70_001] = True
    exp[100:200] = False
    np.testing.assert_array_equal(unpack_mask(out.cpu().numpy().view(np.uint32), n), exp)
    pos = torch.zeros(1, dtype=torch.int64, device="cuda")
    L.check(L.lib.gx_bitmask_first_unset(ptr(out), n, ptr(pos), stream_ptr()))
    assert int(pos.item()) == int(np.argmin(exp))
