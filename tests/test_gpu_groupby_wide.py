"""groupby on SEVERAL key columns in one partition pass, rows compared inside the LDS tables (gx_groupby_sum_count_wide;
the reference: hash the row once, compare rows in the probe -- primitive_row_operators.cuh:95-163, 247-268 -- under
cudf::groupby::aggregate, groupby.hpp:189-240).  Oracle: oracle/cudf_oracle.py::groupby_agg on the rows' key tuples folded
into one int64 (the test's key ranges allow it), sums to 1 ulp of the correctly rounded sum, counts and keys exact.  The device's
own verdicts are pinned too: skewed keys must come back as "take the fallback" (-2) and still produce the right result
through ops.groupby_sum_count_tables."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cudf_oracle as orc


@pytest.fixture(scope="module")
def gx():
    import torch
    assert torch.cuda.is_available()
    import cudf_amd  # noqa: F401
    from cudf_amd import Column, ops
    return Column, ops


def _fold(cols, radices):
    f = np.zeros(len(cols[0]), np.int64)
    for c, r in zip(cols, radices):
        f = f * r + c.astype(np.int64)
    return f


def _check(Column, ops, keys, radices, offsets, vals, expect_wide=True):
    kc = [Column.from_numpy(k) for k in keys]
    vc = Column.from_numpy(vals)
    if expect_wide is not None:
        r = ops._groupby_sum_count_wide(kc, vc)
        assert (r is not None) == expect_wide, "the device's verdict on the one-pass path is not the expected one"
    gk, s, cv, ca = ops.groupby_sum_count_tables(kc, vc)
    got_keys = [g.to_numpy() for g in gk]
    for g, k in zip(got_keys, keys):
        assert g.dtype == k.dtype
    folded_in = _fold([k.astype(np.int64) + o for k, o in zip(keys, offsets)], radices)
    folded_out = _fold([g.astype(np.int64) + o for g, o in zip(got_keys, offsets)], radices)
    o = np.argsort(folded_out, kind="stable")
    ek, res = orc.groupby_agg(folded_in, vals, ["sum", "count_valid"], None, None)
    np.testing.assert_array_equal(folded_out[o], ek)
    np.testing.assert_array_equal(cv.to_numpy()[o], res["count_valid"][0])
    np.testing.assert_array_equal(ca.to_numpy()[o], res["count_valid"][0])
    if vals.dtype.kind == "f":
        assert np.all(orc.ulp_diff(s.to_numpy()[o], res["sum"][0]) <= 1)
    else:
        np.testing.assert_array_equal(s.to_numpy()[o], res["sum"][0])


@pytest.mark.parametrize("vdtype", ["float64", "int64", "int32"])
def test_two_int64_key_columns(gx, vdtype):
    Column, ops = gx
    rng = np.random.default_rng(1)
    n = 1_500_003
    k0 = rng.integers(-500, 500, n).astype(np.int64)
    k1 = rng.integers(0, 700, n).astype(np.int64)
    vals = (rng.random(n) * 200 - 50).astype(vdtype) if vdtype.startswith("f") else rng.integers(-1000, 1000, n).astype(vdtype)
    _check(Column, ops, [k0, k1], [1000, 700], [500, 0], vals)


def test_three_and_four_mixed_width_columns(gx):
    Column, ops = gx
    rng = np.random.default_rng(2)
    n = 900_017
    k0 = rng.integers(0, 40, n).astype(np.int32)
    k1 = rng.integers(0, 200, n).astype(np.uint8)
    k2 = rng.integers(-30, 30, n).astype(np.int64)
    k3 = rng.integers(0, 5, n).astype(np.int16)
    vals = rng.random(n) * 10
    _check(Column, ops, [k0, k1, k2], [40, 200, 60], [0, 0, 30], vals)
    _check(Column, ops, [k0, k1, k2, k3], [40, 200, 60, 5], [0, 0, 30, 0], vals)


def test_many_groups_split_tables_and_bound_retry(gx):
    """~1.2e6 groups out of 2e6 rows: more than the first max_groups bound (the -1 / retry protocol) and more groups per
    partition than half an LDS table (nsub > 1: a partition's keys dealt to several workgroups)"""
    Column, ops = gx
    rng = np.random.default_rng(3)
    n = 2_000_003
    k0 = rng.integers(0, 2000, n).astype(np.int64)
    k1 = rng.integers(0, 1500, n).astype(np.int64)
    vals = rng.random(n)
    _check(Column, ops, [k0, k1], [2000, 1500], [0, 0], vals)


def test_rows_that_share_hash_bits_but_differ(gx):
    """few distinct values per column, many rows: every slot sees a stream of equal tags; and two columns whose values are
    swapped between rows ((a, b) vs (b, a)) must stay different groups"""
    Column, ops = gx
    rng = np.random.default_rng(4)
    n = 700_001
    a = rng.integers(0, 50, n).astype(np.int64)
    b = rng.integers(0, 50, n).astype(np.int64)
    vals = rng.integers(-5, 5, n).astype(np.int64)
    _check(Column, ops, [a, b], [50, 50], [0, 0], vals)


def test_skewed_keys_stay_on_the_one_pass_path(gx):
    """half of the rows carry ONE key tuple.  The slot capacities come from a sample of the rows, so the hot partition simply
    gets a large slot (a fixed mean + margin capacity would overflow -- and did, on every input with fewer than ~1e7 groups,
    because partition sizes carry the variance of the GROUP sizes)"""
    Column, ops = gx
    rng = np.random.default_rng(5)
    n = 1_200_000
    k0 = rng.integers(0, 300, n).astype(np.int64)
    k1 = rng.integers(0, 300, n).astype(np.int64)
    hot = rng.random(n) < 0.5
    k0[hot], k1[hot] = 7, 11
    vals = rng.random(n)
    _check(Column, ops, [k0, k1], [300, 300], [0, 0], vals, expect_wide=True)


def test_more_groups_than_the_lds_tables_hold_device_asks_for_the_fallback(gx):
    """a caller's group bound that is far too low decides how many workgroups share a partition: ONE per partition here, and
    ~4600 groups per partition do not fit a 2816-slot table (4 key columns) -> the device reports -2 at the first attempt and
    ops.groupby_sum_count_tables takes the round-2 path over row hashes"""
    Column, ops = gx
    rng = np.random.default_rng(7)
    n = 3_000_000
    k0 = rng.integers(0, 3000, n).astype(np.int64)
    k1 = rng.integers(0, 2000, n).astype(np.int64)
    z = np.zeros(n, np.int64)
    vals = rng.random(n)
    kc = [Column.from_numpy(k) for k in (k0, k1, z, z)]
    assert ops._groupby_sum_count_wide(kc, Column.from_numpy(vals), max_groups_hint=1024) is None
    gk, s, cv, ca = ops.groupby_sum_count_tables(kc, Column.from_numpy(vals))
    assert s.size == len(np.unique(k0 * 2000 + k1))
    assert int(cv.to_numpy().astype(np.int64).sum()) == n


def test_nullable_or_small_inputs_take_the_round_2_path(gx):
    Column, ops = gx
    rng = np.random.default_rng(6)
    n = 400_000
    k0 = rng.integers(0, 30, n).astype(np.int64)
    k1 = rng.integers(0, 30, n).astype(np.int64)
    vals = rng.random(n)
    mask = rng.random(n) > 0.1
    assert ops._groupby_sum_count_wide([Column.from_numpy(k0, mask), Column.from_numpy(k1)], Column.from_numpy(vals)) is None
    assert ops._groupby_sum_count_wide([Column.from_numpy(k0[:1000]), Column.from_numpy(k1[:1000])], Column.from_numpy(vals[:1000])) is None


def test_c_abi_argument_checks(gx):
    Column, ops = gx
    from cudf_amd import _lib as L
    nb = ctypes.c_size_t(0)
    kp = (ctypes.c_void_p * 2)(0, 0)
    assert L.lib.gx_groupby_sum_count_wide(1, kp, L.FLOAT64, None, 10, 10, kp, None, None, None, None, ctypes.byref(nb), None) == -1  # nkeys < 2
    assert L.lib.gx_groupby_sum_count_wide(5, kp, L.FLOAT64, None, 10, 10, kp, None, None, None, None, ctypes.byref(nb), None) == -1  # nkeys > 4
    assert L.lib.gx_groupby_sum_count_wide(2, kp, L.INT8, None, 10, 10, kp, None, None, None, None, ctypes.byref(nb), None) == -2     # value dtype
    assert L.lib.gx_groupby_sum_count_wide(2, None, L.FLOAT64, None, 1 << 20, 1 << 20, None, None, None, None, None, ctypes.byref(nb), None) == 0
    assert nb.value > 0
