"""GPU parity: radix sort / sorted_order / gather through the C ABI vs the CPU oracle."""
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


def _rand(dtype, n, rng, lowcard=False):
    dt = np.dtype(dtype)
    if dt.kind == "f":
        v = rng.standard_normal(n).astype(dt)
        if n > 16:
            v[rng.integers(0, n, n // 50 + 1)] = np.nan
            v[rng.integers(0, n, n // 50 + 1)] = -np.nan
            v[rng.integers(0, n, n // 50 + 1)] = 0.0
            v[rng.integers(0, n, n // 50 + 1)] = -0.0
            v[rng.integers(0, n, n // 100 + 1)] = np.inf
            v[rng.integers(0, n, n // 100 + 1)] = -np.inf
        return v
    info = np.iinfo(dt)
    if lowcard:
        return rng.integers(100, 10000, n).astype(dt) if dt.itemsize > 1 else rng.integers(0, 7, n).astype(dt)
    return rng.integers(info.min, info.max, n, dtype=dt, endpoint=True)


DTYPES = ["int8", "uint8", "int16", "uint16", "int32", "uint32", "int64", "uint64", "float32", "float64"]
SIZES = [0, 1, 2, 63, 64, 65, 1000, 8191, 8192, 8193, 100_003, 1_000_003]


@pytest.mark.parametrize("algo", [0, 1, 2])
@pytest.mark.parametrize("dtype", DTYPES)
def test_sort_keys_matches_oracle(gx, dtype, algo):
    Column, ops = gx
    from cudf_amd import _lib
    _lib.lib.gx_sort_set_algorithm(algo)
    try:
        rng = np.random.default_rng(42)
        for n in SIZES:
            for lowcard in (False, True):
                v = _rand(dtype, n, rng, lowcard)
                for asc in (True, False):
                    got = ops.sort(Column.from_numpy(v), ascending=asc).to_numpy()
                    exp = orc.sort_keys(v, asc)
                    assert got.tobytes() == exp.tobytes(), (dtype, n, lowcard, asc, algo)
    finally:
        _lib.lib.gx_sort_set_algorithm(0)


@pytest.mark.parametrize("algo", [0, 1, 2])
@pytest.mark.parametrize("dtype", ["int32", "int64", "uint64", "float32", "float64", "int8", "uint16"])
def test_sorted_order_matches_oracle(gx, dtype, algo):
    Column, ops = gx
    from cudf_amd import _lib
    _lib.lib.gx_sort_set_algorithm(algo)
    try:
        rng = np.random.default_rng(7)
        for n in [0, 1, 65, 5121, 10_240, 200_001]:
            v = _rand(dtype, n, rng, lowcard=True)  # many ties: stability is observable
            for asc in (True, False):
                got = ops.sorted_order(Column.from_numpy(v), ascending=asc).to_numpy()
                np.testing.assert_array_equal(got, orc.sorted_order(v, None, asc), err_msg=f"{dtype} {n} {asc}")
    finally:
        _lib.lib.gx_sort_set_algorithm(0)


@pytest.mark.parametrize("dtype", ["int32", "int64", "float64", "uint8"])
def test_sorted_order_with_nulls(gx, dtype):
    Column, ops = gx
    rng = np.random.default_rng(9)
    for n in [1, 10, 1000, 70_001]:
        v = _rand(dtype, n, rng, lowcard=True)
        valid = rng.random(n) > 0.3
        if n == 10:
            valid[:] = False  # all null
        for asc in (True, False):
            for before in (True, False):
                got = ops.sorted_order(Column.from_numpy(v, valid), asc, before).to_numpy()
                exp = orc.sorted_order(v, valid, asc, before)
                np.testing.assert_array_equal(got, exp, err_msg=f"{dtype} {n} asc={asc} before={before}")


@pytest.mark.parametrize("case", gv.SORT, ids=lambda c: c["name"])
def test_reference_golden_sort(gx, case):
    Column, ops = gx
    vals, mask = gv.col(case["values"], case["dtype"], case["valid"])
    got = ops.sorted_order(Column.from_numpy(vals, mask), case["ascending"], case["null_before"]).to_numpy()
    exp = np.array(case["expected"], np.int32)
    if case["compare"] == "indices":
        np.testing.assert_array_equal(got, exp)
    else:
        m = np.ones(len(vals), bool) if mask is None else mask
        np.testing.assert_array_equal(m[got], m[exp])
        np.testing.assert_array_equal(vals[got][m[got]], vals[exp][m[exp]])
    if mask is None:
        assert ops.sort(Column.from_numpy(vals), case["ascending"]).to_numpy().tobytes() == vals[exp].tobytes()


def test_sort_by_key_and_gather(gx):
    Column, ops = gx
    rng = np.random.default_rng(3)
    n = 50_000
    keys = rng.integers(0, 500, n).astype(np.int64)
    pay = rng.standard_normal(n)
    pvalid = rng.random(n) > 0.1
    out = ops.sort_by_key([Column.from_numpy(pay, pvalid), Column.from_numpy(keys)], Column.from_numpy(keys))
    order = orc.sorted_order(keys)
    np.testing.assert_array_equal(out[1].to_numpy(), keys[order])
    np.testing.assert_array_equal(out[0].valid_numpy(), pvalid[order])
    got = out[0].to_numpy()
    np.testing.assert_array_equal(got[pvalid[order]], pay[order][pvalid[order]])
    assert out[0].null_count == int((~pvalid).sum())
    # JoinNoMatch entries become nulls under the NULLIFY policy
    gm = np.array([3, -2**31, 0, n - 1, -2**31], np.int32)
    g = ops.gather(Column.from_numpy(keys), Column.from_numpy(gm), nullify_out_of_bounds=True)
    np.testing.assert_array_equal(g.valid_numpy(), [True, False, True, True, False])
    np.testing.assert_array_equal(g.to_numpy()[[0, 2, 3]], keys[[3, 0, n - 1]])
    with pytest.raises(RuntimeError, match="Mismatch in number of rows"):
        ops.sort_by_key([Column.from_numpy(pay[:10])], Column.from_numpy(keys))


def test_sort_large_properties(gx):
    """Size-independent properties at a size the oracle does not run at: sortedness, permutation
    checksum (sum and xor of splitmix64(element)), idempotence."""
    Column, ops = gx
    n = 50_000_000
    col = ops.random_column(np.int64, n, seed=42)
    s_in = ops.checksum(col)
    out = ops.sort(col)
    s_out = ops.checksum(out)
    assert s_out[2] == 0
    assert s_in[:2] == s_out[:2]
    again = ops.sort(out)
    assert ops.checksum(again) == s_out
    desc = ops.sort(col, ascending=False)
    cd = ops.checksum(desc, descending=True)
    assert cd[2] == 0 and cd[:2] == s_in[:2]
    # argsort: gather(col, order) == sorted
    order = ops.sorted_order(col)
    g = ops.gather(col, order)
    assert ops.checksum(g) == s_out
    import torch
    o = order.data[: n * 4].view(torch.int32).to(torch.int64)
    assert int(torch.bincount(o[: 1_000_000] % 1024).sum()) == 1_000_000
    assert int(o.sum().item()) == n * (n - 1) // 2  # a permutation of iota


def _sort_info(ops_mod, col, ascending=True):
    """Run gx_sort_keys directly so the scratch (and the device-side plan in it) can be inspected."""
    import ctypes
    from cudf_amd import _lib as L
    from cudf_amd.column import Column, device_bytes, ptr, stream_ptr
    out = Column.empty(col.dtype, col.size)
    nb = ctypes.c_size_t(0)
    args = (col.gx, col.data_ptr, out.data_ptr, col.size, int(not ascending))
    L.check(L.lib.gx_sort_keys(*args, None, ctypes.byref(nb), stream_ptr()), "query")
    tmp = device_bytes(nb.value)
    L.check(L.lib.gx_sort_keys(*args, ptr(tmp), ctypes.byref(nb), stream_ptr()), "sort")
    info = (ctypes.c_int32 * 8)()
    L.check(L.lib.gx_sort_info(ptr(tmp), info, stream_ptr()), "info")
    st = ctypes.c_int(0)
    L.check(L.lib.gx_sort_status(ptr(tmp), ctypes.byref(st), stream_ptr()), "status")
    assert st.value == 0
    return out.to_numpy(), list(info)


@pytest.mark.parametrize("dtype", ["int64", "uint64", "float64"])
def test_hybrid_msd_sort_matches_oracle(gx, dtype):
    """64-bit keys-only sorts of >= 2^22 rows take the hybrid MSD path (two partition passes + LDS
    local sort); skewed inputs must fall back to the LSD passes on the device.  Bit-exact either way."""
    Column, ops = gx
    rng = np.random.default_rng(77)
    n = (1 << 22) + 12_345

    def spread(m):
        # keys whose leading bits are uniform (for doubles: random bit patterns, incl. NaNs/Infs/-0.0)
        if dtype == "float64":
            return rng.integers(-2**63, 2**63 - 1, m, dtype=np.int64).view(np.float64)
        return _rand(dtype, m, rng, False)

    for asc in (True, False):
        v = spread(n)
        got, info = _sort_info(ops, Column.from_numpy(v), asc)
        assert got.tobytes() == orc.sort_keys(v, asc).tobytes(), (dtype, asc, info)
        assert info[0] == 1 and info[1] == 1, f"uniform keys must use the hybrid path: {info}"
        assert 0 < info[6] <= 16384
    # larger: 9+ bits at level 1, several tiles per bucket
    n = 20_000_000
    v = spread(n)
    got, info = _sort_info(ops, Column.from_numpy(v))
    assert got.tobytes() == orc.sort_keys(v, True).tobytes()
    assert info[1] == 1
    # skew: a third of the keys in one narrow cluster -> their level-0 bucket holds more keys than all of its cells -> device-side
    # fallback.  Since round 4 the plan declines as soon as the exact level-0 histogram shows it (attempted = 0) instead of
    # spending both partition passes on a cell that must overflow
    v = spread(6_000_000)
    if dtype == "float64":
        v[::3] = 1.0 + rng.random(len(v[::3])) * 1e-9
    else:
        v[::3] = (rng.integers(0, 1 << 20, len(v[::3])) + (1 << 40)).astype(dtype)
    got, info = _sort_info(ops, Column.from_numpy(v))
    assert got.tobytes() == orc.sort_keys(v, True).tobytes()
    assert info[0] == 0 and info[1] == 0 and info[7] > 0, f"expected the LSD fallback: {info}"
    # low-entropy keys (two active bytes): hybrid is not attempted
    v = _rand(dtype, 5_000_000, rng, True) if dtype != "float64" else np.floor(rng.random(5_000_000) * 50)
    got, info = _sort_info(ops, Column.from_numpy(v))
    assert got.tobytes() == orc.sort_keys(v, True).tobytes()


def test_hybrid_knob_off_uses_lsd(gx):
    Column, ops = gx
    from cudf_amd import _lib
    rng = np.random.default_rng(78)
    v = rng.integers(-2**63, 2**63 - 1, 5_000_000, dtype=np.int64)
    _lib.lib.gx_sort_set_hybrid(0)
    try:
        got, info = _sort_info(ops, Column.from_numpy(v))
    finally:
        _lib.lib.gx_sort_set_hybrid(1)
    assert got.tobytes() == np.sort(v).tobytes()
    assert info[0] == 0 and info[7] == 8


@pytest.mark.parametrize("dtype,lo,hi,passes", [("int64", -1000, 1000, 3), ("int64", -(1 << 40), 1 << 40, 6), ("int64", -1, 1, 1),
                                                 ("int32", -100, 100, 2), ("int32", -70_000, 70_000, 4), ("int16", -100, 100, 2),
                                                 ("int64", 0, 1000, 2), ("int64", -3000, -1000, 2)])
def test_sign_extension_bytes_are_not_sorted_on(gx, dtype, lo, hi, passes):
    """Signed keys spread around zero: the bytes between the highest data bit and the top byte are copies of the sign (0x00 / 0xFF)
    -- two values, so not a constant digit, yet the top byte already separates the signs.  k_hist_all reduces OR(key ^ sign
    extension), k_plan skips those passes (round 4): [-1000, 1000) takes bytes 0, 1 and 7.  Bit-exact in both directions, and the
    number of LSD passes is pinned; ranges on one side of zero keep taking their constant-byte skips."""
    Column, ops = gx
    rng = np.random.default_rng(79)
    v = rng.integers(lo, hi, 3_000_001).astype(dtype)
    v[5], v[77] = lo, hi - 1
    for asc in (True, False):
        got, info = _sort_info(ops, Column.from_numpy(v), asc)
        assert got.tobytes() == orc.sort_keys(v, asc).tobytes(), (dtype, lo, hi, asc, info)
        assert info[7] == passes, info
    # the permutation (pairs passes) follows the same plan; ties keep their input order
    order = ops.sorted_order(Column.from_numpy(v)).to_numpy()
    np.testing.assert_array_equal(order, np.argsort(v, kind="stable").astype(np.int32))


def test_look_back_path_takes_the_sign_fold(gx):
    """6e6 int64 keys uniform in [-1e12, 1e12) (below the cursor path's 2^25 rows: the look-back path plans from the exact masks
    of k_hy_hist): sign fold at height 40 -> shift0 33, every cell sorted in LDS; the permutation (pairs) takes the same plan"""
    Column, ops = gx
    rng = np.random.default_rng(80)
    v = rng.integers(-10**12, 10**12, 6_000_000, dtype=np.int64)
    v[::100_000] = v[7]                                  # ties: the order must be stable
    for asc in (True, False):
        got, info = _sort_info(ops, Column.from_numpy(v), asc)
        assert got.tobytes() == orc.sort_keys(v, asc).tobytes()
        assert info[0] == 1 and info[1] == 1 and info[2] == 33, info
    order = ops.sorted_order(Column.from_numpy(v)).to_numpy()
    np.testing.assert_array_equal(order, np.argsort(v, kind="stable").astype(np.int32))
    order = ops.sorted_order(Column.from_numpy(v), ascending=False).to_numpy()
    np.testing.assert_array_equal(order, orc.sorted_order(v, None, False).astype(np.int32))


@pytest.mark.parametrize("dtype", ["int64", "float64", "uint64"])
def test_hybrid_sorted_order_is_stable(gx, dtype):
    """sorted_order (key + iota payload) of >= 2^22 rows takes the hybrid path with the packed
    (low key bits, position) local sort: ties must come out in input order, both directions."""
    Column, ops = gx
    rng = np.random.default_rng(91)
    n = (1 << 22) + 4321
    if dtype == "float64":
        v = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64).view(np.float64)
        v[::1000] = 0.0
        v[1::1000] = -0.0
    else:
        v = _rand(dtype, n, rng, False)
    v[::7] = v[3]                       # a heavy duplicate: ties across many tiles of one cell
    v[5::11] = v[5]
    for asc in (True, False):
        got = ops.sorted_order(Column.from_numpy(v), ascending=asc).to_numpy()
        np.testing.assert_array_equal(got, orc.sorted_order(v, None, asc), err_msg=f"{dtype} asc={asc}")
    # larger, no heavy duplicates: 8 level-1 bits, full cells
    n = 20_000_000
    v = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64).view(np.dtype(dtype)) if dtype != "uint64" else _rand(dtype, n, rng, False)
    got = ops.sorted_order(Column.from_numpy(v)).to_numpy()
    np.testing.assert_array_equal(got, orc.sorted_order(v, None, True))


def test_hybrid_sort_presorted_and_few_values(gx):
    """Inputs whose waves hit a single bin (already sorted, reverse sorted, a handful of distinct
    values spread over the top bits) exercise the wave-uniform ranking path."""
    Column, ops = gx
    rng = np.random.default_rng(5)
    n = 6_000_000
    base = np.sort(rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64))
    for v in (base, base[::-1].copy()):
        got, info = _sort_info(ops, Column.from_numpy(v))
        assert got.tobytes() == base.tobytes() and info[1] == 1, info
        order = ops.sorted_order(Column.from_numpy(v)).to_numpy()
        np.testing.assert_array_equal(order, np.argsort(v, kind="stable"))
