"""k_local_place (gx_sort.hip): the counting placement + per-thread window networks that sort the 8192-key cells of the
hybrid radix sort (cudf::sort / sorted_order of one 64-bit integer column, n >= 2^22; replaces the cub::DeviceRadixSort
passes of cpp/src/sort/sort_radix.cu:66-117 and sorted_order_radix.cu:83-94).  A cell whose 13-bit counting pass finds a
bin with more than 9 keys is left to k_local_sort's sub-bucket path; the device decides per cell.  Every case is compared
bit for bit with the oracle, and the test pins WHICH kernel sorted the cells (gx_sort_place_info = cells left to
k_local_sort), so that a silent fall-back cannot hide a broken placement behind a correct result and vice versa.
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import c_oracle
from oracle import cudf_oracle as orc


@pytest.fixture(scope="module")
def gx():
    import torch
    assert torch.cuda.is_available()
    import cudf_amd  # noqa: F401
    from cudf_amd import Column, ops, _lib as L
    yield Column, ops, L
    L.lib.gx_sort_set_experiment(0)
    L.lib.gx_sort_set_order_words(0)


def _sort(gx, v, descending=False):
    """gx_sort_keys through the C ABI -> (sorted array, info8, cells left to k_local_sort)"""
    Column, ops, L = gx
    col = Column.from_numpy(v)
    out = Column.empty(v.dtype, v.size)
    tmp = ops._run(L.lib.gx_sort_keys, col.gx, col.data_ptr, out.data_ptr, col.size, int(descending))
    ops._check_sort_status(tmp)
    info = (ctypes.c_int32 * 8)()
    L.check(L.lib.gx_sort_info(ops.ptr(tmp), info, ops.stream_ptr()), "gx_sort_info")
    todo = ctypes.c_int32(-1)
    L.check(L.lib.gx_sort_place_info(ops.ptr(tmp), ctypes.byref(todo), ops.stream_ptr()), "gx_sort_place_info")
    return out.to_numpy(), list(info), todo.value


def _order(gx, v, descending=False):
    Column, ops, L = gx
    return ops.sorted_order(Column.from_numpy(v), ascending=not descending).to_numpy()


def _cells(info):
    return 256 << info[4]


@pytest.mark.parametrize("dtype", ["int64", "uint64"])
@pytest.mark.parametrize("n", [(1 << 22) + 777, 20_000_003, 40_000_000])
def test_uniform_keys_are_placed(gx, dtype, n):
    """uniform keys: the cells go through the placement (a bin of 10 keys has probability 1e-9 .. 6e-8 depending on how full
    the cells are, so at most a stray cell may be left to the sub-bucket path), both directions; n covers the look-back path
    (< 2^25) and the cursor path"""
    rng = np.random.default_rng(n % 1000)
    info_t = np.iinfo(dtype)
    v = rng.integers(info_t.min, info_t.max, n, dtype=dtype, endpoint=True)
    for desc in (False, True):
        got, info, todo = _sort(gx, v, desc)
        exp = np.sort(v)[::-1] if desc else np.sort(v)
        assert got.tobytes() == exp.tobytes(), (dtype, n, desc, info)
        assert info[1] == 1, f"the hybrid path must have sorted the column: {info}"
        assert 0 <= todo <= 2, f"{todo} of {_cells(info)} cells were left to the sub-bucket path"


def test_crowded_cells_go_to_the_sub_bucket_path(gx):
    """a few keys repeated 40 times each: exactly their cells are crowded -> k_local_sort sorts those, k_local_place the rest"""
    rng = np.random.default_rng(3)
    n = 30_000_000
    v = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64)
    hot = rng.choice(n, 25, replace=False)
    for h in hot:
        v[rng.choice(n, 40, replace=False)] = v[h]
    got, info, todo = _sort(gx, v)
    assert got.tobytes() == c_oracle.sort_i64(v).tobytes()
    assert info[1] == 1 and 1 <= todo <= 25 + 2, (info, todo)
    # every cell crowded: each key 20 times (cells of ~4900 keys +- 300: they still fit their slots)
    n = 20_000_000
    runs = rng.integers(-2**63, 2**63 - 1, n // 20, dtype=np.int64)
    v = np.repeat(runs, 20)
    rng.shuffle(v)
    got, info, todo = _sort(gx, v, True)
    assert got.tobytes() == c_oracle.sort_i64(v, descending=True).tobytes()
    assert info[1] == 1 and todo > _cells(info) // 2, (info, todo)


def test_knob_32_takes_the_sub_bucket_path_and_agrees(gx):
    Column, ops, L = gx
    rng = np.random.default_rng(4)
    v = rng.integers(0, 2**64 - 1, 9_000_000, dtype=np.uint64)
    L.lib.gx_sort_set_experiment(32)
    try:
        a, info_a, todo_a = _sort(gx, v)
    finally:
        L.lib.gx_sort_set_experiment(0)
    b, info_b, todo_b = _sort(gx, v)
    assert a.tobytes() == b.tobytes() == np.sort(v).tobytes()
    assert info_a[1] == 1 and info_b[1] == 1 and todo_a == 0 and todo_b == 0


@pytest.mark.parametrize("kind", ["unique", "ties", "heavy", "fewbits"])
def test_sorted_order_through_the_placement(gx, kind):
    """pairs: the packed (low key bits, position) words are distinct, so ties of the KEY do not crowd a bin unless ten of
    them meet in one 13-bit bin of the word; tie order (stable) is checked against the oracle, both directions"""
    rng = np.random.default_rng(11)
    n = 12_000_000
    if kind == "unique":
        v = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64)
    elif kind == "ties":  # every key about four times
        v = rng.integers(-2**63, 2**63 - 1, n // 4, dtype=np.int64)[rng.integers(0, n // 4, n)]
    elif kind == "heavy":  # 64 copies of each key: crowded bins -> sub-bucket path for most cells
        v = rng.integers(-2**63, 2**63 - 1, n // 64, dtype=np.int64)[rng.integers(0, n // 64, n)]
    else:  # 20 key bits over 1.2e7 rows: few bits left below the level-1 digit, the word's digit reaches into the position
        v = rng.integers(0, 1 << 20, n, dtype=np.int64) << 30
    for desc in (False, True):
        got = _order(gx, v, desc)
        np.testing.assert_array_equal(got, orc.sorted_order(v, None, not desc), err_msg=f"{kind} desc={desc}")


@pytest.mark.parametrize("log2_cells", [12, 13])
def test_placement_at_the_cell_capacity(gx, log2_cells):
    """n just below the limit of a level-1 size class: cells 95 % full on average, some above 8000 keys (2^12 cells: the
    look-back path, 2^13: the cursor path); a handful of crowded cells is expected (6e-8 per bin)"""
    rng = np.random.default_rng(8)
    n = int(0.95 * 8192 * (1 << log2_cells))
    v = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64)
    got, info, todo = _sort(gx, v)
    assert got.tobytes() == c_oracle.sort_i64(v).tobytes()
    assert info[1] == 1 and _cells(info) == 1 << log2_cells and 0 <= todo < 32 and info[6] > 7900, (info, todo)


@pytest.mark.parametrize("n", [6_000_000, 20_000_000])
def test_float64_keys_through_the_placement(gx, n):
    """float keys travel as packed (sortable key bits, position) words in 16384-key cells: -0.0 / +0.0 ties and equal NaN
    payload classes must keep input order, the original bits come back through the position (pairs_write_out)"""
    rng = np.random.default_rng(n % 97)
    v = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64).view(np.float64).copy()
    v[::100_000] = 0.0   # a few hundred zeros of either sign; the random bit patterns bring n / 2048 NaNs (one cell)
    v[1::100_000] = -0.0
    for desc in (False, True):
        got, info, todo = _sort(gx, v, desc)
        assert got.tobytes() == orc.sort_keys(v, not desc).tobytes(), (n, desc, info)
        assert info[1] == 1, info
        # the zeros and the NaNs crowd their cells, the rest is placed
        assert 0 <= todo <= 8, f"{todo} of {_cells(info)} cells were left to the sub-bucket path"
    got = _order(gx, v)
    np.testing.assert_array_equal(got, orc.sorted_order(v, None, True))


# ------------------------------------------------------------------------------------------------
# 32-bit integer keys on the cursor path (n >= 2^25): two atomic-cursor partition levels + k_local_place on 32-bit words;
# crowded cells (and every cell when fewer than 13 key bits are left) go to k_local_sort on widened words; key ranges too narrow
# for two partition levels, rejected samples and overflowing cells to the LSD passes -- all decided on the device
# ------------------------------------------------------------------------------------------------
def _sort32(gx, v, descending=False):
    Column, ops, L = gx
    col = Column.from_numpy(v)
    out = Column.empty(v.dtype, v.size)
    tmp = ops._run(L.lib.gx_sort_keys, col.gx, col.data_ptr, out.data_ptr, col.size, int(descending))
    ops._check_sort_status(tmp)
    info = (ctypes.c_int32 * 8)()
    L.check(L.lib.gx_sort_info(ops.ptr(tmp), info, ops.stream_ptr()), "gx_sort_info")
    st = ctypes.c_int32(-1)
    L.check(L.lib.gx_sort_cursor_state(ops.ptr(tmp), ctypes.byref(st), ops.stream_ptr()), "gx_sort_cursor_state")
    todo = ctypes.c_int32(-1)
    L.check(L.lib.gx_sort_place_info(ops.ptr(tmp), ctypes.byref(todo), ops.stream_ptr()), "gx_sort_place_info")
    return out.to_numpy(), list(info), st.value, todo.value


@pytest.mark.parametrize("dtype", ["int32", "uint32"])
@pytest.mark.parametrize("n", [40_000_000, 70_000_000])
def test_32bit_keys_uniform_take_the_cursor_path(gx, dtype, n):
    rng = np.random.default_rng(n % 1013)
    ii = np.iinfo(dtype)
    v = rng.integers(ii.min, ii.max, n, dtype=dtype, endpoint=True)
    for desc in (False, True):
        got, info, state, todo = _sort32(gx, v, desc)
        assert got.tobytes() == c_oracle.sort_32(v, descending=desc).tobytes(), (dtype, n, desc, info, state)
        assert state == 3 and info[1] == 1, f"uniform 32-bit keys must be sorted by the cursor path: state {state}, {info}"
        assert 0 <= todo <= 3


@pytest.mark.parametrize("kind", ["duplicates", "narrow", "very_narrow", "sorted", "skewed", "outlier", "multiples"])
def test_32bit_keys_fallbacks_are_exact(gx, kind):
    """inputs on which the 32-bit cursor path must take one of its fallbacks (or survive): bit-exact either way, and the
    device's decision is pinned where it is deterministic"""
    rng = np.random.default_rng(hash_seed32(kind))
    n = 36_000_000
    if kind == "duplicates":  # 200000 distinct values over the whole range, 180 copies each: every cell crowded -> k_local_sort
        v = rng.integers(-2**31, 2**31 - 1, 200_000, dtype=np.int32)[rng.integers(0, 200_000, n)]
    elif kind == "narrow":  # 22 varying bits: 9 left below level 1, fewer than the counting pass takes -> k_local_sort for every cell
        v = rng.integers(0, 1 << 22, n, dtype=np.int32)
    elif kind == "very_narrow":  # 16 varying bits: nothing for two partition levels to do -> state 4, LSD without an up-front read
        v = rng.integers(0, 1 << 16, n, dtype=np.int32)
    elif kind == "sorted":
        v = np.sort(rng.integers(-2**31, 2**31 - 1, n, dtype=np.int32))
    elif kind == "skewed":  # 80 % of the keys in one level-0 bin
        v = rng.integers(-2**31, 2**31 - 1, n, dtype=np.int32)
        hot = rng.random(n) < 0.8
        v[hot] = (v[hot] & 0x00FFFFFF) | (37 << 24)
    elif kind == "outlier":  # one unsampled row above the sampled range: the verdict of level 0 rejects the plan
        v = rng.integers(0, 1 << 30, n, dtype=np.int32)
        v[12345] = np.int32(2**31 - 1)
    else:  # multiples of 2^20: nothing varies below level 1 beyond the cell digit
        v = (rng.integers(0, 1 << 11, n, dtype=np.int32) << 20).astype(np.int32)
    got, info, state, todo = _sort32(gx, v)
    assert got.tobytes() == np.sort(v).tobytes(), (kind, info, state, todo)
    if kind == "duplicates":
        assert state == 3 and info[1] == 1 and todo > _cells(info) // 2, (info, state, todo)
    if kind == "narrow":
        assert state == 3 and info[1] == 1 and todo == 0, (info, state, todo)
    if kind == "very_narrow":
        assert state == 4 and info[1] == 0, (info, state)
    if kind == "sorted":
        assert state == 3 and info[1] == 1, (info, state)
    got, info, state, todo = _sort32(gx, v, True)
    assert got.tobytes() == np.sort(v)[::-1].tobytes(), (kind, "descending", info, state, todo)


def hash_seed32(s):
    import zlib
    return zlib.crc32(s.encode()) + 32


@pytest.mark.parametrize("dtype,kind", [("int32", "wide"), ("uint32", "ties"), ("int32", "few")])
def test_sorted_order_of_32bit_keys_as_a_word_sort(gx, dtype, kind):
    """n >= 2^25, no nulls: (sortable key << 32) | row goes through the unstable 64-bit keys-only sort -- the word order IS the
    stable order of the pairs; checked against the oracle's stable order in both directions"""
    Column, ops, L = gx
    L.lib.gx_sort_set_order_words(1)  # an experiment, off by default (gx_knobs.h)
    rng = np.random.default_rng(hash_seed32(kind + dtype))
    n = 34_000_000
    ii = np.iinfo(dtype)
    if kind == "wide":
        v = rng.integers(ii.min, ii.max, n, dtype=dtype, endpoint=True)
    elif kind == "ties":  # ~35 rows per key
        v = rng.integers(ii.min, ii.max, 1_000_000, dtype=dtype, endpoint=True)[rng.integers(0, 1_000_000, n)]
    else:  # 1000 distinct keys in a narrow range: long runs of equal keys inside every cell
        v = rng.integers(100, 1100, n).astype(dtype)
    try:
        for desc in (False, True):
            got = _order(gx, v, desc)
            np.testing.assert_array_equal(got, orc.sorted_order(v, None, not desc), err_msg=f"{dtype} {kind} desc={desc}")
    finally:
        L.lib.gx_sort_set_order_words(0)


def _order_info(gx, v, descending=False):
    """gx_sorted_order through the C ABI -> (permutation, info8, cells left to k_local_sort)"""
    Column, ops, L = gx
    col = Column.from_numpy(v)
    out = Column.empty(np.int32, v.size)
    tmp = ops._run(L.lib.gx_sorted_order, col.gx, col.data_ptr, None, col.size, 0, int(descending), 1, out.data_ptr)
    ops._check_sort_status(tmp)
    info = (ctypes.c_int32 * 8)()
    L.check(L.lib.gx_sort_info(ops.ptr(tmp), info, ops.stream_ptr()), "gx_sort_info")
    todo = ctypes.c_int32(-1)
    L.check(L.lib.gx_sort_place_info(ops.ptr(tmp), ctypes.byref(todo), ops.stream_ptr()), "gx_sort_place_info")
    return out.to_numpy(), list(info), todo.value


@pytest.mark.parametrize("dtype", ["int64", "int32"])
def test_fuller_buckets_take_one_more_level1_bit(gx, dtype):
    """n at 95 % of a size class and keys in a range that is not a power of two (the top digit uses 160 of 256 bins): buckets are
    1.6x fuller than n / 256, the cells would overflow -- the device sees it in the exact level-0 histogram and takes one more
    level-1 bit instead of handing the column to the LSD passes"""
    rng = np.random.default_rng(21)
    n = int(0.95 * 8192 * (1 << 13))  # cursor path, 5 level-1 bits from n alone
    hi = int(0.625 * 2**40) if dtype == "int64" else int(0.625 * 2**31)
    v = rng.integers(0, hi, n, dtype=dtype)
    got, info, todo = _sort(gx, v)
    assert got.tobytes() == np.sort(v).tobytes()
    assert info[1] == 1 and info[4] == 6 and 0 < info[6] <= 8192, f"expected the hybrid path with 6 level-1 bits: {info}"


def _order_info(gx, v, descending=False):
    """gx_sorted_order through the C ABI -> (permutation, info8)"""
    Column, ops, L = gx
    col = Column.from_numpy(v)
    out = Column.empty(np.int32, v.size)
    tmp = ops._run(L.lib.gx_sorted_order, col.gx, col.data_ptr, None, col.size, 0, int(descending), 1, out.data_ptr)
    ops._check_sort_status(tmp)
    info = (ctypes.c_int32 * 8)()
    L.check(L.lib.gx_sort_info(ops.ptr(tmp), info, ops.stream_ptr()), "gx_sort_info")
    return out.to_numpy(), list(info)


@pytest.mark.parametrize("desc", [False, True])
def test_sorted_order_of_a_key_range_that_is_not_a_power_of_two_takes_wider_cells(gx, desc):
    """Round 4 (DESIGN's measured 91 ms cliff at 1e9 rows): sorted_order of keys in [0, 1e12) at the top of a size
    class.  The top digit uses 233 of 256 bins, so the buckets are 1.1x fuller than n / 256 and 8192-key cells overflow; the
    look-back path of the pairs has no extra level-1 bit to take, but k_hy_plan's stage 1 sees the exact level-0 histogram and
    switches the sort to 16384-key cells (both cell-sort instantiations are enqueued, one is a no-op).  Stable order bit-exact
    against the oracle; the hybrid path must have produced it (no LSD pass), with cells above 8192 keys.  Full-range keys of the
    same size from a power-of-two range keep the 8192-key cells."""
    rng = np.random.default_rng(21)
    n = 7_800_000                                            # n / 2^10 = 7617: 97 % of the 8192-key size class
    v = rng.integers(0, 1_000_000_000_000, n, dtype=np.int64)
    got, info = _order_info(gx, v, desc)
    np.testing.assert_array_equal(got, orc.sorted_order(v, None, not desc))
    assert info[1] == 1 and info[7] == -1                    # hybrid ok, every LSD pass skipped
    assert 8192 < info[6] <= 16384                           # largest cell: only the wider cells hold it
    w = rng.integers(0, 1 << 40, n, dtype=np.int64)            # a power-of-two range of the same width: every bin used, cells fit
    got, info = _order_info(gx, w, desc)
    np.testing.assert_array_equal(got, orc.sorted_order(w, None, not desc))
    assert info[1] == 1 and info[6] <= 8192
