"""GPU: cudf::sorted_order / stable_sorted_order of a TABLE of numeric key columns in one word sort (round 6).

The reference sorts the row indices under the lexicographic row comparator (cpp/src/sort/sort_impl.cuh:61-93).  Rounds 1-5 ran LSD over
the columns: one argsort + a random 8-byte gather + a random 4-byte gather per extra column.  gx_sorted_order_table
(cudf_amd/csrc/gx_order.hip) nests the columns' ranks into one 64-bit word per row, sorts the words keys-only and puts runs of equal
ranks right by comparing whole tuples.  Every case: the permutation bit-exact against the NumPy oracle (np.lexsort on the comparator's
element order, stable), through the C ABI and through libcudf.so's cudf::sorted_order / stable_sorted_order."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cudf_oracle as orc
from tests.test_gpu_cpp_parity import Dev, Out, shim  # noqa: F401  (fixture)
from tests.test_gpu_sort_splitters import _keys

N = 40_000_003


@pytest.fixture(scope="module")
def gx():
    import torch
    assert torch.cuda.is_available()
    import cudf_amd  # noqa: F401
    from cudf_amd import Column, ops, _lib as L
    return Column, ops, L


def _order(gx, cols, asc):
    Column, ops, L = gx
    return ops.sorted_order_table([Column.from_numpy(c) for c in cols], asc).to_numpy()


def _check(gx, cols, asc):
    got = _order(gx, cols, asc)
    np.testing.assert_array_equal(got, orc.sorted_order_table(cols, asc))


def test_reference_vectors(gx):
    """cpp/tests/sort/sort_test.cpp:148-175 (Sort.WithAllValid) and stable_sort_tests.cpp:150-171 (StableSort.WithAllValid: a full tie
    keeps its input order), their numeric columns"""
    from tests.golden import reference_vectors as gv
    for case in gv.SORT_TABLE:
        for dt in (np.int8, np.int16, np.int32, np.int64, np.uint8, np.uint32, np.uint64, np.float32, np.float64):
            got = _order(gx, [np.array(c, dt) for c in case["cols"]], case["ascending"])
            assert got.tolist() == case["expected"], (case["name"], dt)


CASES = ["lowcard_x_random", "random_x_any", "tiny_domains", "bell_x_lognormal", "correlated", "equal_lead_sorted", "equal_lead_reversed",
         "zipf_x_zipf"]


def _case(kind, rng, n=N):
    if kind == "lowcard_x_random":     # 1000 leading values, a million-row tie group each at 1e9 rows: the second column does the work
        return [rng.integers(0, 1000, n, dtype=np.int64), rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64)]
    if kind == "random_x_any":         # no ties in the leading column
        return [rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64), rng.integers(0, 5, n, dtype=np.int64)]
    if kind == "tiny_domains":         # 1000 distinct tuples: whole tuples tie by the 40 000
        return [rng.integers(0, 10, n, dtype=np.int64), rng.integers(0, 100, n, dtype=np.int64)]
    if kind == "bell_x_lognormal":
        return [_keys("normal", rng, n) >> 24, _keys("lognormal", rng, n)]    # (>> 24: ~2^17 distinct leading values -> ties)
    if kind == "correlated":           # the second column's distribution GIVEN the first is narrow: the marginal rank resolves little
        a = rng.integers(0, 3000, n, dtype=np.int64)
        return [a, a * 1_000_000_007 + rng.integers(0, 50, n, dtype=np.int64)]
    if kind == "equal_lead_sorted":
        return [np.full(n, 7, np.int64), np.sort(rng.integers(-2**62, 2**62, n, dtype=np.int64))]
    if kind == "equal_lead_reversed":
        return [np.full(n, 7, np.int64), np.sort(rng.integers(-2**62, 2**62, n, dtype=np.int64))[::-1].copy()]
    if kind == "zipf_x_zipf":
        return [_keys("zipf", rng, n), _keys("zipf", rng, n)]
    raise AssertionError(kind)


@pytest.mark.parametrize("kind", CASES)
@pytest.mark.parametrize("asc", [(True, True), (False, True), (True, False)])
def test_two_int64_columns(gx, kind, asc):
    rng = np.random.default_rng(abs(hash(kind)) % 1000)
    _check(gx, _case(kind, rng), list(asc))


def test_mixed_types_floats_with_nan_and_zeros(gx):
    """int32 x float64 (NaN of both signs with payloads, -0.0 / +0.0, infinities) x int8: NaN greatest and equivalent in BOTH directions
    (row_operator/common_utils.cuh:157-169), so ties among NaN rows fall to the third column, then to the row"""
    rng = np.random.default_rng(17)
    a = rng.integers(-20, 20, N).astype(np.int32)
    f = rng.standard_normal(N)
    f[::5] = np.nan
    f[1::5].view(np.uint64)[:] = 0xFFF8000000000077
    f[2::25] = -0.0
    f[3::25] = 0.0
    f[4::125] = np.inf
    f[9::125] = -np.inf
    t = rng.integers(-128, 127, N).astype(np.int8)
    for asc in ([True, True, True], [True, False, True], [False, False, False]):
        _check(gx, [a, f, t], asc)


def test_narrow_columns_and_many_of_them(gx):
    rng = np.random.default_rng(23)
    n = 20_000_001
    cols = [rng.integers(0, 2, n).astype(bool), rng.integers(0, 4, n).astype(np.uint8), rng.integers(-3, 3, n).astype(np.int16),
            rng.standard_normal(n).astype(np.float32).round(1), rng.integers(0, 3, n).astype(np.uint16),
            rng.integers(-2, 2, n).astype(np.int32), rng.integers(0, 2**40, n).astype(np.uint64), rng.integers(0, 9, n).astype(np.uint32)]
    _check(gx, cols, [True, False, True, False, True, False, True, False])


@pytest.mark.parametrize("n", [1, 2, 3, 64, 65, 513, 4096, 70_001, 1_000_003])
def test_small_tables(gx, n):
    rng = np.random.default_rng(n)
    cols = [rng.integers(0, 7, n).astype(np.int32), (rng.integers(0, 50, n) * 0.5).astype(np.float64), rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64)]
    _check(gx, cols, [False, True, False])
    _check(gx, cols[:1], [True])
    _check(gx, [cols[2], cols[0]], [True, True])


def test_argument_errors(gx):
    Column, ops, L = gx
    c = Column.from_numpy(np.arange(10, dtype=np.int64))
    with pytest.raises(ValueError):
        ops.sorted_order_table([c, c], [True])
    with pytest.raises(ValueError):
        ops.sorted_order_table([c] * 9, True)
    assert ops.sorted_order_table([Column.from_numpy(np.empty(0, np.int64))] * 2, True).size == 0


def test_through_libcudf(shim):  # noqa: F811
    """cudf::sorted_order and cudf::stable_sorted_order of 2- and 3-column tables at 3e6 rows (above the C++ layer's table-path threshold)
    and at 1e5 rows (below it: the loop over the columns) agree with the oracle"""
    rng = np.random.default_rng(31)
    for n in (3_000_001, 100_003):
        a = rng.integers(0, 300, n).astype(np.int32)
        b = rng.integers(-2**40, 2**40, n, dtype=np.int64) >> rng.integers(0, 40, n)
        f = (rng.integers(-1000, 1000, n) * 0.125).astype(np.float64)
        for cols, desc in (([a, b], [0, 1]), ([a, f, b], [1, 0, 0]), ([f, a], [1, 1])):
            devs = [Dev(c) for c in cols]
            k = len(cols)
            dt = (ctypes.c_int * k)(*[d.tid for d in devs])
            dp = (ctypes.c_void_p * k)(*[d.p.value for d in devs])
            de = (ctypes.c_int * k)(*desc)
            for stable in (0, 1):
                out = Out(np.int32, n)
                shim("shim_table_sorted_order", k, dt, dp, None, None, n, de, None, stable, out.p)
                np.testing.assert_array_equal(out.get(n), orc.sorted_order_table(cols, [not d for d in desc]))
