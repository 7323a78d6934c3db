"""GPU: the COUNTING SORT of narrow key ranges (round 5; gx_sort.hip k_cs_count / k_cs_scan / k_cs_fill, FastPlan::state 5).

cudf::sort of an integer column whose varying bits are its low 15 or fewer -- the reference sort benchmark's own distribution,
keys uniform in [100, 10001) (cpp/benchmarks/sort/sort.cpp:24-26) -- is one histogram pass + one fill instead of LSD passes
(cub::DeviceRadixSort behind cpp/src/sort/sort_radix.cu:52-161 runs its digit passes whatever the values).  The plan comes from
the cursor path's SAMPLE; the count checks every key against it, and a key outside the sampled range must send the column to
the LSD passes on the device (state 4) -- never a wrong result.  Every case: bit-exact against the plain-C oracle, device state
pinned.
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import c_oracle

N = 34_000_003  # just above 2^25 (the cursor path's threshold), not a multiple of anything


@pytest.fixture(scope="module")
def gx():
    import torch
    assert torch.cuda.is_available()
    import cudf_amd  # noqa: F401
    from cudf_amd import Column, ops, _lib as L
    yield Column, ops, L
    L.lib.gx_sort_set_counting(1)


def _sort_with_state(gx, v, descending=False, offset=0):
    Column, ops, L = gx
    col = Column.from_numpy(v)
    n = v.size - offset
    out = Column.empty(v.dtype, n)
    tmp = ops._run(L.lib.gx_sort_keys, col.gx, ctypes.c_void_p(col.data_ptr.value + offset * v.dtype.itemsize), out.data_ptr, n, int(descending))
    ops._check_sort_status(tmp)
    st = ctypes.c_int32(-1)
    L.check(L.lib.gx_sort_cursor_state(ops.ptr(tmp), ctypes.byref(st), ops.stream_ptr()), "gx_sort_cursor_state")
    return out.to_numpy(), st.value


def _oracle(v, descending=False):
    if v.dtype == np.int64:
        return c_oracle.sort_i64(v, descending=descending)
    s = np.sort(v, kind="stable")
    return s[::-1].copy() if descending else s


@pytest.mark.parametrize("dtype,lo,hi", [("int64", 100, 10001), ("int64", 0, 32768), ("int64", 5, 7), ("int32", 100, 10001), ("uint32", 0, 20000),
                                          ("int64", -30000, -20000), ("uint64", (1 << 63) + 4096, (1 << 63) + 8192)])
@pytest.mark.parametrize("descending", [False, True])
def test_narrow_key_range_is_counted(gx, dtype, lo, hi, descending):
    rng = np.random.default_rng(hi & 0xFFFF)
    v = rng.integers(lo, hi, N, dtype=np.dtype(dtype))
    got, state = _sort_with_state(gx, v, descending)
    assert got.tobytes() == _oracle(v, descending).tobytes()
    assert state == 5, f"keys in [{lo}, {hi}): state {state}, expected the counting sort (5)"


def test_skewed_counts_and_a_sliced_column(gx):
    """one value holds 90 % of the rows (every lane of a wave adds to the same LDS counter), a few values occur once (tiles of the
    fill that cover many groups), and the column starts at an odd element (the count's 16-byte loads begin behind a head)"""
    rng = np.random.default_rng(11)
    v = rng.integers(0, 3000, N, dtype=np.int64)
    v[rng.random(N) < 0.9] = 1234
    v[1000:1000 + 3000] = np.arange(3000)         # every value present; some only here
    for off in (0, 1, 3):
        got, state = _sort_with_state(gx, v, False, offset=off)
        assert got.tobytes() == c_oracle.sort_i64(v[off:]).tobytes()
        assert state == 5


def test_an_unsampled_outlier_sends_the_column_to_the_lsd_passes(gx):
    """ONE key outside the range the sample saw (a sentinel among small ids), at a row no sampled chunk contains: the count's exact
    check raises cs_fail, k_cs_scan turns state 5 into 4, the LSD passes sort the column -- bit-exact"""
    rng = np.random.default_rng(12)
    for outlier in (1 << 40, -1, 40000):
        v = rng.integers(100, 10001, N, dtype=np.int64)
        v[64 + 5] = outlier                        # the sample takes chunks [c * 8 * 64, c * 8 * 64 + 64)
        got, state = _sort_with_state(gx, v)
        assert got.tobytes() == c_oracle.sort_i64(v).tobytes()
        assert state == 4, f"outlier {outlier}: state {state}"


def test_knob_off_takes_the_lsd_passes(gx):
    Column, ops, L = gx
    rng = np.random.default_rng(13)
    v = rng.integers(100, 10001, N, dtype=np.int64)
    L.lib.gx_sort_set_counting(0)
    try:
        got, state = _sort_with_state(gx, v)
    finally:
        L.lib.gx_sort_set_counting(1)
    assert got.tobytes() == c_oracle.sort_i64(v).tobytes()
    assert state == 4
