"""GPU: FLOAT64 keys on the cursor path (round 5).

The reference sorts a float column as (isnan * (idx + 1), value) pairs with cub's STABLE radix sort (cpp/src/sort/sort_radix.cu:
36-117): NaNs end up last in input order, and -0.0 / +0.0 -- which cub's key transform merges -- keep their input order.  On a
column with no NaN and no -0.0 every pair of equal-comparing keys is bit-identical, so an UNORDERED sort gives the same bytes: such
columns -- ordinary float data -- take the cursor path on the IEEE total-order flip (K_FTOTAL), with the splitter mode doing what
bit digits cannot on sign / exponent / mantissa patterns.  Level 0 checks EVERY key; one NaN or -0.0 and the stable look-back path
sorts the column (decided on the device).  Every case: bit-exact against the oracle, path pinned.
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cudf_oracle as orc

N = 36_000_001


@pytest.fixture(scope="module")
def gx():
    import torch
    assert torch.cuda.is_available()
    import cudf_amd  # noqa: F401
    from cudf_amd import Column, ops, _lib as L
    yield Column, ops, L
    L.lib.gx_sort_set_float_cursor(1)


def _sort(gx, v, descending=False):
    Column, ops, L = gx
    col = Column.from_numpy(v)
    out = Column.empty(v.dtype, v.size)
    tmp = ops._run(L.lib.gx_sort_keys, col.gx, col.data_ptr, out.data_ptr, col.size, int(descending))
    ops._check_sort_status(tmp)
    st = ctypes.c_int32(-1)
    L.check(L.lib.gx_sort_cursor_state(ops.ptr(tmp), ctypes.byref(st), ops.stream_ptr()), "gx_sort_cursor_state")
    info = (ctypes.c_int32 * 4)()
    L.check(L.lib.gx_sort_split_info(ops.ptr(tmp), info, ops.stream_ptr()), "gx_sort_split_info")
    return out.to_numpy(), st.value, list(info)


def _clean_expected(v, descending):
    s = np.sort(v)          # no NaN, no -0.0: equal keys are bit-identical, any correct sort gives these bytes
    return s[::-1].copy() if descending else s


@pytest.mark.parametrize("kind", ["normal", "uniform01", "lognormal", "wide_exponents", "all_negative"])
@pytest.mark.parametrize("descending", [False, True])
def test_clean_float64_columns_take_the_cursor_path(gx, kind, descending):
    rng = np.random.default_rng(abs(hash(kind)) % 997)
    if kind == "normal":
        v = rng.standard_normal(N)
    elif kind == "uniform01":
        v = rng.random(N)
    elif kind == "lognormal":
        v = np.exp(rng.standard_normal(N) * 4.0)
    elif kind == "wide_exponents":           # every exponent from 1e-300 to 1e300, both signs, +-Inf
        v = np.ldexp(rng.random(N) + 0.5, rng.integers(-990, 990, N)) * np.where(rng.random(N) < 0.5, -1.0, 1.0)
        v[::100_003] = np.inf
        v[7::100_019] = -np.inf
    else:
        v = -np.abs(rng.standard_normal(N)) - 1e-9
    v[v == 0] = 1.0                          # no zero of either sign by accident
    got, state, info = _sort(gx, v, descending)
    assert got.tobytes() == _clean_expected(v, descending).tobytes()
    assert state == 3, f"{kind}: state {state} (info {info})"


@pytest.mark.parametrize("what", ["nan_sampled", "nan_unsampled", "negzero_unsampled", "mixed"])
def test_nan_or_negative_zero_sends_the_column_to_the_stable_path(gx, what):
    rng = np.random.default_rng(11)
    v = rng.standard_normal(N)
    if what == "nan_sampled":
        v[::4096] = np.nan                   # rows 0, 4096, ...: inside sampled chunks
    elif what == "nan_unsampled":
        v[64 + 5] = np.nan                   # the sample takes chunks [c * 8 * 64, c * 8 * 64 + 64): row 69 is never sampled
        v[64 + 7] = -np.nan
    elif what == "negzero_unsampled":
        v[64 + 9] = -0.0
        v[1000:1010] = 0.0
    else:
        v[::1013] = np.nan
        v[3::1019] = -np.nan
        v[7::997] = -0.0
        v[9::991] = 0.0
        v[11::983] = np.inf
    for descending in (False, True):
        got, state, _ = _sort(gx, v, descending)
        assert got.tobytes() == orc.sort_keys(v, ascending=not descending).tobytes()   # the reference's order incl. NaN payloads and zero signs
        assert state in (0, 2), f"{what}: state {state}"


def test_knob_off_keeps_the_look_back_path(gx):
    Column, ops, L = gx
    rng = np.random.default_rng(12)
    v = rng.random(N)
    L.lib.gx_sort_set_float_cursor(0)
    try:
        got, state, _ = _sort(gx, v)
    finally:
        L.lib.gx_sort_set_float_cursor(1)
    assert got.tobytes() == np.sort(v).tobytes()
    assert state == 0
