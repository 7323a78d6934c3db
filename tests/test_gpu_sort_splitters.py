"""GPU: the SPLITTER mode of the cursor path (round 5; gx_sort.hip k_sp_plan / k_sp_sample / k_sp_level0, equal-width cells in
k_hf_scatter<.., 1, ..> and k_local_place, equality buckets).

cub::DeviceRadixSort behind cudf::sort costs the same on any value distribution (cpp/src/sort/sort_radix.cu:52-161).  Two levels of
bit digits do not: bell-shaped, lognormal, Zipf-like and clustered int64 columns left level-0 buckets of 6 - 13 x the mean and fell
to 4 - 8 LSD passes.  Such a column is now cut on sample-chosen splitters.  Every case: bit-exact against the plain-C oracle, and
the PATH is pinned -- splitter mode on (gx_sort_split_info), the cursor path accepted (state 3) -- so that a silent fall-back to the
LSD passes cannot hide a broken splitter plan behind a correct result.
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import c_oracle

N = 40_000_003


@pytest.fixture(scope="module")
def gx():
    import torch
    assert torch.cuda.is_available()
    import cudf_amd  # noqa: F401
    from cudf_amd import Column, ops, _lib as L
    yield Column, ops, L
    L.lib.gx_sort_set_splitters(1)


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
    big = (ctypes.c_int64 * 3)()
    L.lib.gx_sort_big_info(ops.ptr(tmp), big, ops.stream_ptr())
    return out.to_numpy(), st.value, list(info), list(big)


def _keys(kind, rng, n=N):
    if kind == "normal":        # bell-shaped around zero: the central level-0 buckets of a bit digit hold 12.8 x the mean
        return np.round(rng.standard_normal(n) * float(1 << 40)).astype(np.int64)
    if kind == "lognormal":     # densities over many octaves
        return np.round(np.exp(rng.standard_normal(n) * 3.0 + 25.0)).astype(np.int64)
    if kind == "zipf":          # floor(u^-5) clipped to 2^31: 18 % of the rows are 1 -- equality buckets
        u = np.maximum(rng.random(n), 2.0 ** -53)
        return np.floor(np.minimum(u ** -5.0, float(1 << 31))).astype(np.int64)
    if kind == "clusters":      # two far-apart clusters: two level-0 buckets of n / 2 under a bit digit
        c = np.where(rng.random(n) < 0.5, -(1 << 50), 1 << 50)
        return (c + np.round(rng.standard_normal(n) * float(1 << 30))).astype(np.int64)
    if kind == "normal_tail":   # 99 % of the keys in a narrow bell, 1 % spread over the whole range: one level-0 bucket of a bit digit holds nearly all
        v = np.round(rng.standard_normal(n) * float(1 << 30)).astype(np.int64)
        t = rng.random(n) < 0.01
        v[t] = rng.integers(-2**50, 2**50, int(t.sum()), dtype=np.int64)
        return v
    if kind == "normal_hot":    # bell-shaped + one value in 3 % of the rows (an equality bucket inside a smooth density)
        v = np.round(rng.standard_normal(n) * float(1 << 40)).astype(np.int64)
        v[rng.random(n) < 0.03] = 123456789012
        return v
    if kind == "steps":         # 300 distinct values with very different weights: nearly every bucket is an equality bucket
        vals = rng.integers(-2**62, 2**62, 300, dtype=np.int64)
        w = rng.random(300) ** 4
        return vals[rng.choice(300, n, p=w / w.sum())]
    raise AssertionError(kind)


@pytest.mark.parametrize("kind", ["normal", "normal_tail", "lognormal", "zipf", "clusters", "normal_hot"])
@pytest.mark.parametrize("descending", [False, True])
def test_uneven_value_distributions_take_the_splitters(gx, kind, descending):
    rng = np.random.default_rng(abs(hash(kind)) % 1000 + (7 if descending else 0))
    v = _keys(kind, rng)
    got, state, info, big = _sort(gx, v, descending)
    assert got.tobytes() == c_oracle.sort_i64(v, descending=descending).tobytes()
    assert info[0] == 1, f"{kind}: splitter mode not taken (state {state}, info {info})"
    assert state == 3, f"{kind}: the cursor path did not accept the splitter plan (state {state}, info {info}, big {big})"
    if kind in ("zipf", "normal_hot"):
        assert info[2] >= 1, f"{kind}: no equality bucket ({info})"
    assert big[2] < 0.1 * N, f"{kind}: {big[2]} keys went through the big-cell path"


def test_few_distinct_values_with_uneven_weights(gx):
    """300 distinct 64-bit values with very uneven weights.  Heavy values get equality buckets; k_sp_plan predicts from the sorted
    sample what share of the rows carries a value that overfills a cell without being alone in a bucket, and declines to the LSD
    passes (state 4) when the big-cell path would have to carry the column.  Either way: bit-exact, and a consistent path"""
    rng = np.random.default_rng(5)
    v = _keys("steps", rng)
    got, state, info, _ = _sort(gx, v)
    assert got.tobytes() == c_oracle.sort_i64(v).tobytes()
    assert (info[0] == 1 and state == 3 and info[2] >= 1) or (info[0] == 0 and state == 4), (state, info)


def test_uint64_keys_and_the_knob(gx):
    Column, ops, L = gx
    rng = np.random.default_rng(77)
    v = (_keys("lognormal", rng).astype(np.uint64) << np.uint64(20)) + np.uint64(5)
    got, state, info, _ = _sort(gx, v)
    assert got.tobytes() == np.sort(v).tobytes()
    assert info[0] == 1 and state == 3
    L.lib.gx_sort_set_splitters(0)          # the knob off: declined to the LSD passes as before round 5, same bytes
    try:
        got2, state2, info2, _ = _sort(gx, v)
    finally:
        L.lib.gx_sort_set_splitters(1)
    assert got2.tobytes() == got.tobytes()
    assert info2[0] == 0 and state2 in (2, 4)


def test_uniform_keys_keep_the_bit_digits(gx):
    rng = np.random.default_rng(78)
    v = rng.integers(-2**63, 2**63 - 1, N, dtype=np.int64)
    got, state, info, _ = _sort(gx, v)
    assert got.tobytes() == c_oracle.sort_i64(v).tobytes()
    assert info[0] == 0 and state == 3


def test_unsampled_outliers_under_the_splitters(gx):
    """keys far outside what the 16384-key sample saw -- INT64_MIN / MAX sentinels, a stray cluster -- land in the first / last
    bucket's end cells (the cell map clamps); still bit-exact"""
    rng = np.random.default_rng(79)
    v = _keys("normal", rng)
    v[12345] = np.iinfo(np.int64).min
    v[23456] = np.iinfo(np.int64).max
    v[100_000:100_050] = np.iinfo(np.int64).max - np.arange(50)
    v[200_000:200_050] = np.iinfo(np.int64).min + np.arange(50)
    got, state, info, _ = _sort(gx, v)
    assert got.tobytes() == c_oracle.sort_i64(v).tobytes()
    assert info[0] == 1 and state == 3
