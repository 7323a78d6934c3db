"""GPU: cudf::sorted_order of a 64-bit column as a keys-only sort of (monotone rank, row) words (round 6; cudf_amd/csrc/gx_order.hip).

cub's SortPairs behind cudf::sorted_order costs the same on any value distribution and is stable (cpp/src/sort/sorted_order_radix.cu:56-179,
:81).  The round-3 pairs path sorted uniform keys fast and declined everything uneven to 4 - 8 LSD pair passes.  The word sort runs on the
keys-only paths, which round 5 made insensitive to the distribution; a last pass puts runs of equal ranks right by (key, row).  Every case:
the permutation bit-exact against the plain-C / NumPy oracle INCLUDING the order of ties, in both directions, with the path pinned
(gx_sort_order_map_info: the plan ran; how many runs went to the long-run list)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import c_oracle
from oracle import cudf_oracle as orc
from tests.test_gpu_sort_splitters import _keys

N = 40_000_003


@pytest.fixture(scope="module")
def gx():
    import torch
    assert torch.cuda.is_available()
    import cudf_amd  # noqa: F401
    from cudf_amd import Column, ops, _lib as L
    yield Column, ops, L
    L.lib.gx_sort_set_order_map(1)


def _order(gx, v, descending=False, expect_path=True):
    Column, ops, L = gx
    col = Column.from_numpy(v)
    out = Column.empty(np.int32, v.size)
    tmp = ops._run_sort(L.lib.gx_sorted_order, col.gx, col.data_ptr, None, col.size, 0, int(descending), 1, out.data_ptr)
    ops._check_sort_status(tmp)
    info = (ctypes.c_int32 * 5)()
    if expect_path:
        L.check(L.lib.gx_sort_order_map_info(ops.ptr(tmp), v.size, info, ops.stream_ptr()), "gx_sort_order_map_info")
    return out.to_numpy(), list(info)


def _expect(v, descending):
    if v.dtype == np.int64:
        return c_oracle.sorted_order_i64(v, descending=descending)
    return orc.sorted_order(v, None, not descending)


@pytest.mark.parametrize("kind", ["uniform", "normal", "lognormal", "zipf", "clusters", "normal_tail", "normal_hot", "steps"])
@pytest.mark.parametrize("descending", [False, True])
def test_int64_distributions(gx, kind, descending):
    rng = np.random.default_rng(abs(hash(kind)) % 1000 + (11 if descending else 0))
    v = rng.integers(-2**63, 2**63 - 1, N, dtype=np.int64) if kind == "uniform" else _keys(kind, rng)
    got, info = _order(gx, v, descending)
    np.testing.assert_array_equal(got, _expect(v, descending).astype(np.int32))
    assert info[1] == 0 and info[2] == 26 and info[3] == 38      # bits(n - 1) = 26 row bits, 64 - 26 rank bits
    if kind == "uniform":
        assert info[4] == 4096 and info[0] <= 4                  # every bucket is ~2^52 wide against ~2^26 ranks: lossy; no run beyond 16 rows (the clamped ends aside)


@pytest.mark.parametrize("shape", ["sorted", "reversed", "all_equal", "two_values", "narrow_range", "extremes", "uint64"])
def test_shapes(gx, shape):
    rng = np.random.default_rng(5)
    if shape == "sorted":
        v = np.sort(rng.integers(-2**62, 2**62, N, dtype=np.int64))
    elif shape == "reversed":
        v = np.sort(rng.integers(-2**62, 2**62, N, dtype=np.int64))[::-1].copy()
    elif shape == "all_equal":
        v = np.full(N, -77, np.int64)
    elif shape == "two_values":
        v = np.where(rng.random(N) < 0.3, np.int64(-5), np.int64(2**61)).astype(np.int64)
    elif shape == "narrow_range":      # the reference benchmark's own distribution (cpp/benchmarks/sort/sort.cpp:24-26): heavy ties
        v = rng.integers(100, 10001, N, dtype=np.int64)
    elif shape == "extremes":
        v = rng.integers(-2**63, 2**63 - 1, N, dtype=np.int64)
        v[::1000] = np.iinfo(np.int64).min
        v[1::1000] = np.iinfo(np.int64).max
    else:
        v = rng.integers(0, 2**64 - 1, N, dtype=np.uint64)
    for desc in (False, True):
        got, info = _order(gx, v, desc)
        exp = orc.sorted_order(v, None, not desc) if v.dtype != np.int64 else c_oracle.sorted_order_i64(v, descending=desc)
        np.testing.assert_array_equal(got, exp.astype(np.int32))
        assert info[1] == 0
        if shape in ("all_equal", "two_values", "narrow_range"):
            assert info[0] == 0       # duplicates are never a "run to fix": lossless buckets, every value owns its ranks, rows spread over them


@pytest.mark.parametrize("kind", ["normal", "uniform01", "lognormal", "both_signs", "specials"])
@pytest.mark.parametrize("descending", [False, True])
def test_float64(gx, kind, descending):
    rng = np.random.default_rng(21 + (3 if descending else 0))
    if kind == "normal":
        v = rng.standard_normal(N)
    elif kind == "uniform01":
        v = rng.random(N)
    elif kind == "lognormal":
        v = np.exp(rng.standard_normal(N) * 5.0)
    elif kind == "both_signs":
        v = rng.standard_normal(N) * 10.0 ** rng.integers(-300, 300, N)
    else:  # NaN (both signs, payloads), -0.0 / +0.0, +-inf, denormals: NaN last and equivalent, zeros equivalent, ties by row
        v = rng.standard_normal(N)
        v[::7] = np.nan
        v[1::7].view(np.uint64)[:] = 0xFFF8000000000123   # a negative NaN with a payload
        v[2::7] = -0.0
        v[3::7] = 0.0
        v[4::49] = np.inf
        v[5::49] = -np.inf
        v[6::49] = 5e-324
    got, info = _order(gx, v, descending)
    np.testing.assert_array_equal(got, orc.sorted_order(v, None, not descending).astype(np.int32))
    assert info[1] == 0


def test_a_density_spike_the_sample_cannot_see_goes_through_the_long_run_list(gx):
    """Uniform 64-bit keys (buckets 2^52 wide against 2^26 ranks at this size: 26 key bits dropped) + 60 000 CONSECUTIVE integers: distinct keys, one
    rank -- a run of 60 000 rows that only the long-run pass can order (LDS cannot hold it: the global-scratch network), and two runs of
    3000 (the LDS network); rows of the runs are shuffled so that row order and key order disagree."""
    rng = np.random.default_rng(99)
    v = rng.integers(-2**63, 2**63 - 1, N, dtype=np.int64)
    for base, cnt in ((12345678901234, 60_000), (-9876543210987, 3000), (2**61 + 17, 3000)):
        pos = rng.choice(N, cnt, replace=False)
        v[pos] = base + rng.permutation(cnt)
    for desc in (False, True):
        got, info = _order(gx, v, desc)
        np.testing.assert_array_equal(got, c_oracle.sorted_order_i64(v, descending=desc).astype(np.int32))
        assert info[0] >= 3 and info[1] == 0


@pytest.mark.parametrize("copies", [3, 100, 3000])
def test_wide_keys_with_duplicates(gx, copies):
    """Random 64-bit ids, `copies` rows each (a foreign-key column of hashed ids): every bucket is lossy, every id is ONE run of equal
    keys -- up to 16 rows the run pass's threads, beyond that the listed runs, which one WAVE each finds in order (k_om_medium: no
    workgroup, no network; a first version gave each of the n / 100 runs a workgroup).  Also multiples of 2^20 (few far-apart values)."""
    rng = np.random.default_rng(12 + copies)
    ids = rng.integers(-2**63, 2**63 - 1, N // copies + 1, dtype=np.int64)
    v = ids[rng.integers(0, ids.size, N)]
    w = rng.integers(0, N // copies + 1, N, dtype=np.int64) << 20
    for keys in (v, w):
        for desc in (False, True):
            got, info = _order(gx, keys, desc)
            np.testing.assert_array_equal(got, c_oracle.sorted_order_i64(keys, descending=desc).astype(np.int32))
            assert info[1] == 0 and (copies > 100 or info[4] > 3000), info   # (13 000 ids of 3000 rows: most buckets hold ONE id, lossless)
            if copies == 100:
                assert info[0] > 300_000, info   # the runs WERE listed


def test_listed_runs_out_of_order_are_ranked_inside_a_wave(gx):
    """Runs of 17 - 128 DISTINCT keys under one rank (consecutive integers inside uniform 64-bit keys), rows shuffled: the wave pass marks
    (k_om_medium) ranks them where it finds them"""
    rng = np.random.default_rng(77)
    v = rng.integers(-2**63, 2**63 - 1, N, dtype=np.int64)
    for j in range(2000):
        cnt = int(rng.integers(17, 129))
        pos = rng.choice(N, cnt, replace=False)
        v[pos] = int(rng.integers(-2**62, 2**62)) + rng.permutation(cnt)
    for desc in (False, True):
        got, info = _order(gx, v, desc)
        np.testing.assert_array_equal(got, c_oracle.sorted_order_i64(v, descending=desc).astype(np.int32))
        assert info[0] >= 1500 and info[1] == 0


def test_knob_off_is_the_round3_pairs_path_and_agrees(gx):
    Column, ops, L = gx
    rng = np.random.default_rng(4)
    v = rng.integers(-2**63, 2**63 - 1, N, dtype=np.int64)
    a, _ = _order(gx, v)
    L.lib.gx_sort_set_order_map(0)
    try:
        b, _ = _order(gx, v, expect_path=False)
    finally:
        L.lib.gx_sort_set_order_map(1)
    np.testing.assert_array_equal(a, b)


def test_below_the_threshold_nothing_changes(gx):
    rng = np.random.default_rng(8)
    v = rng.integers(-1000, 1000, 3_000_000, dtype=np.int64)
    got, _ = _order(gx, v, expect_path=False)
    np.testing.assert_array_equal(got, c_oracle.sorted_order_i64(v).astype(np.int32))
