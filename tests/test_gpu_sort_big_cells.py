"""GPU: BIG cells of the cursor-path sort (gx_sort.hip: HybridPlan::big, k_big_plan, k_hf_scatter<.., 2, ..>, the LSD passes in
their X mode, k_big_distribute; round 4).

Two MSD partition levels bring every cell down to <= 8192 keys -- unless one VALUE is repeated more often than that: its copies
share every bit, so they land in one cell whatever the digits (zeros, a sentinel, a default id).  Before round 4 that cell
overflowed its slot and the WHOLE column fell back to eight LSD passes (5.4x the time at 1e9 rows: profiles/
r3_run23_local_place_sizes_ab.txt).  Now the overflowing cells are sorted on their own and every other cell keeps the fast path;
cub::DeviceRadixSort behind the reference's cudf::sort is insensitive to the value distribution (cpp/src/sort/sort_radix.cu:52-161).
Each case is compared bit for bit with the plain-C oracle, and the test pins what the device decided: cursor path accepted
(state 3), big cells handled through X (mode 1) with the expected number of keys -- or, when the big cells hold more than half
of the column, the whole-column fallback (mode 0), still bit-exact.
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import c_oracle

N = 40_000_000  # > 2^25: the smallest size class of the cursor path


@pytest.fixture(scope="module")
def gx():
    import torch
    assert torch.cuda.is_available()
    import cudf_amd  # noqa: F401
    from cudf_amd import Column, ops, _lib as L
    # this module pins the BIT-DIGIT levels and their big-cell machinery (round 4); since round 5 a column that uneven is cut on
    # splitters first (tests/test_gpu_sort_splitters.py), so the splitter mode is switched off here
    L.lib.gx_sort_set_splitters(0)
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
    big = (ctypes.c_int64 * 3)()
    L.check(L.lib.gx_sort_big_info(ops.ptr(tmp), big, ops.stream_ptr()), "gx_sort_big_info")
    info = (ctypes.c_int32 * 8)()
    L.check(L.lib.gx_sort_info(ops.ptr(tmp), info, ops.stream_ptr()), "gx_sort_info")
    return out.to_numpy(), st.value, list(big), list(info)


@pytest.mark.parametrize("copies,descending", [(100_000, False), (1_000_000, True), (12_000_000, False)])
def test_one_hot_value_costs_its_cell_not_the_column(gx, copies, descending):
    rng = np.random.default_rng(copies)
    v = rng.integers(-2**63, 2**63 - 1, N, dtype=np.int64)
    hot = v[12345]
    v[rng.choice(N, copies, replace=False)] = hot
    got, state, big, info = _sort(gx, v, descending)
    assert got.tobytes() == c_oracle.sort_i64(v, descending=descending).tobytes()
    assert state == 3                                     # the sample was representative: the cursor path kept the column
    assert big[0] == 1 and big[1] == 1                    # ONE big cell, sorted through X
    assert copies <= big[2] <= copies + 3 * 8192          # X = the copies + the cell's ordinary keys
    assert info[1] == 1                                   # hybrid ok: every other cell took the LDS cell sort


def test_several_hot_values_and_the_extremes(gx):
    rng = np.random.default_rng(7)
    v = rng.integers(-2**63, 2**63 - 1, N, dtype=np.int64)
    hots = [np.iinfo(np.int64).min, np.iinfo(np.int64).max, 0, -1, int(v[5]), int(v[6]) ^ 1]
    pos = rng.permutation(N)
    k = 0
    for i, h in enumerate(hots):
        c = 20_000 + 150_000 * i
        v[pos[k:k + c]] = h
        k += c
    got, state, big, info = _sort(gx, v)
    assert got.tobytes() == c_oracle.sort_i64(v).tobytes()
    assert state == 3 and big[0] == 1 and 5 <= big[1] <= 6 and big[2] >= k     # (0 and -1 are neighbours but sit in different cells)
    got, state, big, _ = _sort(gx, v, True)
    assert got.tobytes() == c_oracle.sort_i64(v, descending=True).tobytes() and state == 3 and big[0] == 1


def test_hot_value_in_int32_keys(gx):
    rng = np.random.default_rng(9)
    v = rng.integers(-2**31, 2**31 - 1, N).astype(np.int32)
    v[rng.choice(N, 700_000, replace=False)] = -77
    got, state, big, _ = _sort(gx, v)
    assert got.tobytes() == c_oracle.sort_32(v).tobytes()
    assert state == 3 and big[0] == 1 and big[1] >= 1 and big[2] >= 700_000
    u = v.view(np.uint32)
    got, state, big, _ = _sort(gx, u, True)
    assert got.tobytes() == c_oracle.sort_32(u, descending=True).tobytes() and state == 3 and big[0] == 1


def test_big_cells_beyond_half_the_column_fall_back_to_the_lsd_passes(gx):
    """75 % of the rows carry one value: X would not fit the two work areas (half the padded level-0 buffer each) -> the whole-column
    LSD passes, as before round 4"""
    rng = np.random.default_rng(11)
    v = rng.integers(-2**63, 2**63 - 1, N, dtype=np.int64)
    v[rng.choice(N, int(0.75 * N), replace=False)] = 424242
    got, state, big, info = _sort(gx, v)
    assert got.tobytes() == c_oracle.sort_i64(v).tobytes()
    assert big[0] == 0 and info[1] == 0 and info[7] > 0   # no X sort; hybrid not ok; LSD passes active


def test_few_distinct_values_every_cell_big(gx):
    """1000 distinct values x 40 000 copies: every populated cell is big and together they are the whole column -> fallback"""
    rng = np.random.default_rng(12)
    vals = rng.integers(-2**63, 2**63 - 1, 1000, dtype=np.int64)
    v = vals[rng.integers(0, 1000, N)]
    got, state, big, info = _sort(gx, v)
    assert got.tobytes() == c_oracle.sort_i64(v).tobytes()
    assert big[0] == 0


def test_uniform_keys_have_no_big_cell(gx):
    rng = np.random.default_rng(13)
    v = rng.integers(-2**63, 2**63 - 1, N, dtype=np.int64)
    got, state, big, info = _sort(gx, v)
    assert got.tobytes() == c_oracle.sort_i64(v).tobytes()
    assert state == 3 and big == [0, 0, 0] and info[1] == 1 and info[7] == -1   # k_plan2 marked every LSD pass as skipped


@pytest.mark.parametrize("dist", ["zipf", "two_clusters", "normal"])
def test_value_distributions_the_levels_cannot_split(gx, dist):
    """Zipf-like values (floor(u^-5): 18 % of the rows are 1, 97 % below 2^24) and two narrow clusters far apart put nearly every
    key into cells that must overflow.  Round 4 run 14 measured 56 - 240 ms per 1e9 rows for such columns (level 1: an atomic per
    overflowing (tile, bin) on ONE flag word); now the sample (k_hf_plan stage 1) or the exact level-0 histogram (stage 2) sees
    that more keys sit in big cells than X holds and the plan goes straight to the LSD passes: state 4, bit-exact.  Bell-shaped
    values around zero take the sign fold; at 4e7 rows their central buckets are overfull, not hopeless: whichever way the device
    decides (big cells through X, or the LSD passes), the result is bit-exact and the state is a decided one."""
    rng = np.random.default_rng(31)
    if dist == "zipf":
        v = np.minimum(np.floor(np.maximum(rng.random(N), 2.0 ** -53) ** -5.0), float(1 << 31)).astype(np.int64)
    elif dist == "two_clusters":
        v = rng.integers(0, 1 << 20, N, dtype=np.int64) + np.where(rng.random(N) < 0.5, np.int64(1) << 60, np.int64(0))
    else:
        v = np.round(rng.standard_normal(N) * float(1 << 40)).astype(np.int64)
    for descending in (False, True):
        got, state, big, info = _sort(gx, v, descending)
        assert got.tobytes() == c_oracle.sort_i64(v, descending=descending).tobytes()
        if dist == "normal":
            assert state in (3, 4), (state, big, info)
        else:
            assert state == 4 and big[0] == 0 and info[1] == 0 and info[7] > 0, (state, big, info)
