"""GPU: the builds of the partitioned hash join (gx_join.hip, round 4): the WINDOW build (knob 0, default: k_bw_split regroups a
partition's rows by 2^12-slot window, k_bw_build composes every window in LDS and writes it once; chains that leave a window are
carried through LDS or parked, k_bw_fixup inserts the parked rows through the global tag words; a parked list that overflows
falls back to the round-2 kernels on the device) and the sub-table build (knob 2: k_bs_build / k_bs_fixup -- one workgroup owns
a 2^17-slot sub-table and claims slots through the 4-bit tags it keeps in LDS).  Checked here against the oracle's inner
join (multiset of pairs, cpp/tests/join/join_tests.cpp:1186-1210) and against the round-2 build kernel (knob 1): random keys,
keys CRAFTED to sit at the end of their sub-table (the Fibonacci slot hash is a bijection, so a home slot can be chosen) so
that hundreds / hundreds of thousands of chains cross a sub-table boundary -- the parked-row list and its in-place overflow
marks -- repeated keys, 4-byte keys, and the direct-walk kernels (lookup / semi / count) that read the same table.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cudf_oracle as orc

PHI = 0x9E3779B97F4A7C15
PHI_INV = pow(PHI, -1, 1 << 64)
SUB = 17


@pytest.fixture(scope="module")
def gx():
    import torch
    assert torch.cuda.is_available()
    import cudf_amd  # noqa: F401
    from cudf_amd import Column, ops, _lib
    yield Column, ops, _lib
    _lib.lib.gx_join_set_build_kernel(0)


def _keys_with_home(slots, lg, salt):
    """distinct uint64 keys whose table slot ((key * PHI) >> (64 - lg)) is slots[i]: key = (slot << (64 - lg) | low bits) * PHI^-1"""
    low = (np.arange(len(slots), dtype=np.uint64) * np.uint64(2654435761) + np.uint64(salt)) & np.uint64((1 << (64 - lg)) - 1)
    target = (slots.astype(np.uint64) << np.uint64(64 - lg)) | low
    with np.errstate(over="ignore"):
        return target * np.uint64(PHI_INV)      # uint64 arithmetic wraps mod 2^64: exactly the modular inverse product


def _check_join(ops, Column, build, probe, kernel, _lib):
    _lib.lib.gx_join_set_build_kernel(kernel)
    hj = ops.HashJoin(Column.from_numpy(build))
    l, r = hj.inner_join(Column.from_numpy(probe))
    gl, gr = orc.canonical_pairs(l.to_numpy(), r.to_numpy())
    el, er = orc.inner_join(probe, build)
    np.testing.assert_array_equal(gl, el)
    np.testing.assert_array_equal(gr, er)
    assert hj.inner_join_size(Column.from_numpy(probe)) == len(el)       # the direct-walk count kernel reads the same slots
    return hj


def test_crafted_keys_have_the_requested_home():
    lg = 21
    slots = np.array([0, 1, (1 << SUB) - 1, (1 << lg) - 1, 12345], np.uint64)
    k = _keys_with_home(slots, lg, 7)
    got = [((int(x) * PHI) & ((1 << 64) - 1)) >> (64 - lg) for x in k]
    assert got == [int(s) for s in slots]


@pytest.mark.parametrize("n,dtype", [(1_000_000, "int64"), (3_000_001, "int64"), (1_500_000, "int32")])
def test_subtable_build_random_keys(gx, n, dtype):
    Column, ops, _lib = gx
    rng = np.random.default_rng(n)
    if dtype == "int64":
        build = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64)
        build[: n // 50] = build[n // 2: n // 2 + n // 50]                       # 2 % of the keys twice
        probe = np.concatenate([build[rng.integers(0, n, n)], rng.integers(-2**63, 2**63 - 1, 2 * n, dtype=np.int64)])
    else:
        build = rng.permutation(1 << 23)[:n].astype(np.int32) - (1 << 22)
        probe = rng.integers(-(1 << 22), 1 << 22, 3 * n).astype(np.int32)
    rng.shuffle(probe)
    assert _lib.lib.gx_join_partition_bits(np.dtype(dtype).itemsize, _lib.lib.gx_join_table_bytes(np.dtype(dtype).itemsize, n, 0.5)) >= 3
    for kernel in ((0, 2, 1) if n <= 1_500_000 else (0, 2)):
        _check_join(ops, Column, build, probe, kernel, _lib)


@pytest.mark.parametrize("crossing", [600, 280_000], ids=["parked_list", "list_overflows_marks_in_place"])
def test_chains_that_leave_their_subtable(gx, crossing):
    """`crossing` rows are homed in the last 64 slots of the sub-tables (spread over all 16 of a 2^21-slot table): all but 64 per
    sub-table run off its end and are inserted by k_bs_fixup behind the boundary -- also behind the LAST sub-table, i.e. wrapped
    to slot 0.  280 000 of them overflow the 2^18-entry lists: knob 2 marks the rest in place and finds them in a scanning pass,
    the window build (knob 0) raises `failed` and the gated round-2 kernels build the table."""
    Column, ops, _lib = gx
    lg, n = 21, 1_000_000
    rng = np.random.default_rng(crossing)
    nsub = 1 << (lg - SUB)
    sub = rng.integers(0, nsub, crossing).astype(np.uint64)
    home = (sub << np.uint64(SUB)) + np.uint64((1 << SUB) - 64) + rng.integers(0, 64, crossing).astype(np.uint64)
    crafted = _keys_with_home(home, lg, 99)
    assert len(np.unique(crafted)) == crossing
    rest = rng.integers(0, 2**63 - 1, n - crossing, dtype=np.int64).astype(np.uint64)
    build = np.concatenate([crafted, rest]).view(np.int64)
    rng.shuffle(build)
    assert _lib.lib.gx_join_table_bytes(8, n, 0.5) == _lib.lib.gx_join_table_bytes(8, (1 << 20) - 1, 0.5)     # a 2^21-slot table
    # (> 2^22 probe rows: the partitioned probe, whose LDS tag windows straddle the sub-table boundaries the chains crossed)
    probe = np.concatenate([build, rng.integers(-2**63, 2**63 - 1, 4 * n, dtype=np.int64), crafted.view(np.int64)])
    rng.shuffle(probe)
    hj = _check_join(ops, Column, build, probe, 0, _lib)
    # the other readers of the table walk the slots directly (no tags): lookup of distinct keys, semi join
    small = probe[:300_000].copy()                                       # (below 2^22 rows: gx_join_probe walks the slots)
    sel = hj.semi_join(Column.from_numpy(small))
    np.testing.assert_array_equal(np.sort(sel.to_numpy()), orc.semi_join([small], [build]))
    l, r = hj.inner_join(Column.from_numpy(small))
    gl, gr = orc.canonical_pairs(l.to_numpy(), r.to_numpy())
    el, er = orc.inner_join(small, build)
    np.testing.assert_array_equal(gl, el)
    np.testing.assert_array_equal(gr, er)
    _check_join(ops, Column, build, probe, 2, _lib)
    _check_join(ops, Column, build, probe, 1, _lib)


def test_one_key_repeated_beyond_a_subtable_tail(gx):
    """40 000 copies of ONE key homed 1000 slots before the end of its sub-table + 30 000 copies of a key in the LAST sub-table's
    tail (the chain wraps to slot 0): most copies are parked and chained behind the boundary by the fix-up pass."""
    Column, ops, _lib = gx
    lg, n = 21, 1_000_000
    rng = np.random.default_rng(5)
    k1 = _keys_with_home(np.array([(3 << SUB) + (1 << SUB) - 1000], np.uint64), lg, 1)[0]
    k2 = _keys_with_home(np.array([(1 << lg) - 500], np.uint64), lg, 2)[0]
    build = rng.integers(0, 2**63 - 1, n, dtype=np.int64).astype(np.uint64)
    build[:40_000] = k1
    build[40_000:70_000] = k2
    build = build.view(np.int64)
    rng.shuffle(build)
    probe = np.concatenate([rng.integers(-2**63, 2**63 - 1, 3_000, dtype=np.int64), np.array([k1, k2, k1], np.uint64).view(np.int64),
                            build[rng.integers(0, n, 5_000)]])
    for kernel in (0, 2, 1):
        _check_join(ops, Column, build, probe, kernel, _lib)
