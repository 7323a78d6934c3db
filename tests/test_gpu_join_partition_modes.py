"""The round-3 partitioned probe: speculative hist-free partition into padded (partition, XCD range)
slots by a persistent scatter kernel + probe over the region table, against the oracle and against the round-2 exact
path; and the DEVICE-SIDE FALLBACK: keys skewed enough to overflow a slot must come out right through the exact
sequence enqueued behind the speculative one (no host round trip on either branch)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cudf_oracle as orc


@pytest.fixture(scope="module")
def gx():
    import torch
    assert torch.cuda.is_available()
    import cudf_amd  # noqa: F401
    from cudf_amd import Column, ops, _lib
    yield Column, ops, _lib
    _lib.lib.gx_join_set_partition_mode(1, 0)
    _lib.lib.gx_join_set_experiment(133)


def _pairs(l, r):
    return orc.canonical_pairs(l.to_numpy(), r.to_numpy())


def _inputs(rng, dtype, shape, nb, npr):
    build = rng.permutation(3 * nb)[:nb].astype(dtype)
    if shape == "dup_build":
        build[:5000] = build[20000:25000]
    probe = rng.integers(0, 4 * nb, npr).astype(dtype)
    if shape == "hot_key":           # 35 % of the probe rows carry ONE key: its (partition, range) slots overflow
        probe[rng.random(npr) < 0.35] = build[7]
    elif shape == "one_partition":   # every probe key hashes into the same partition: all eight of its slots overflow
        lg = 21
        cand = np.arange(1, 40_000_000, dtype=np.uint64)
        with np.errstate(over="ignore"):
            part = (cand * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(64 - (lg - 17))
        pool = cand[part == 3][:200_000].astype(dtype)
        probe = pool[rng.integers(0, len(pool), npr)]
        build[: len(pool) // 2] = pool[::2]
    elif shape == "edge_chains":     # long chains (60 equal build keys each) that start in the last slots of a sub-table
        lg = 21                      # and of the table: tag windows that do not settle a row, full deferral queues
        cand = np.arange(1, 6_000_000, dtype=np.uint64)
        with np.errstate(over="ignore"):
            slot = (cand * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(64 - lg)
        edge = cand[(slot & np.uint64((1 << 17) - 1)) >= np.uint64((1 << 17) - 3)]
        last = cand[slot >= np.uint64((1 << lg) - 2)]
        hot = np.concatenate([edge[:40], last[:2], cand[1000:1040]]).astype(dtype)
        build = np.concatenate([build[: nb - 60 * hot.size], np.repeat(hot, 60)])
        rng.shuffle(build)
        probe[::500] = hot[np.arange(probe[::500].size) % hot.size]
    return build, probe


CASES = [("int64", sh, md) for sh in ("uniform", "dup_build", "hot_key", "one_partition", "edge_chains") for md in (0, 2)] + \
        [("int32", "uniform", 0), ("int32", "edge_chains", 0), ("int64", "uniform", 1), ("int64", "edge_chains", 3)]


@pytest.mark.parametrize("dtype,shape,mode", CASES)
def test_speculative_partition_probe_matches_oracle(gx, dtype, shape, mode):
    """mode = the probe knobs: bit 0 rows requested at the top of a trip, bit 1 unsettled rows parked in the per-wave queue
    instead of being finished in place (0 = the default kernel)."""
    Column, ops, _lib = gx
    rng = np.random.default_rng(1234)
    nb, npr = 600_000, (1 << 20) + 1234          # table 2^21 slots -> 16 partitions
    build, probe = _inputs(rng, dtype, shape, nb, npr)
    el, er = orc.inner_join(probe, build)
    old_min = ops.HashJoin.PARTITIONED_MIN_ROWS
    ops.HashJoin.PARTITIONED_MIN_ROWS = 1 << 20
    try:
        for spec in (2, 0):                      # 2: the speculative path forced at this (small) size; 0: round-2 exact path
            _lib.lib.gx_join_set_partition_mode(spec, mode)
            hj = ops.HashJoin(Column.from_numpy(build))
            assert _lib.lib.gx_join_partition_bits(hj.key_size, hj.table_bytes) >= 3
            l, r = hj.inner_join(Column.from_numpy(probe))
            got = _pairs(l, r)
            np.testing.assert_array_equal(got[0], el)
            np.testing.assert_array_equal(got[1], er)
            if spec == 2 and shape in ("uniform", "hot_key", "edge_chains"):  # left-outer form: unmatched rows pair with JoinNoMatch
                pl, pr = hj.left_join(Column.from_numpy(probe))
                wl, wr = orc.left_join([probe], [build])
                a, b = _pairs(pl, pr), orc.canonical_pairs(wl, wr)
                np.testing.assert_array_equal(a[0], b[0])
                np.testing.assert_array_equal(a[1], b[1])
    finally:
        ops.HashJoin.PARTITIONED_MIN_ROWS = old_min
        _lib.lib.gx_join_set_partition_mode(1, 0)


def test_speculative_partition_probe_default_threshold(gx):
    """Above the default row threshold (1.7e7 rows) the speculative path is what runs without any knob: uniform keys and a
    hot key (fallback) against the closed form."""
    import torch
    Column, ops, _lib = gx
    _lib.lib.gx_join_set_partition_mode(1, 0)
    nb, n = 3_000_000, 30_000_000
    bk = Column.empty(np.int64, nb)
    bkt = bk.data[: nb * 8].view(torch.int64)
    torch.manual_seed(3)
    bkt.copy_(torch.randperm(nb, device="cuda") * 7 + 3)
    for hot in (False, True):
        pk = ops.random_column(np.int64, n, seed=9, lo=0, hi=int(nb / 0.3))
        pkt = pk.data[: n * 8].view(torch.int64)
        pkt.mul_(7).add_(3)
        if hot:
            pkt[::3] = bkt[11]
        l, r = ops.HashJoin(bk).inner_join(pk)
        li = l.data[: l.size * 4].view(torch.int32).long()
        ri = r.data[: r.size * 4].view(torch.int32).long()
        assert bool((pkt[li] == bkt[ri]).all())
        hitrows = torch.nonzero(pkt < 7 * nb).flatten()   # 7u+3 with u < nb is a build key exactly once
        assert l.size == hitrows.numel()
        assert int(li.sum().item()) == int(hitrows.sum().item())
        assert int(torch.unique(li).numel()) == l.size


@pytest.mark.parametrize("dtype,shape,kernel", [(dt, sh, k) for k in (2, 3, 4, 5, 6, 7) for dt, sh in
                                                 (("int64", "uniform"), ("int64", "dup_build"), ("int64", "hot_key"), ("int64", "one_partition"),
                                                  ("int64", "edge_chains"), ("int32", "uniform"), ("int32", "edge_chains"))])
def test_l2_resident_direct_probe_matches_oracle(gx, dtype, shape, kernel):
    """Round 5: the alternative probe kernels behind the same partition pass, region table and fallback gating.  4 / 5 =
    k_pj4_probe_tags (the LDS-tag probe without pipeline and staging, 2 / 4 rows per thread): tag windows, in-place chain
    continuation, wrap at the table's end, duplicate build keys.  2 / 3 = k_pj3_probe_direct (4 / 2 rows per thread) -- the same
    region table and fallback gating, but every row reads its home slot from the L2-resident sub-table and walks its chain there
    (no tags).  Inner and left-outer pairs against the oracle on uniform keys, duplicate build keys (rows that reserve their own
    output run), a hot key and one-partition inputs (slot overflow -> the gated exact sequence runs the same kernel over exact
    partitions), and chains that cross a sub-table's end and the table's end (the walk wraps through `mask`)."""
    Column, ops, _lib = gx
    rng = np.random.default_rng(4321)
    nb, npr = 600_000, (1 << 20) + 4321
    build, probe = _inputs(rng, dtype, shape, nb, npr)
    el, er = orc.inner_join(probe, build)
    old_min = ops.HashJoin.PARTITIONED_MIN_ROWS
    ops.HashJoin.PARTITIONED_MIN_ROWS = 1 << 20
    try:
        _lib.lib.gx_join_set_probe_kernel(kernel)
        _lib.lib.gx_join_set_partition_mode(2, 0)
        hj = ops.HashJoin(Column.from_numpy(build))
        assert _lib.lib.gx_join_partition_bits(hj.key_size, hj.table_bytes) >= 3
        l, r = hj.inner_join(Column.from_numpy(probe))
        got = _pairs(l, r)
        np.testing.assert_array_equal(got[0], el)
        np.testing.assert_array_equal(got[1], er)
        if shape in ("uniform", "hot_key", "edge_chains", "dup_build"):
            pl, pr = hj.left_join(Column.from_numpy(probe))
            wl, wr = orc.left_join([probe], [build])
            a, b = _pairs(pl, pr), orc.canonical_pairs(wl, wr)
            np.testing.assert_array_equal(a[0], b[0])
            np.testing.assert_array_equal(a[1], b[1])
    finally:
        ops.HashJoin.PARTITIONED_MIN_ROWS = old_min
        _lib.lib.gx_join_set_probe_kernel(0)
        _lib.lib.gx_join_set_partition_mode(1, 0)


@pytest.mark.parametrize("shape,xp", [(sh, xp) for xp in (0, 1, 3, 4, 5, 12, 129, 133, 133 | (1 << 24), 4 | (1 << 24)) for sh in ("uniform", "dup_build", "hot_key", "one_partition", "edge_chains")])
def test_record_form_partition_probe_matches_oracle(gx, shape, xp):
    """Round 6: the partition pass writes 12-byte {key, row} records (k_pj2_scatter_rec: 16384- or, xp bit 1, 24576-row tiles
    through 8192-position LDS windows, the ragged tail as a second launch) and the pipelined probe reads a lane's four rows as
    three 16-byte loads.  Inner and left-outer pairs against the oracle; hot_key / one_partition overflow their slots, so the
    gated EXACT sequence runs the record kernels over exactly sized partitions.  xp bit 2: the pipelined service wave of
    k_pj2_probe_pipe (a ticket is its (region, piece in region); ticket atomic and fill counter travel one trip ahead)."""
    Column, ops, _lib = gx
    rng = np.random.default_rng(777)
    nb, npr = 600_000, (1 << 20) + 1234          # 64 (42) full tiles + a tail of 1234 (17618) rows
    build, probe = _inputs(rng, "int64", shape, nb, npr)
    el, er = orc.inner_join(probe, build)
    old_min = ops.HashJoin.PARTITIONED_MIN_ROWS
    ops.HashJoin.PARTITIONED_MIN_ROWS = 1 << 20
    try:
        _lib.lib.gx_join_set_experiment(xp)
        _lib.lib.gx_join_set_partition_mode(2, 0)
        hj = ops.HashJoin(Column.from_numpy(build))
        l, r = hj.inner_join(Column.from_numpy(probe))
        got = _pairs(l, r)
        np.testing.assert_array_equal(got[0], el)
        np.testing.assert_array_equal(got[1], er)
        if shape in ("uniform", "hot_key", "edge_chains"):
            pl, pr = hj.left_join(Column.from_numpy(probe))
            wl, wr = orc.left_join([probe], [build])
            a, b = _pairs(pl, pr), orc.canonical_pairs(wl, wr)
            np.testing.assert_array_equal(a[0], b[0])
            np.testing.assert_array_equal(a[1], b[1])
    finally:
        ops.HashJoin.PARTITIONED_MIN_ROWS = old_min
        _lib.lib.gx_join_set_partition_mode(1, 0)
        _lib.lib.gx_join_set_experiment(133)


@pytest.mark.parametrize("shape,slice_rows", [(sh, sl) for sl in (0, 3) for sh in ("uniform", "dup_build", "hot_key", "edge_chains")])
def test_long_probe_overflow_list_matches_oracle(gx, shape, slice_rows):
    """Round 6, gx_join_set_experiment bit 7: k_pj2_probe_pipe<LONG> (two register sets, every load a trip ahead of its use) sends the
    rows that need a dependent read -- three or more tag candidates, a chain beyond the 16-slot window, a lane's second two-candidate
    row -- to its workgroup's overflow slice, and k_pj2_probe_rare walks them afterwards.  slice_rows = 3: the slices fill at once and the
    rows are walked in place (the fallback).  dup_build / hot_key: duplicate build keys (several pairs per row, the re-walk staging);
    inner and left-outer pairs against the oracle."""
    Column, ops, _lib = gx
    rng = np.random.default_rng(4242)
    nb, npr = 700_000, (1 << 20) + 4321
    build, probe = _inputs(rng, "int64", shape, nb, npr)
    el, er = orc.inner_join(probe, build)
    old_min = ops.HashJoin.PARTITIONED_MIN_ROWS
    ops.HashJoin.PARTITIONED_MIN_ROWS = 1 << 20
    try:
        _lib.lib.gx_join_set_experiment(133)
        _lib.lib.gx_join_set_overflow_slice(slice_rows)
        _lib.lib.gx_join_set_partition_mode(2, 0)
        hj = ops.HashJoin(Column.from_numpy(build))
        for _ in range(2):  # the object is probed again: the overflow counters are rewritten by every launch
            l, r = hj.inner_join(Column.from_numpy(probe))
            got = _pairs(l, r)
            np.testing.assert_array_equal(got[0], el)
            np.testing.assert_array_equal(got[1], er)
        pl, pr = hj.left_join(Column.from_numpy(probe))
        wl, wr = orc.left_join([probe], [build])
        a, b = _pairs(pl, pr), orc.canonical_pairs(wl, wr)
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
    finally:
        ops.HashJoin.PARTITIONED_MIN_ROWS = old_min
        _lib.lib.gx_join_set_partition_mode(1, 0)
        _lib.lib.gx_join_set_experiment(133)
        _lib.lib.gx_join_set_overflow_slice(0)
