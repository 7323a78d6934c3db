"""GPU: the SHIPPED sharded operators (include/cudf_amd/gxd.h -> cudf_amd/cpp/src/distributed.cpp) with W > 1 ranks.

SURVEY.md 8(e): when only one GPU can be reached, "validate the partition/exchange logic with N logical ranks on one device
(device-to-device copies standing in for RCCL)".  `gxd_comm_create_loopback` gives W communicators that share an in-process
fabric behind the operators' transport seam; W host threads drive them concurrently, so every r != rank branch -- the count
matrix indexing, the send offsets and receive positions of the grouped exchange, the (rank << s) | row codes with several
sources, the splitter all-gather, the groupby's second payload exchange -- executes exactly as it would over RCCL, and the
concatenated result is compared with the oracle on the concatenated input.  (The reference's analogue: the shuffle of
cpp/libcudf_streaming/src/partition_utils.cpp:72-117 + partition.cpp:56-80 and cudf_polars' collectives/sort.py.)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import c_oracle
from oracle import cudf_oracle as orc


def _expected_pairs(probe, build):
    """canonically sorted (left, right) pairs of the inner join: the plain-C oracle (a hash join) for the large cases -- the
    NumPy oracle's sort-merge takes ~10 s per case at these sizes -- pinned to the NumPy one on the small case"""
    l, r = c_oracle.inner_join_i64(probe, build)
    return orc.canonical_pairs(l.astype(np.int64), r.astype(np.int64))


def _shards(rng, total, world, kind):
    """split [0, total) into `world` contiguous shards: even / skewed (rank 0 holds half) / one empty shard"""
    if kind == "even":
        cuts = np.linspace(0, total, world + 1).astype(np.int64)
    elif kind == "skewed":
        w = np.array([world] + [1] * (world - 1), dtype=np.float64)
        cuts = np.concatenate([[0], np.round(np.cumsum(w) / w.sum() * total)]).astype(np.int64)
    elif kind == "empty":                                     # the middle rank has nothing, the others differ in size
        w = rng.integers(1, 5, world).astype(np.float64)
        w[world // 2] = 0
        cuts = np.concatenate([[0], np.round(np.cumsum(w) / w.sum() * total)]).astype(np.int64)
    else:
        raise ValueError(kind)
    cuts[-1] = total
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]


@pytest.fixture(scope="module", params=[2, 3, 8])
def fabric(request):
    import torch
    from cudf_amd import gxd
    torch.cuda.set_device(0)
    comms = gxd.Communicator.loopback(request.param)
    yield comms
    for c in comms:
        c.close()


def _cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("dtype,kind,chunks,slot_scale", [
    ("int64", "even", 1, 0.0), ("int64", "skewed", 4, 0.0), ("int64", "empty", 3, 0.0), ("int64", "skewed", 2, 0.4),
    ("float64", "empty", 2, 0.0), ("int32", "skewed", 1, 0.0), ("float64", "even", 2, 0.4)])
def test_loopback_sort(fabric, dtype, kind, chunks, slot_scale):
    """gxd_sort over W ranks: the concatenation of the per-rank results in rank order is the oracle's sort of the concatenated
    input -- bit-exact (floats included: NaN last, -0.0 / 0.0 in the oracle's order)."""
    from cudf_amd import gxd
    W = len(fabric)
    rng = np.random.default_rng(100 + W)
    total = 3_500_003
    if dtype == "float64":
        v = rng.standard_normal(total) * 1e6
        v[::1013] = np.nan
        v[3::1019] = -np.nan                                # either sign of NaN sorts last (common_utils.cuh:157-169)
        v[7::997] = -0.0
        v[9::991] = np.inf
    elif dtype == "int32":
        v = rng.integers(-2**31, 2**31 - 1, total).astype(np.int32)
    else:
        v = rng.integers(-2**62, 2**62, total, dtype=np.int64)
        v[: total // 7] = v[total // 2]                     # a heavy duplicate: splitters may coincide
    sh = _shards(rng, total, W, kind)
    ins = [_cuda(v[a:b]) for a, b in sh]
    gxd.set_slot_scale(slot_scale)
    try:
        outs = gxd.run_ranks(fabric, lambda r, c: c.sort(ins[r], chunks=chunks, force_exchange=True))
    finally:
        gxd.set_slot_scale(0.0)
    got = np.concatenate([o.cpu().numpy() for o in outs])
    exp = orc.sort_keys(v)
    if dtype == "float64":
        # keys that COMPARE equal but differ in bits (-0.0 / 0.0, the NaN payloads) may arrive in any source / chunk order:
        # cudf::sort is not a stable sort, so values are compared, not the order inside a tie
        assert np.array_equal(got, exp, equal_nan=True)
        assert np.signbit(got[got == 0]).sum() == np.signbit(v[v == 0]).sum()
    else:
        assert got.tobytes() == exp.tobytes()
    if kind == "even" and slot_scale == 0.0 and dtype == "int64" and W > 1:
        sizes = np.array([o.numel() for o in outs])
        assert sizes.max() < 2.5 * total / W, sizes       # the sampled splitters balance the ranges (duplicates aside)


@pytest.mark.parametrize("dtype,kind,shape", [
    ("int64", "even", "uniform"), ("int64", "skewed", "uniform"), ("int64", "empty", "hot_value"), ("int32", "skewed", "uniform"),
    ("int64", "even", "narrow_range"), ("int64", "even", "heavy_bin"), ("int64", "even", "outlier_high"), ("int64", "skewed", "outlier_low"),
    ("int32", "even", "outlier_high")])
def test_loopback_sort_fused(fabric, dtype, kind, shape):
    """gxd_sort with the exchange BETWEEN the sort's two partition levels (gx_sortx_*): level 0 on every rank with common digit
    positions, whole level-0 bins dealt to ranks by the all-gathered histogram, one span per peer, level 1 + cell sort on the
    receiver.  Bit-exact against the oracle on the concatenation; `hot_value` puts a big cell on one receiver (sorted through X),
    `heavy_bin` one level-0 bin too heavy for a rank -- the collective decision falls back to the sample-sort path.
    `outlier_high` / `outlier_low`: ONE key per rank with a bit no rank's SAMPLE saw -- a sentinel above ids
    below 1e12, one odd key among even ones -- at a row the 1-in-8 / 1-in-32 chunk sample does not read.  The digit positions and
    the receiver's skipped bytes come from the sampled masks; level 0's exact masks must expose the outlier and every rank must
    take the sample-sort path (the fused path would truncate its level-0 digit / leave it unsorted inside its cell)."""
    import torch
    from cudf_amd import gxd
    W = len(fabric)
    rng = np.random.default_rng(400 + W)
    total = W * 2_300_000 + 12_345
    if dtype == "int32":
        v = rng.integers(-2**31, 2**31 - 1, total).astype(np.int32)
        if shape == "outlier_high":
            v = rng.integers(0, 1 << 27, total).astype(np.int32)
    else:
        v = rng.integers(-2**63, 2**63 - 1, total, dtype=np.int64)
        if shape == "outlier_high":
            v = rng.integers(0, 1_000_000_000_000, total, dtype=np.int64)
        elif shape == "outlier_low":
            v &= np.int64(-2)                                                    # every key even
        if shape == "narrow_range":
            v = rng.integers(0, 1_000_000_000_000, total, dtype=np.int64)       # the top digit uses 233 of 256 bins, not byte aligned
        elif shape == "hot_value":
            v[rng.choice(total, 300_000, replace=False)] = v[17]                # > 8192 copies of one key: a big cell
        elif shape == "heavy_bin":
            hot = rng.random(total) < 0.8
            v[hot] = (v[hot] & np.int64((1 << 52) - 1)) | np.int64(37 << 52)     # 80 % of all keys inside one level-0 bin
    sh = _shards(rng, total, W, kind)
    if shape.startswith("outlier"):
        for a, b in sh[:: max(1, W - 1)]:    # first and last rank: row 100 of a shard lies in a chunk the sample skips (stride >= 8)
            if b - a > 200:
                v[a + 100] = (np.iinfo(v.dtype).max if a == 0 else -1) if shape == "outlier_high" else v[a + 100] | 1
    ins = [_cuda(v[a:b]) for a, b in sh]
    gxd.set_sort_mode(2)                                                          # fused from 2^21 rows per rank
    try:
        def rank_fn(r, c):
            out = c.sort(ins[r], force_exchange=True)
            return out.cpu().numpy(), c.last_timing()[0]
        outs = gxd.run_ranks(fabric, rank_fn)
    finally:
        gxd.set_sort_mode(0)
    got = np.concatenate([o[0] for o in outs])
    assert got.tobytes() == orc.sort_keys(v).tobytes()
    fused = [o[1] == -1.0 for o in outs]
    assert all(fused) or not any(fused)                                           # a collective decision
    if shape.startswith("outlier"):
        assert not any(fused)                 # the exact masks of level 0 expose the unsampled bit: every rank falls back together
    elif shape == "heavy_bin":
        if W >= 8:
            assert not any(fused)             # 80 % of 18 M keys do not fit one rank's receive area: the sample-sort path ran
    else:
        assert all(fused)
        sizes = np.array([len(o[0]) for o in outs])
        if shape == "uniform":
            assert sizes.max() < 1.3 * total / W + 70_000, sizes                  # whole bins: the imbalance is below one bin


@pytest.mark.parametrize("kind,chunks,slot_scale,row_bits,nbuild,nprobe", [
    ("even", 1, 0.0, 0, 2_600_000, 5_000_011),      # partitioned probe, (rank << s) | row decode on both sides
    ("skewed", 4, 0.0, 0, 2_600_000, 5_000_011),    # chunked probe overlapping the exchange
    ("empty", 2, 0.4, 0, 2_600_000, 4_000_003),     # one empty shard + every chunk's slots overflow -> exact re-partition
    ("skewed", 1, 0.0, 20, 5_000_000, 4_000_011),   # row field of 2^20: build shards exceed it -> positions + gather fallback,
                                                     #   probe shards cut into several chunks by the row field
    ("even", 2, 0.0, 0, 40_000, 300_000),           # small tables: the plain probe + segment-table gather path
])
def test_loopback_join(fabric, kind, chunks, slot_scale, row_bits, nbuild, nprobe):
    """gxd_join_build + gxd_join_probe over W ranks: the union of the per-rank (global probe row, global build row) pairs is the
    oracle's inner join of the concatenated inputs (multiset equality after canonical sort, cpp/tests/join/join_tests.cpp:1186-1210).
    Build keys repeat (pairs > probe rows on some ranks), every rank probes twice against the same table."""
    from cudf_amd import gxd
    W = len(fabric)
    rng = np.random.default_rng(200 + W)
    build = rng.permutation(3 * nbuild)[:nbuild].astype(np.int64) * 7 - 3 * nbuild
    build[: nbuild // 20] = build[nbuild // 2: nbuild // 2 + nbuild // 20]          # 5 % of the build keys occur twice
    probe = rng.integers(-3 * nbuild, 18 * nbuild, nprobe).astype(np.int64)
    hit = rng.random(nprobe) < 0.4
    probe[hit] = build[rng.integers(0, nbuild, int(hit.sum()))]
    bsh = _shards(rng, nbuild, W, kind)
    psh = _shards(rng, nprobe, W, "skewed" if kind == "even" else "even")
    bins = [_cuda(build[a:b]) for a, b in bsh]
    pins = [_cuda(probe[a:b]) for a, b in psh]
    half = [_cuda(probe[a:a + (b - a) // 3]) for a, b in psh]

    def rank_fn(r, c):
        hj = gxd.HashJoin(c, bins[r], force_exchange=True)
        try:
            l1, r1 = hj.inner_join(pins[r], chunks=chunks)
            l2, r2 = hj.inner_join(half[r], chunks=1)        # probe-many on the same table
            return l1.cpu().numpy(), r1.cpu().numpy(), l2.cpu().numpy(), r2.cpu().numpy()
        finally:
            hj.close()

    gxd.set_slot_scale(slot_scale)
    gxd.set_row_bits(row_bits)
    try:
        outs = gxd.run_ranks(fabric, rank_fn)
    finally:
        gxd.set_slot_scale(0.0)
        gxd.set_row_bits(0)
    gl = np.concatenate([o[0] for o in outs])
    gr = np.concatenate([o[1] for o in outs])
    el, er = orc.inner_join(probe, build) if nbuild < 100_000 else _expected_pairs(probe, build)
    if nbuild < 100_000:
        cl, cr = _expected_pairs(probe, build)              # (the two oracles agree)
        np.testing.assert_array_equal(cl, el)
        np.testing.assert_array_equal(cr, er)
    a, b = orc.canonical_pairs(gl, gr)
    np.testing.assert_array_equal(a, el)
    np.testing.assert_array_equal(b, er)
    # second probe: rank r's third of its shard -- global probe rows are (first row of r's SECOND-call shard) + local row
    third = np.concatenate([probe[a:a + (b - a) // 3] for a, b in psh])
    el2, er2 = _expected_pairs(third, build)
    a2, b2 = orc.canonical_pairs(np.concatenate([o[2] for o in outs]), np.concatenate([o[3] for o in outs]))
    np.testing.assert_array_equal(a2, el2)
    np.testing.assert_array_equal(b2, er2)
    if W > 1 and nbuild >= 1_000_000:
        assert sum(1 for o in outs if len(o[0]) > 0) >= 2    # pairs come from several ranks: decode saw >= 2 sources


@pytest.mark.parametrize("kdtype,vdtype,kind,slot_scale,ngroups", [
    ("int32", "float64", "even", 0.0, 200_000), ("int64", "int32", "skewed", 0.0, 50_000),
    ("int64", "float64", "empty", 0.4, 300_000),   # the exact re-partition used to clobber the partial counts (MISC_D)
    ("int32", "int64", "skewed", 0.4, 7)])
def test_loopback_groupby(fabric, kdtype, vdtype, kind, slot_scale, ngroups):
    """gxd_groupby_sum_count over W ranks: every group ends on exactly one rank; keys, counts and integer sums bit-exact, float
    sums (integer-valued here, so exact in any order) equal."""
    from cudf_amd import gxd
    W = len(fabric)
    rng = np.random.default_rng(300 + W)
    n = 4_000_037
    ids = (rng.integers(0, 2**40, ngroups) if kdtype == "int64" else rng.permutation(2**24)[:ngroups]).astype(kdtype)
    keys = ids[rng.integers(0, ngroups, n)]
    vals = rng.integers(-50, 100, n).astype(vdtype)
    sh = _shards(rng, n, W, kind)
    kin = [_cuda(keys[a:b]) for a, b in sh]
    vin = [_cuda(vals[a:b]) for a, b in sh]
    gxd.set_slot_scale(slot_scale)
    try:
        outs = gxd.run_ranks(fabric, lambda r, c: tuple(t.cpu().numpy() for t in
                                                        c.groupby_sum_count(kin[r], vin[r], max_groups=1 << 16, force_exchange=True)))
    finally:
        gxd.set_slot_scale(0.0)
    for k, _, _ in outs:
        assert np.all(np.diff(k) > 0)                          # ascending, distinct inside a rank
    gk = np.concatenate([o[0] for o in outs])
    gs = np.concatenate([o[1] for o in outs])
    gc = np.concatenate([o[2] for o in outs])
    uk, inv = np.unique(keys, return_inverse=True)
    assert len(gk) == len(uk)                                  # every group on exactly ONE rank
    o = np.argsort(gk, kind="stable")
    np.testing.assert_array_equal(gk[o], uk)
    np.testing.assert_array_equal(gc[o], np.bincount(inv))
    es = np.bincount(inv, weights=vals.astype(np.float64))
    np.testing.assert_array_equal(gs[o].astype(np.float64), es)
    if W > 1 and ngroups > 1000:
        assert sum(1 for k, _, _ in outs if len(k) > 0) == W   # the hash split reaches every rank


def test_loopback_world_sizes_and_errors():
    """world 5 (neither a power of two nor a divisor of anything): hash destinations are a multiply-shift, not a mask;
    more than 16 ranks is rejected with a message."""
    import torch
    from cudf_amd import gxd
    torch.cuda.set_device(0)
    with pytest.raises(RuntimeError, match="at most 16 ranks"):
        gxd.Communicator.loopback(17)
    comms = gxd.Communicator.loopback(5)
    try:
        rng = np.random.default_rng(5)
        build = rng.permutation(4_000_000)[:1_500_000].astype(np.int64)
        probe = rng.integers(0, 5_000_000, 2_000_000).astype(np.int64)
        bsh = _shards(rng, len(build), 5, "skewed")
        psh = _shards(rng, len(probe), 5, "empty")
        bins = [_cuda(build[a:b]) for a, b in bsh]
        pins = [_cuda(probe[a:b]) for a, b in psh]

        def rank_fn(r, c):
            hj = gxd.HashJoin(c, bins[r])
            try:
                l, rr = hj.inner_join(pins[r])
                return l.cpu().numpy(), rr.cpu().numpy()
            finally:
                hj.close()
        outs = gxd.run_ranks(comms, rank_fn)
        a, b = orc.canonical_pairs(np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs]))
        el, er = orc.inner_join(probe, build)
        np.testing.assert_array_equal(a, el)
        np.testing.assert_array_equal(b, er)
        v = rng.integers(-2**40, 2**40, 1_000_003, dtype=np.int64)
        ssh = _shards(rng, len(v), 5, "even")
        vin = [_cuda(v[a:b]) for a, b in ssh]
        souts = gxd.run_ranks(comms, lambda r, c: c.sort(vin[r]).cpu().numpy())
        assert np.concatenate(souts).tobytes() == np.sort(v).tobytes()
    finally:
        for c in comms:
            c.close()
