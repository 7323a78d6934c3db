"""GPU parity: hash join / groupby through the C ABI vs the CPU oracle and the reference's goldens."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cudf_oracle as orc
from tests.golden import reference_vectors as gv


@pytest.fixture(scope="module")
def gx():
    import torch
    assert torch.cuda.is_available()
    import cudf_amd
    from cudf_amd import Column, ops
    return Column, ops


def _pairs(l, r):
    return orc.canonical_pairs(l.to_numpy(), r.to_numpy())


@pytest.mark.parametrize("dtype", ["int64", "int32", "uint64", "uint32"])
def test_inner_join_matches_oracle(gx, dtype):
    Column, ops = gx
    rng = np.random.default_rng(1)
    for nl, nr, dup in [(0, 5, False), (5, 0, False), (1, 1, False), (1000, 300, False), (70_000, 9_000, True),
                        (9_000, 70_000, True), (300_001, 100_003, False)]:
        if nr:
            right = rng.permutation(max(nr * 3, 1))[:nr].astype(dtype)
            if dup and nr > 10:
                right[: nr // 10] = right[nr // 2: nr // 2 + nr // 10]  # duplicate build keys
        else:
            right = np.empty(0, dtype)
        left = rng.integers(0, max(nr * 4, 1), nl).astype(dtype)
        l, r = ops.inner_join(Column.from_numpy(left), Column.from_numpy(right))
        gl, gr = _pairs(l, r)
        el, er = orc.inner_join(left, right)
        np.testing.assert_array_equal(gl, el, err_msg=f"{dtype} {nl}x{nr}")
        np.testing.assert_array_equal(gr, er)


def test_hash_join_class_sequential_probes_and_sizes(gx):
    Column, ops = gx
    rng = np.random.default_rng(2)
    build = (rng.permutation(50_000)[:20_000] * 7).astype(np.int64)
    hj = ops.HashJoin(Column.from_numpy(build), load_factor=0.5)
    for seed in range(3):
        probe = np.random.default_rng(seed).integers(0, 400_000, 60_000).astype(np.int64)
        pc = Column.from_numpy(probe)
        el, er = orc.inner_join(probe, build)
        assert hj.inner_join_size(pc) == len(el)
        l, r = hj.inner_join(pc)
        gl, gr = _pairs(l, r)
        np.testing.assert_array_equal(gl, el)
        np.testing.assert_array_equal(gr, er)
        # exact output_size supplied by the caller (hash_join.hpp:145-150)
        l2, r2 = hj.inner_join(pc, output_size=len(el))
        assert _pairs(l2, r2)[0].tolist() == el.tolist()
        ll, lr = hj.left_join(pc)
        xl, xr = orc.left_join(probe, build)
        gl, gr = _pairs(ll, lr)
        np.testing.assert_array_equal(gl, xl)
        np.testing.assert_array_equal(gr, xr)
    with pytest.raises(TypeError):
        hj.inner_join(Column.from_numpy(np.zeros(3, np.int32)))
    with pytest.raises(ValueError):
        ops.HashJoin(Column.from_numpy(build), load_factor=0.0)


@pytest.mark.parametrize("nulls_equal", [True, False])
def test_join_with_nulls(gx, nulls_equal):
    Column, ops = gx
    rng = np.random.default_rng(3)
    left = rng.integers(0, 50, 2000).astype(np.int64)
    right = rng.integers(0, 50, 300).astype(np.int64)
    lv = rng.random(2000) > 0.1
    rv = rng.random(300) > 0.1
    l, r = ops.inner_join(Column.from_numpy(left, lv), Column.from_numpy(right, rv), nulls_equal)
    gl, gr = _pairs(l, r)
    el, er = orc.inner_join([left], [right], [lv], [rv], nulls_equal)
    np.testing.assert_array_equal(gl, el)
    np.testing.assert_array_equal(gr, er)


@pytest.mark.parametrize("case", [c for c in gv.JOIN if len(c["left"]) == 1 and c.get("how", "inner") == "inner"],
                         ids=lambda c: c["name"])
def test_reference_golden_join_single_key(gx, case):
    Column, ops = gx
    (lc, lm), (rc, rm) = gv.col(case["left"][0], case["dtype"]), gv.col(case["right"][0], case["dtype"])
    for eq in case["nulls_equal"]:
        l, r = ops.inner_join(Column.from_numpy(lc, lm), Column.from_numpy(rc, rm), eq)
        l, r = l.to_numpy(), r.to_numpy()
        if "expected_pairs" in case:
            assert sorted(zip(l.tolist(), r.tolist())) == sorted(case["expected_pairs"])
        else:
            lp = [np.array(c) for c in case["left_payload"]]
            rp = [np.array(c) for c in case["right_payload"]]
            rows = sorted(tuple(int(c[i]) for c in lp) + tuple(int(c[j]) for c in rp) for i, j in zip(l, r))
            assert rows == sorted(case["expected_rows"])


@pytest.mark.parametrize("dtype", ["int64", "int32"])
def test_partitioned_probe_matches_direct_probe(gx, dtype):
    """Large probes against tables beyond the L2s take the partitioned probe (radix-partition on the
    table's top hash bits, XCD-affine probe); it must produce the same pair multiset as the direct
    probe and as the oracle, incl. duplicate build keys and left-outer rows."""
    import ctypes
    import torch
    Column, ops = gx
    from cudf_amd import _lib as L
    from cudf_amd.column import device_bytes, ptr, stream_ptr
    rng = np.random.default_rng(21)
    nb, npr = 1_200_000, (1 << 22) + 777
    build = rng.permutation(3 * nb)[:nb].astype(dtype)
    build[:1000] = build[5000:6000]                      # duplicate build keys
    probe = rng.integers(0, 4 * nb, npr).astype(dtype)
    hj = ops.HashJoin(Column.from_numpy(build))
    assert L.lib.gx_join_partition_bits(hj.key_size, hj.table_bytes) >= 3
    pc = Column.from_numpy(probe)
    l, r = hj.inner_join(pc)                              # partitioned (n >= 2^22, no nulls)
    gl, gr = _pairs(l, r)
    el, er = orc.inner_join(probe, build)
    np.testing.assert_array_equal(gl, el)
    np.testing.assert_array_equal(gr, er)
    # direct probe of the same table, for the left-outer form
    old = ops.HashJoin.PARTITIONED_MIN_ROWS
    try:
        ops.HashJoin.PARTITIONED_MIN_ROWS = 1 << 62
        dl, dr = hj.left_join(pc)
    finally:
        ops.HashJoin.PARTITIONED_MIN_ROWS = old
    pl, pr = hj.left_join(pc)
    a = _pairs(dl, dr)
    b = _pairs(pl, pr)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


@pytest.mark.parametrize("dtype", ["int64", "int32"])
def test_partitioned_probe_chains_cross_subtables(gx, dtype):
    """The partitioned probe walks collision chains on 4-bit tags held in LDS, one 2^17-slot sub-table
    per workgroup.  Build keys that hash to the LAST slots of a sub-table, each duplicated many times,
    force chains across the sub-table edge (and around the end of the table): same pairs as the oracle."""
    Column, ops = gx
    from cudf_amd import _lib as L
    rng = np.random.default_rng(33)
    nb, npr = 1_500_000, (1 << 22) + 5
    ksz = np.dtype(dtype).itemsize
    tb = L.lib.gx_join_table_bytes(ksz, nb, 0.5)
    lg = int(np.log2((tb - 256) // (16 if ksz == 8 else 8)))
    cand = np.arange(1, 3_000_000, dtype=np.uint64)
    with np.errstate(over="ignore"):
        slot = (cand * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(64 - lg)
    edge = cand[(slot & np.uint64((1 << 17) - 1)) >= np.uint64((1 << 17) - 3)]     # last 3 slots of a sub-table
    last = cand[slot >= np.uint64((1 << lg) - 2)]                                  # last slots of the table
    hot = np.concatenate([edge[:40], last[:2]]).astype(dtype)
    assert hot.size >= 20
    base = rng.permutation(3_000_000)[: nb - 60 * hot.size].astype(dtype)
    build = np.concatenate([base, np.repeat(hot, 60)])
    rng.shuffle(build)
    probe = rng.integers(0, 4_000_000, npr).astype(dtype)
    probe[:: 1000] = hot[np.arange(probe[::1000].size) % hot.size]                 # make sure the hot keys are probed
    hj = ops.HashJoin(Column.from_numpy(build))
    assert hj.table_bytes == tb and L.lib.gx_join_partition_bits(ksz, tb) >= 3
    l, r = hj.inner_join(Column.from_numpy(probe))
    gl, gr = _pairs(l, r)
    el, er = orc.inner_join(probe, build)
    np.testing.assert_array_equal(gl, el)
    np.testing.assert_array_equal(gr, er)


def test_join_large_properties(gx):
    """Size-independent checks at 2e7 x 2e6: every emitted pair has equal keys, count equals the
    oracle-free closed form (distinct build keys, known hit set)."""
    import torch
    Column, ops = gx
    nb, n = 2_000_000, 20_000_000
    bk = Column.empty(np.int64, nb)
    bkt = bk.data[: nb * 8].view(torch.int64)
    bkt.copy_(torch.randperm(nb, device="cuda") * 3 + 1)
    pk = ops.random_column(np.int64, n, seed=5, lo=0, hi=int(nb / 0.3))
    pkt = pk.data[: n * 8].view(torch.int64)
    pkt.mul_(3).add_(1)
    l, r = ops.inner_join(pk, bk)
    li = l.data[: l.size * 4].view(torch.int32).long()
    ri = r.data[: r.size * 4].view(torch.int32).long()
    assert bool((pkt[li] == bkt[ri]).all())
    expected = int((pkt < 3 * nb).sum().item())  # key 3u+1 with u < nb is present exactly once
    assert l.size == expected
    assert int(torch.unique(li).numel()) == expected  # each probe row at most once


@pytest.fixture(params=[1, 2], ids=["global_table", "lds_partitioned"])
def gb_algo(request):
    """Both groupby kernels families must give the same answers: 1 = global-atomic table,
    2 = hash-partition + per-partition LDS tables (forced for every n > 0)."""
    from cudf_amd import _lib
    _lib.lib.gx_groupby_set_algorithm(request.param, 1)
    yield request.param
    _lib.lib.gx_groupby_set_algorithm(0, 1)


@pytest.mark.parametrize("vdtype", ["float64", "float32", "int32", "int64"])
@pytest.mark.parametrize("kdtype", ["int32", "int64"])
def test_groupby_sum_count_matches_oracle(gx, kdtype, vdtype, gb_algo):
    Column, ops = gx
    rng = np.random.default_rng(4)
    for n, g in [(0, 1), (1, 1), (1000, 7), (200_000, 1000), (300_000, 150_000)]:
        keys = rng.integers(-g // 2, g // 2 + 1, n).astype(kdtype)
        if n > 10:
            keys[:3] = -1  # the reserved-slot key of the int64 table
        if np.dtype(vdtype).kind == "f":
            vals = (rng.random(n) * 1000 - 300).astype(vdtype)
        else:
            vals = rng.integers(-2**31, 2**31 - 1, n).astype(vdtype)
        kv = rng.random(n) > 0.05
        vv = rng.random(n) > 0.2
        k, s, cv, ca = ops.groupby_sum_count(Column.from_numpy(keys, kv), Column.from_numpy(vals, vv), max_groups_hint=64)
        o = np.argsort(k.to_numpy(), kind="stable")
        ek, res = orc.groupby_agg(keys, vals, ["sum", "count_valid", "count_all"], kv, vv)
        np.testing.assert_array_equal(k.to_numpy()[o], ek)
        np.testing.assert_array_equal(cv.to_numpy()[o], res["count_valid"][0])
        np.testing.assert_array_equal(ca.to_numpy()[o], res["count_all"][0])
        es, ev = res["sum"]
        got = s.to_numpy()[o]
        if np.dtype(vdtype).kind == "f":
            assert got.dtype == np.dtype(vdtype)
            if vdtype == "float64":
                assert np.all(orc.ulp_diff(got[ev], es[ev]) <= 1), "f64 SUM must be within 1 ulp of the exact sum"
            else:
                np.testing.assert_allclose(got[ev], es[ev], rtol=2e-7)
        else:
            assert got.dtype == np.int64
            np.testing.assert_array_equal(got[ev], es[ev])


@pytest.mark.parametrize("nsplit", [1, 3])
@pytest.mark.parametrize("nulls", [False, True])
def test_groupby_partitioned_large(gx, nulls, nsplit):
    """The auto path at sizes where the LDS-partitioned kernels run (n >= 2^19): few groups (every
    partition fits its LDS table), ~8k groups per partition (LDS tables fill up: rows spill to the
    global table and the partial sums merge), the all-ones key (dedicated LDS slot) and a hot key."""
    Column, ops = gx
    from cudf_amd import _lib
    _lib.lib.gx_groupby_set_algorithm(0, nsplit)
    try:
        rng = np.random.default_rng(11)
        for n, g, kdtype in [(1_500_000, 50_000, "int32"), (2_500_000, 2_000_000, "int32"), (1_200_000, 300_000, "int64")]:
            keys = rng.integers(-g // 2, g // 2, n).astype(kdtype)
            keys[::97] = -1
            keys[5::3] = 12345 if n < 2_000_000 else keys[5::3]
            vals = rng.random(n) * 2000.0 - 700.0
            kv = (rng.random(n) > 0.03) if nulls else None
            vv = (rng.random(n) > 0.15) if nulls else None
            k, s, cv, ca = ops.groupby_sum_count(Column.from_numpy(keys, kv), Column.from_numpy(vals, vv),
                                                 max_groups_hint=1 << 21)
            o = np.argsort(k.to_numpy(), kind="stable")
            ek, res = orc.groupby_agg(keys, vals, ["sum", "count_valid", "count_all"], kv, vv)
            np.testing.assert_array_equal(k.to_numpy()[o], ek)
            np.testing.assert_array_equal(cv.to_numpy()[o], res["count_valid"][0])
            np.testing.assert_array_equal(ca.to_numpy()[o], res["count_all"][0])
            es, ev = res["sum"]
            assert np.all(orc.ulp_diff(s.to_numpy()[o][ev], es[ev]) <= 1)
        # integer values: exact, wrapping int64 sums
        n = 1_000_000
        keys = rng.integers(0, 70_000, n).astype(np.int32)
        vals = rng.integers(-2**62, 2**62, n).astype(np.int64)
        k, s, cv, ca = ops.groupby_sum_count(Column.from_numpy(keys), Column.from_numpy(vals))
        o = np.argsort(k.to_numpy(), kind="stable")
        ek, res = orc.groupby_agg(keys, vals, ["sum", "count_valid"])
        np.testing.assert_array_equal(k.to_numpy()[o], ek)
        np.testing.assert_array_equal(s.to_numpy()[o], res["sum"][0])
        np.testing.assert_array_equal(cv.to_numpy()[o], res["count_valid"][0])
    finally:
        _lib.lib.gx_groupby_set_algorithm(0, 1)


@pytest.mark.parametrize("vdtype", ["float64", "float32", "int32", "int64", "uint8", "int16"])
@pytest.mark.parametrize("kdtype", ["int32", "int64"])
def test_groupby_min_max_matches_oracle(gx, kdtype, vdtype, gb_algo):
    Column, ops = gx
    rng = np.random.default_rng(14)
    for n, g in [(0, 1), (1, 1), (1000, 7), (250_000, 3000)]:
        keys = rng.integers(-g // 2, g // 2 + 1, n).astype(kdtype)
        if np.dtype(vdtype).kind == "f":
            vals = (rng.random(n) * 2000 - 1000).astype(vdtype)
            if n > 100:
                vals[::50] = -0.0
                vals[1::97] = np.inf
        else:
            info = np.iinfo(vdtype)
            vals = rng.integers(info.min, info.max, n, dtype=vdtype, endpoint=True)
        kv = rng.random(n) > 0.05
        vv = rng.random(n) > 0.3
        k, mn, mx, cv = ops.groupby_min_max(Column.from_numpy(keys, kv), Column.from_numpy(vals, vv), max_groups_hint=64)
        o = np.argsort(k.to_numpy(), kind="stable")
        ek, res = orc.groupby_agg(keys, vals, ["min", "max", "count_valid"], kv, vv)
        np.testing.assert_array_equal(k.to_numpy()[o], ek)
        np.testing.assert_array_equal(cv.to_numpy()[o], res["count_valid"][0])
        emn, ev = res["min"]
        emx, _ = res["max"]
        assert mn.dtype == np.dtype(vdtype)
        np.testing.assert_array_equal(mn.to_numpy()[o][ev], emn[ev])   # -0.0 == +0.0 under array_equal
        np.testing.assert_array_equal(mx.to_numpy()[o][ev], emx[ev])


@pytest.mark.parametrize("nsplit", [1, 3])
@pytest.mark.parametrize("nulls", [False, True])
def test_groupby_min_max_partitioned_large(gx, nulls, nsplit):
    """MIN / MAX on the auto path at sizes where the LDS-partitioned kernels run (n >= 2^19): partitions that fit
    their LDS table, partitions that overflow it (rows spill to the global table), the all-ones key, a hot key,
    groups whose values are all null, NaN / -0.0 / inf values."""
    Column, ops = gx
    from cudf_amd import _lib
    _lib.lib.gx_groupby_set_algorithm(0, nsplit)
    try:
        rng = np.random.default_rng(21)
        for n, g, kdtype, vdtype in [(1_500_000, 50_000, "int32", "float64"), (2_500_000, 2_000_000, "int32", "int64"),
                                     (1_200_000, 300_000, "int64", "float32"), (1_000_000, 40_000, "int64", "int16")]:
            keys = rng.integers(-g // 2, g // 2, n).astype(kdtype)
            keys[::97] = -1
            if n < 2_000_000:
                keys[5::3] = 12345
            if np.dtype(vdtype).kind == "f":
                vals = (rng.random(n) * 2000 - 1000).astype(vdtype)
                vals[::50] = -0.0
                vals[1::97] = np.inf
                vals[3::1013] = -np.inf
            else:
                info = np.iinfo(vdtype)
                vals = rng.integers(info.min, info.max, n, dtype=vdtype, endpoint=True)
            kv = (rng.random(n) > 0.03) if nulls else None
            vv = (rng.random(n) > 0.4) if nulls else None
            if nulls:
                vv[keys == 7] = False                      # a group with only null values keeps its slot, count 0
            k, mn, mx, cv = ops.groupby_min_max(Column.from_numpy(keys, kv), Column.from_numpy(vals, vv), max_groups_hint=1 << 21)
            o = np.argsort(k.to_numpy(), kind="stable")
            ek, res = orc.groupby_agg(keys, vals, ["min", "max", "count_valid"], kv, vv)
            np.testing.assert_array_equal(k.to_numpy()[o], ek)
            np.testing.assert_array_equal(cv.to_numpy()[o], res["count_valid"][0])
            emn, ev = res["min"]
            emx, _ = res["max"]
            np.testing.assert_array_equal(mn.to_numpy()[o][ev], emn[ev])
            np.testing.assert_array_equal(mx.to_numpy()[o][ev], emx[ev])
    finally:
        _lib.lib.gx_groupby_set_algorithm(0, 1)


@pytest.mark.parametrize("vdtype", ["int32", "int64", "float64"])
@pytest.mark.parametrize("case", [c for c in gv.GROUPBY if c["agg"] in ("sum", "count_valid", "count_all", "mean")],
                         ids=lambda c: c["name"])
def test_reference_golden_groupby(gx, case, vdtype, gb_algo):
    Column, ops = gx
    vdtype = case.get("vals_dtype", vdtype)
    keys, km = gv.col(case["keys"], "int32", case.get("keys_valid"))
    vals, vm = gv.col(case["vals"], vdtype, case.get("vals_valid"))
    k, s, cv, ca = ops.groupby_sum_count(Column.from_numpy(keys, km), Column.from_numpy(vals, vm))
    o = np.argsort(k.to_numpy(), kind="stable")
    np.testing.assert_array_equal(k.to_numpy()[o], np.array(case["expect_keys"], np.int32))
    ev = np.array(case["expect_valid"], bool)
    exp = np.array(case["expect"])
    cvn = cv.to_numpy()[o]
    if case["agg"] == "sum":
        np.testing.assert_array_equal(cvn > 0, ev)
        np.testing.assert_array_equal(s.to_numpy()[o][ev], exp[ev].astype(s.dtype))
    elif case["agg"] == "count_valid":
        np.testing.assert_array_equal(cvn, exp)
    elif case["agg"] == "count_all":
        np.testing.assert_array_equal(ca.to_numpy()[o], exp)
    else:  # MEAN = SUM / COUNT_VALID in double (hash_compound_agg_finalizer.cu:92-133)
        np.testing.assert_array_equal(cvn > 0, ev)
        mean = s.to_numpy()[o][ev].astype(np.float64) / cvn[ev]
        assert np.all(orc.ulp_diff(mean, exp[ev].astype(np.float64)) <= 1)


@pytest.mark.parametrize("vdtype", ["int32", "int64", "float64"])
@pytest.mark.parametrize("case", gv.GROUPBY_SCAN, ids=lambda c: c["name"])
def test_reference_golden_groupby_scan(gx, case, vdtype):
    Column, ops = gx
    keys, km = gv.col(case["keys"], "int32", case.get("keys_valid"))
    vals, vm = gv.col(case["vals"], vdtype, case.get("vals_valid"))
    keep = np.ones(len(keys), bool) if km is None else km
    kc = Column.from_numpy(keys[keep])
    vc = Column.from_numpy(vals[keep], None if vm is None else vm[keep])
    order = ops.sorted_order(kc)                      # sort_helper.cu:73-118 stable_sorted_order(keys)
    sk, sv = ops.gather(kc, order), ops.gather(vc, order)
    out = ops.groupby_scan(sk, sv, "sum")
    np.testing.assert_array_equal(sk.to_numpy(), np.array(case["expect_keys"], np.int32))
    ev = np.array(case["expect_valid"], bool)
    if sv.mask is not None:
        np.testing.assert_array_equal(sv.valid_numpy(), ev)
    np.testing.assert_array_equal(out.to_numpy()[ev], np.array(case["expect"])[ev].astype(out.dtype))


def test_groupby_scan_matches_oracle(gx):
    Column, ops = gx
    rng = np.random.default_rng(8)
    n = 150_000
    keys = rng.integers(0, 300, n).astype(np.int64)
    for vdtype in ("int32", "float64"):
        vals = (rng.random(n) * 100).astype(vdtype)
        vv = rng.random(n) > 0.1
        kc = Column.from_numpy(keys)
        order = ops.sorted_order(kc)
        sk, sv = ops.gather(kc, order), ops.gather(Column.from_numpy(vals, vv), order)
        out = ops.groupby_scan(sk, sv, "sum").to_numpy()
        ek, eo, ev = orc.groupby_scan_sum(keys, vals, None, vv)
        np.testing.assert_array_equal(sk.to_numpy(), ek)
        if vdtype == "int32":
            np.testing.assert_array_equal(out[ev], eo[ev])
        else:
            np.testing.assert_allclose(out[ev], eo[ev], rtol=1e-12)


@pytest.mark.parametrize("spec", [2, 0])
@pytest.mark.parametrize("shape", ["uniform", "hot_key", "one_partition"])
@pytest.mark.parametrize("nulls", [False, True])
@pytest.mark.parametrize("pbits", [9, 8])
def test_groupby_speculative_partition_pass(gx, spec, shape, nulls, pbits):
    """Round 3: the partition pass of the LDS-partitioned groupby runs without its histogram into padded (partition, XCD range)
    slots (spec = 2 forces that at this size; by default it starts at 2^22 rows) and falls back ON THE DEVICE to the exact
    histogram path when a slot overflows: uniform keys (speculation holds), one hot key and keys confined to one partition
    (it must not), SUM / COUNT and MIN / MAX, with and without nulls -- all against the oracle."""
    Column, ops = gx
    from cudf_amd import _lib
    rng = np.random.default_rng(5)
    n = 1_300_003
    keys = rng.integers(-40_000, 40_000, n).astype(np.int64)
    if shape == "hot_key":
        keys[rng.random(n) < 0.3] = 7
    elif shape == "one_partition":  # keys whose partition hash agrees in its top 8 bits
        cand = np.arange(1, 3_000_000, dtype=np.uint64)
        with np.errstate(over="ignore"):
            part = (cand * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(56)
        pool = cand[part == 5][:3000].astype(np.int64)
        keys = pool[rng.integers(0, len(pool), n)]
    vals = rng.random(n) * 300.0 - 100.0
    kv = (rng.random(n) > 0.04) if nulls else None
    vv = (rng.random(n) > 0.2) if nulls else None
    _lib.lib.gx_groupby_set_partition_mode(spec)
    assert _lib.lib.gx_groupby_set_partition_bits(pbits) == 0  # 512 partitions (default since round 3) / the round-2 layout
    try:
        kc, vc = Column.from_numpy(keys, kv), Column.from_numpy(vals, vv)
        k, s, cv, ca = ops.groupby_sum_count(kc, vc)
        o = np.argsort(k.to_numpy(), kind="stable")
        ek, res = orc.groupby_agg(keys, vals, ["sum", "count_valid", "count_all", "min", "max"], kv, vv)
        np.testing.assert_array_equal(k.to_numpy()[o], ek)
        np.testing.assert_array_equal(cv.to_numpy()[o], res["count_valid"][0])
        np.testing.assert_array_equal(ca.to_numpy()[o], res["count_all"][0])
        ev = res["sum"][1]
        assert np.all(orc.ulp_diff(s.to_numpy()[o][ev], res["sum"][0][ev]) <= 1)
        k2, mn, mx, cv2 = ops.groupby_min_max(kc, vc)
        o2 = np.argsort(k2.to_numpy(), kind="stable")
        np.testing.assert_array_equal(k2.to_numpy()[o2], ek)
        em = res["min"][1]
        np.testing.assert_array_equal(mn.to_numpy()[o2][em], res["min"][0][em])
        np.testing.assert_array_equal(mx.to_numpy()[o2][em], res["max"][0][em])
    finally:
        _lib.lib.gx_groupby_set_partition_mode(1)
        _lib.lib.gx_groupby_set_partition_bits(0)   # the default: 512 partitions, 256 chosen per call for dense ids


@pytest.mark.parametrize("keys_kind", ["dense_ids", "sparse_ids", "dense_int32", "loose_bound"])
def test_groupby_partition_bits_chosen_on_the_device(gx, keys_kind):
    """Round 4: in the default mode k_slot_plan picks 256 partitions for DENSE ids (sampled keys below 2 x max_groups, max_groups a
    real bound) and 512 otherwise; the sample's 512 bins fold pairwise, scatter / aggregate / the gated exact pass read the bits
    from the call's plan.  Results against the oracle for ids that take the 8-bit layout, sparse ids that keep 9 bits, and a
    max_groups that is no bound at all (no choice is made); one hot key on top makes a slot overflow, so the exact pass runs on
    the plan's bits as well."""
    Column, ops = gx
    from cudf_amd import _lib
    rng = np.random.default_rng(17)
    n = 5_000_011                                        # >= 2^22: the speculative pass (and with it the sample) runs by default
    ngroups = 200_000
    if keys_kind == "sparse_ids":
        ids = rng.integers(0, 2**62, ngroups, dtype=np.int64)
        keys = ids[rng.integers(0, ngroups, n)]
    elif keys_kind == "dense_int32":
        keys = rng.integers(0, ngroups, n).astype(np.int32)
    else:
        keys = rng.integers(0, ngroups, n).astype(np.int64)
    keys[rng.random(n) < 0.25] = keys[11]                # a hot key: its slot overflows -> the gated exact pass
    vals = rng.integers(-1000, 1000, n).astype(np.float64)   # integer-valued: sums exact in any order
    hint = n if keys_kind == "loose_bound" else 1 << 18
    _lib.lib.gx_groupby_set_partition_bits(0)
    k, s, cv, ca = ops.groupby_sum_count(Column.from_numpy(keys), Column.from_numpy(vals), max_groups_hint=hint)
    o = np.argsort(k.to_numpy(), kind="stable")
    ek, res = orc.groupby_agg(keys, vals, ["sum", "count_valid"])
    np.testing.assert_array_equal(k.to_numpy()[o], ek)
    np.testing.assert_array_equal(cv.to_numpy()[o], res["count_valid"][0])
    np.testing.assert_array_equal(s.to_numpy()[o], res["sum"][0])


def _plan_info(ops, L, tmp, max_groups):
    import ctypes
    info = (ctypes.c_int32 * 4)()
    L.check(L.lib.gx_groupby_plan_info(ops.ptr(tmp), max_groups, info, ops.stream_ptr()), "gx_groupby_plan_info")
    return list(info)


def _groupby_raw(Column, ops, L, keys, vals, kv=None, max_groups=1 << 20):
    """gx_groupby_sum_count through the C ABI with the scratch kept, so that the path the plan chose can be read back"""
    from cudf_amd.ops import _run, _dev_i64
    kc, vc = Column.from_numpy(keys, kv), Column.from_numpy(vals)
    sum_dt = np.float64 if vals.dtype.kind == "f" else np.int64
    ok, osum = Column.empty(keys.dtype, max_groups), Column.empty(sum_dt, max_groups)
    ocv, oca = Column.empty(np.int32, max_groups), Column.empty(np.int32, max_groups)
    ng = _dev_i64()
    tmp = _run(L.lib.gx_groupby_sum_count, kc.gx, kc.data_ptr, kc.mask_ptr if kv is not None else None, vc.gx, vc.data_ptr, None, keys.size, max_groups,
               ok.data_ptr, osum.data_ptr, ocv.data_ptr, oca.data_ptr, ops.ptr(ng))
    g = int(ng.item())
    assert 0 <= g <= max_groups
    for c in (ok, osum, ocv, oca):
        c.size = g
    return ok.to_numpy(), osum.to_numpy(), ocv.to_numpy(), oca.to_numpy(), _plan_info(ops, L, tmp, max_groups)


DENSE_CASES = ["int32_from_zero", "int64_negative_offset", "uint32_high", "zipf_sizes", "wide_span", "few_ids", "key_nulls", "int64_values"]


@pytest.mark.parametrize("case", DENSE_CASES)
def test_groupby_dense_ids_by_direct_address(gx, case):
    """Round 6: dense ids take id-RANGE partitions, travel as (value, 16-bit remainder) and are aggregated in an LDS
    table indexed by the remainder (PartPlan::dense).  Against the oracle, with the path read back: ids from zero (BASELINE config 4's
    shape), a negative offset in int64, uint32 ids near 2^32, Zipf-like group sizes (uneven partitions: the slots come from the sample),
    a span close to the 256 x 7680 limit, a handful of ids, null keys, int64 values (exact sums)."""
    Column, ops = gx
    from cudf_amd import _lib as L
    rng = np.random.default_rng(abs(hash(case)) % 997)
    n = 6_000_017
    vals = rng.random(n) * 200.0 - 100.0
    kv = None
    if case == "int32_from_zero":
        keys = rng.integers(0, 1_000_000, n).astype(np.int32)
    elif case == "int64_negative_offset":
        keys = rng.integers(-700_000, 300_000, n).astype(np.int64) - 5_000_000_000
    elif case == "uint32_high":
        keys = (rng.integers(0, 500_000, n) + (2**32 - 600_000)).astype(np.uint32)
    elif case == "zipf_sizes":
        u = np.maximum(rng.random(n), 1e-12)
        keys = np.minimum(np.floor(u ** -1.2), 900_000).astype(np.int32)
    elif case == "wide_span":
        keys = (rng.integers(0, 950_000, n) * 2).astype(np.int64)          # span 1.9e6 of 1.966e6: ids per partition close to the limit
    elif case == "few_ids":
        keys = rng.integers(10, 17, n).astype(np.int32)
    elif case == "key_nulls":
        keys = rng.integers(0, 300_000, n).astype(np.int32)
        kv = rng.random(n) > 0.05
    else:
        keys = rng.integers(0, 1_000_000, n).astype(np.int32)
        vals = rng.integers(-2**40, 2**40, n).astype(np.int64)
    k, s, cv, ca, info = _groupby_raw(Column, ops, L, keys, vals, kv)
    assert info[0] == 1 and info[1] == 0, f"{case}: the dense path was not taken / fell back ({info})"
    o = np.argsort(k, kind="stable")
    ek, res = orc.groupby_agg(keys, vals, ["sum", "count_valid", "count_all"], kv, None)
    np.testing.assert_array_equal(k[o], ek)
    np.testing.assert_array_equal(cv[o], res["count_valid"][0])
    np.testing.assert_array_equal(ca[o], res["count_all"][0])
    if vals.dtype.kind == "f":
        assert np.all(orc.ulp_diff(s[o], res["sum"][0]) <= 1)
    else:
        np.testing.assert_array_equal(s[o], res["sum"][0])


@pytest.mark.parametrize("case", ["outliers_the_sample_misses", "span_too_wide", "knob_off", "value_nulls"])
def test_groupby_dense_path_declines_or_falls_back(gx, case):
    """keys OUTSIDE the planned id range (a few rows far away that the 1-in-`stride` sample does not see) raise the overflow flag in the
    dense scatter and the exact hash sequence produces the result; a span beyond 256 x 7680 ids, the knob off, value nulls: the hash
    path from the start.  Same results either way."""
    Column, ops = gx
    from cudf_amd import _lib as L
    rng = np.random.default_rng(3)
    n = 9_000_011 if case == "outliers_the_sample_misses" else 6_000_017   # (from 2^23 rows the sample takes every other 64-row chunk)
    keys = rng.integers(0, 400_000, n).astype(np.int64)
    vals = rng.integers(-1000, 1000, n).astype(np.float64)
    vv = None
    if case == "outliers_the_sample_misses":
        pos = 64 + 128 * rng.integers(0, n // 128 - 1, 7) + rng.integers(0, 64, 7)   # rows 64 - 127 of a 128-row step: never sampled
        keys[pos] = 9_000_000_000 + np.arange(7)
    elif case == "span_too_wide":
        keys = keys * 7
    elif case == "value_nulls":
        vv = rng.random(n) > 0.1
    L.lib.gx_groupby_set_dense(0 if case == "knob_off" else 1)
    try:
        if vv is None:
            k, s, cv, ca, info = _groupby_raw(Column, ops, L, keys, vals)
            if case == "outliers_the_sample_misses":
                assert info[0] == 1 and info[1] == 1, info     # planned dense, a row outside the range, the exact sequence ran
            else:
                assert info[0] == 0, info
        else:
            kk, ss, cvv, caa = ops.groupby_sum_count(Column.from_numpy(keys), Column.from_numpy(vals, vv))
            k, s, cv, ca = kk.to_numpy(), ss.to_numpy(), cvv.to_numpy(), caa.to_numpy()
    finally:
        L.lib.gx_groupby_set_dense(1)
    o = np.argsort(k, kind="stable")
    ek, res = orc.groupby_agg(keys, vals, ["sum", "count_valid", "count_all"], None, vv)
    np.testing.assert_array_equal(k[o], ek)
    np.testing.assert_array_equal(cv[o], res["count_valid"][0])
    np.testing.assert_array_equal(ca[o], res["count_all"][0])
    ev = res["sum"][1] if vv is not None else np.ones(len(ek), bool)
    np.testing.assert_array_equal(s[o][ev], res["sum"][0][ev])
