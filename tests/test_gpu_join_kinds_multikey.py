"""GPU parity: semi / anti / distinct joins, row-key encoding (pack / dense rank) and multi-column join /
groupby keys through the C ABI, vs the CPU oracle and the reference's golden vectors."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cudf_oracle as orc
from tests.golden import reference_vectors as gv

NO_MATCH = gv.NO_MATCH


@pytest.fixture(scope="module")
def gx():
    import torch
    assert torch.cuda.is_available()
    import cudf_amd
    from cudf_amd import Column, ops
    return Column, ops


def _cols(Column, lists, dtype):
    out = []
    for c in lists:
        a, m = gv.col(c, dtype)
        out.append(Column.from_numpy(a, m))
    return out


def _np_cols(lists, dtype):
    cols, masks = [], []
    for c in lists:
        a, m = gv.col(c, dtype)
        cols.append(a)
        masks.append(m)
    return cols, masks


# ------------------------------------------------------------------------------------------------
# semi / anti / distinct
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", gv.SEMI_ANTI, ids=lambda c: c["name"])
def test_reference_golden_semi_anti(gx, case):
    Column, ops = gx
    (l,), (r,) = _cols(Column, case["left"], case["dtype"]), _cols(Column, case["right"], case["dtype"])
    fn = ops.left_semi_join if case["how"] == "semi" else ops.left_anti_join
    got = fn(l, r, case["nulls_equal"])
    assert got.dtype == np.int32 and got.to_numpy().tolist() == case["expected"]


@pytest.mark.parametrize("dtype", ["int64", "int32"])
@pytest.mark.parametrize("nulls_equal", [True, False])
def test_semi_anti_match_oracle(gx, dtype, nulls_equal):
    """Ascending left rows with / without a match; null probe rows match only a null build row and only
    under null_equality::EQUAL (filtered_join.cu:124-156)."""
    Column, ops = gx
    rng = np.random.default_rng(5)
    for nl, nr, nulls in [(1, 1, False), (5000, 700, False), (5000, 700, True), (300_001, 50_000, True), (4097, 3, False)]:
        right = rng.integers(0, nr * 2 + 1, nr).astype(dtype)
        left = rng.integers(0, nr * 4 + 1, nl).astype(dtype)
        lv = rng.random(nl) > 0.2 if nulls else None
        rv = rng.random(nr) > 0.2 if nulls else None
        hj = ops.HashJoin(Column.from_numpy(right, rv), nulls_equal)
        lc = Column.from_numpy(left, lv)
        semi, anti = hj.semi_join(lc).to_numpy(), hj.anti_join(lc).to_numpy()
        es = orc.semi_join([left], [right], [lv], [rv], nulls_equal)
        ea = orc.anti_join([left], [right], [lv], [rv], nulls_equal)
        np.testing.assert_array_equal(semi, es)
        np.testing.assert_array_equal(anti, ea)
        assert len(semi) + len(anti) == nl


def test_semi_anti_large_properties(gx):
    """2e7 probe rows: semi + anti partition the rows, both ascending, semi keys all present."""
    import torch
    Column, ops = gx
    nb, n = 1_000_000, 20_000_000
    bk = Column.empty(np.int64, nb)
    bkt = bk.data[: nb * 8].view(torch.int64)
    bkt.copy_(torch.randperm(nb, device="cuda") * 2)          # even keys 0 .. 2nb-2
    pk = ops.random_column(np.int64, n, seed=9, lo=0, hi=4 * nb)
    pkt = pk.data[: n * 8].view(torch.int64)
    hj = ops.HashJoin(bk)
    semi, anti = hj.semi_join(pk), hj.anti_join(pk)
    st = semi.data[: semi.size * 4].view(torch.int32).long()
    at = anti.data[: anti.size * 4].view(torch.int32).long()
    assert semi.size + anti.size == n
    assert bool((st[1:] > st[:-1]).all()) and bool((at[1:] > at[:-1]).all())
    hit = (pkt % 2 == 0) & (pkt < 2 * nb)
    assert semi.size == int(hit.sum().item())
    assert bool(hit[st].all()) and not bool(hit[at].any())


@pytest.mark.parametrize("case", gv.DISTINCT_JOIN, ids=lambda c: c["name"])
def test_reference_golden_distinct_join(gx, case):
    Column, ops = gx
    left, right = _cols(Column, case["left"], case["dtype"]), _cols(Column, case["right"], case["dtype"])
    if len(left) > 1:
        lk, rk = ops.encode_rows([left, right], True)
    else:
        lk, rk = left[0], right[0]
    hj = ops.HashJoin(rk)
    if case["how"] == "inner":
        l, r = hj.inner_join(lk)
        assert sorted(zip(l.to_numpy().tolist(), r.to_numpy().tolist())) == sorted(case["expected_pairs"])
    else:
        assert hj.lookup(lk).to_numpy().tolist() == case["expected"]


def test_lookup_matches_oracle(gx):
    Column, ops = gx
    rng = np.random.default_rng(6)
    right = rng.permutation(400_000)[:100_000].astype(np.int64)      # distinct
    left = rng.integers(0, 400_000, 250_001).astype(np.int64)
    lv = rng.random(left.size) > 0.1
    rv = np.ones(right.size, bool)
    rv[17] = False                                                  # one null build row
    for eq in (True, False):
        hj = ops.HashJoin(Column.from_numpy(right, rv), eq)
        got = hj.lookup(Column.from_numpy(left, lv)).to_numpy()
        np.testing.assert_array_equal(got, orc.distinct_left_join([left], [right], [lv], [rv], eq))


# ------------------------------------------------------------------------------------------------
# building blocks: bitmask copy, pack, dense rank
# ------------------------------------------------------------------------------------------------
def test_bitmask_copy(gx):
    import torch
    from cudf_amd import _lib as L
    from cudf_amd.column import pack_mask, unpack_mask, ptr, stream_ptr
    rng = np.random.default_rng(7)
    for nbits, doff, soff in [(1, 0, 0), (31, 1, 0), (32, 0, 0), (33, 31, 5), (1000, 77, 13), (100_003, 4096, 31), (64, 32, 32), (5, 30, 29)]:
        src = rng.random(soff + nbits) > 0.5
        dst = rng.random(doff + nbits + 70) > 0.5
        d = torch.from_numpy(pack_mask(dst).view(np.int32).copy()).cuda()
        s = torch.from_numpy(pack_mask(src).view(np.int32).copy()).cuda()
        assert L.lib.gx_bitmask_copy(ptr(d), doff, ptr(s), soff, nbits, stream_ptr()) == 0
        exp = dst.copy()
        exp[doff:doff + nbits] = src[soff:soff + nbits]
        np.testing.assert_array_equal(unpack_mask(d.cpu().numpy().view(np.uint32), len(dst)), exp)
        # src == NULL writes ones
        assert L.lib.gx_bitmask_copy(ptr(d), doff, None, 0, nbits, stream_ptr()) == 0
        exp[doff:doff + nbits] = True
        np.testing.assert_array_equal(unpack_mask(d.cpu().numpy().view(np.uint32), len(dst)), exp)


def test_pack_keys(gx):
    Column, ops = gx
    rng = np.random.default_rng(8)
    n = 100_003
    a = rng.integers(-2**31, 2**31, n).astype(np.int32)
    b = rng.integers(0, 2**16, n).astype(np.uint16)
    c = rng.integers(-128, 128, n).astype(np.int8)
    got = ops.pack_keys([Column.from_numpy(a), Column.from_numpy(b), Column.from_numpy(c)]).to_numpy()
    exp = (a.view(np.uint32).astype(np.uint64) << np.uint64(24)) | (b.astype(np.uint64) << np.uint64(8)) | c.view(np.uint8).astype(np.uint64)
    np.testing.assert_array_equal(got, exp)
    # floats are normalised: -0.0 == +0.0, every NaN == NaN
    f = np.array([0.0, -0.0, np.nan, -np.nan, 1.5, np.float32("inf")], np.float32)
    g = ops.pack_keys([Column.from_numpy(f), Column.from_numpy(np.arange(6, dtype=np.int32) * 0)]).to_numpy()
    assert g[0] == g[1] and g[2] == g[3] and len(set(g.tolist())) == 4
    from cudf_amd import _lib as L
    cols = [Column.from_numpy(np.zeros(4, np.int64)), Column.from_numpy(np.zeros(4, np.int32))]
    with pytest.raises(L.GxError):
        ops.pack_keys(cols)                                         # 12 bytes do not fit


@pytest.mark.parametrize("dtype", ["int64", "int32", "uint64", "float64", "float32", "int16", "uint8"])
def test_dense_rank(gx, dtype):
    Column, ops = gx
    rng = np.random.default_rng(9)
    big = [((1 << 22) + 3, False)] if dtype in ("int64", "float64") else []   # >= 2^22: the hybrid sort path
    for n, nulls in [(1, False), (1000, False), (1000, True), (200_003, True)] + big:
        if np.dtype(dtype).kind == "f":
            v = rng.integers(-50, 50, n).astype(dtype) / 4
            if n > 10:
                v[3], v[7], v[8], v[9] = np.nan, -np.nan, 0.0, -0.0
        else:
            info = np.iinfo(dtype)
            v = rng.integers(max(info.min, -300), min(info.max, 300) + 1, n).astype(dtype)
        valid = rng.random(n) > 0.15 if nulls else None
        ids, rep, g = ops.dense_rank(Column.from_numpy(v, valid))
        ids, rep = ids.to_numpy(), rep.to_numpy()
        ok = np.ones(n, bool) if valid is None else valid
        if np.dtype(dtype).kind == "f":                              # equality classes: NaN == NaN, -0.0 == +0.0
            canon = v.copy()
            canon[np.isnan(canon)] = np.nan
            canon[canon == 0] = 0.0
            sk = orc.sortable_bits(canon)                             # ascending, NaN after +Inf
        else:
            sk = v
        uniq, inv = np.unique(sk[ok], return_inverse=True)
        exp = np.full(n, len(uniq), np.int64)                         # nulls: one id after every value
        exp[ok] = inv.reshape(-1)
        np.testing.assert_array_equal(ids, exp.astype(np.int32))
        assert g == len(uniq) + (0 if ok.all() else 1)
        first = np.full(g, n, np.int64)                               # smallest row of every id
        np.minimum.at(first, exp, np.arange(n))
        np.testing.assert_array_equal(rep, first.astype(np.int32))


# ------------------------------------------------------------------------------------------------
# multi-column keys
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [c for c in gv.JOIN if c.get("how", "inner") in ("inner", "left")], ids=lambda c: c["name"])
def test_reference_golden_join_tables(gx, case):
    """Every inner / left golden of join_tests.cpp, single- and multi-column keys, through the row encoding."""
    Column, ops = gx
    fn = ops.inner_join_tables if case.get("how", "inner") == "inner" else ops.left_join_tables
    for eq in case["nulls_equal"]:
        l, r = fn(_cols(Column, case["left"], case["dtype"]), _cols(Column, case["right"], case["dtype"]), eq)
        l, r = l.to_numpy(), r.to_numpy()
        if "expected_size" in case:
            assert len(l) == case["expected_size"]
        if "expected_pairs" in case:
            assert sorted(zip(l.tolist(), r.tolist())) == sorted(case["expected_pairs"])
        else:
            lp = [np.array(c) for c in case["left_payload"]]
            rp = [np.array(c) for c in case["right_payload"]]
            rows = sorted(tuple(int(c[i]) for c in lp) + tuple(int(c[j]) for c in rp) for i, j in zip(l, r))
            assert rows == sorted(case["expected_rows"])


@pytest.mark.parametrize("schema", [("int32", "int32"), ("int64", "int32"), ("int64", "int64", "int64"),
                                    ("int16", "uint8", "int32"), ("float64", "int32"), ("int32", "float32", "int64")],
                         ids=lambda s: "-".join(s))
@pytest.mark.parametrize("nulls", [False, True])
@pytest.mark.parametrize("nulls_equal", [True, False])
def test_multi_column_join_matches_oracle(gx, schema, nulls, nulls_equal):
    Column, ops = gx
    rng = np.random.default_rng(10)
    nl, nr = 20_011, 3_001

    def table(n):
        cols, masks = [], []
        for dt in schema:
            if np.dtype(dt).kind == "f":
                v = rng.integers(-3, 4, n).astype(dt) / 2
                v[rng.random(n) < 0.05] = np.nan
                v[rng.random(n) < 0.05] = -0.0
            else:
                v = rng.integers(0, 12, n).astype(dt)
            cols.append(v)
            masks.append(rng.random(n) > 0.1 if nulls else None)
        return cols, masks

    (lc, lm), (rc, rm) = table(nl), table(nr)
    L = [Column.from_numpy(c, m) for c, m in zip(lc, lm)]
    R = [Column.from_numpy(c, m) for c, m in zip(rc, rm)]
    l, r = ops.inner_join_tables(L, R, nulls_equal)
    gl, gr = orc.canonical_pairs(l.to_numpy(), r.to_numpy())
    el, er = orc.inner_join(lc, rc, lm, rm, nulls_equal)
    np.testing.assert_array_equal(gl, el)
    np.testing.assert_array_equal(gr, er)
    l, r = ops.left_join_tables(L, R, nulls_equal)
    gl, gr = orc.canonical_pairs(l.to_numpy(), r.to_numpy())
    el, er = orc.left_join(lc, rc, lm, rm, nulls_equal)
    np.testing.assert_array_equal(gl, el)
    np.testing.assert_array_equal(gr, er)


@pytest.mark.parametrize("schema", [("int32", "int32"), ("int64", "int16"), ("int64", "int64", "int32"), ("float64", "int8")],
                         ids=lambda s: "-".join(s))
@pytest.mark.parametrize("nulls", [False, True])
def test_multi_column_groupby(gx, schema, nulls):
    """groupby on several key columns: ids from the row encoding, SUM / COUNT per id, keys gathered by
    the first row of every id; rows with a null in any key column are dropped (null_policy::EXCLUDE)."""
    Column, ops = gx
    rng = np.random.default_rng(11)
    n = 150_007
    cols, masks = [], []
    for dt in schema:
        v = (rng.integers(-2, 3, n).astype(dt) / 2) if np.dtype(dt).kind == "f" else rng.integers(0, 9, n).astype(dt)
        cols.append(v)
        masks.append(rng.random(n) > 0.05 if nulls else None)
    vals = rng.integers(-1000, 1000, n).astype(np.int64)
    K = [Column.from_numpy(c, m) for c, m in zip(cols, masks)]
    out_keys, s, cv, ca = ops.groupby_sum_count_tables(K, Column.from_numpy(vals))
    got = {}
    ks = [k.to_numpy() for k in out_keys]
    for i in range(s.size):
        got[tuple(float(k[i]) for k in ks)] = (int(s.to_numpy()[i]), int(cv.to_numpy()[i]), int(ca.to_numpy()[i]))
    ok = np.ones(n, bool)
    for m in masks:
        if m is not None:
            ok &= m
    exp = {}
    for i in np.nonzero(ok)[0]:
        key = tuple(float(c[i]) + 0.0 for c in cols)
        a = exp.setdefault(key, [0, 0, 0])
        a[0] += int(vals[i])
        a[1] += 1
        a[2] += 1
    assert got == {k: tuple(v) for k, v in exp.items()}


def test_hashed_row_keys_large_and_collision_fallback(gx, monkeypatch):
    """Rows wider than 8 bytes are keyed by a 64-bit row hash and the result is certified against the key columns
    (gx_hash_rows64 / gx_rows_mismatch_count).  (1) 2 x int64 keys at a few million rows against the oracle, join
    and groupby; (2) the certificate: equal tables have no mismatch, a perturbed row is counted; (3) a reported
    mismatch (a 64-bit collision, simulated) sends both operators down the exact dense-rank path: same answers."""
    Column, ops = gx
    rng = np.random.default_rng(77)
    nl, nr = 3_000_017, 400_003
    ra = rng.integers(-2**62, 2**62, nr, dtype=np.int64)
    rb = rng.integers(0, 5, nr).astype(np.int64)
    pick = rng.integers(0, nr, nl)
    la, lb = ra[pick].copy(), rb[pick].copy()
    miss = rng.random(nl) < 0.6
    lb[miss] += 7                                            # same first column, different second: no match
    L = [Column.from_numpy(la), Column.from_numpy(lb)]
    R = [Column.from_numpy(ra), Column.from_numpy(rb)]
    el, er = orc.inner_join([la, lb], [ra, rb], [None, None], [None, None], True)

    def check_join():
        l, r = ops.inner_join_tables(L, R)
        gl, gr = orc.canonical_pairs(l.to_numpy(), r.to_numpy())
        np.testing.assert_array_equal(gl, el)
        np.testing.assert_array_equal(gr, er)
        l, r = ops.left_join_tables(L, R)
        assert l.size == len(el) + int(np.count_nonzero(~np.isin(np.arange(nl), el)))

    def check_groupby(cols=None):
        vals = rng.integers(-1000, 1000, nl).astype(np.int64)
        keys, s, cv, _ = ops.groupby_sum_count_tables(cols or L, Column.from_numpy(vals))
        ka, kb = keys[0].to_numpy(), keys[1].to_numpy().astype(np.int64)
        o = np.lexsort((kb, ka))
        packed = np.stack([la, lb], 1)
        uk, inv = np.unique(packed, axis=0, return_inverse=True)
        es = np.zeros(len(uk), np.int64)
        np.add.at(es, inv.ravel(), vals)
        np.testing.assert_array_equal(np.stack([ka[o], kb[o]], 1), uk)
        np.testing.assert_array_equal(s.to_numpy()[o], es)
        np.testing.assert_array_equal(cv.to_numpy()[o], np.bincount(inv.ravel(), minlength=len(uk)))

    check_join()
    check_groupby()
    # (2) the certificate itself
    iota = Column.from_numpy(np.arange(nr, dtype=np.int32))
    assert ops.rows_mismatch_count(R, R, None, iota, nr) == 0
    rb2 = rb.copy()
    rb2[[5, 77, nr - 1]] += 1
    assert ops.rows_mismatch_count(R, [R[0], Column.from_numpy(rb2)], None, None, nr) == 3
    skip = np.arange(nr, dtype=np.int32)
    skip[5] = -1                                             # a negative index = no row: not compared
    assert ops.rows_mismatch_count(R, [R[0], Column.from_numpy(rb2)], None, Column.from_numpy(skip), nr) == 2
    h1, h2 = ops.hash_rows64(R).to_numpy(), ops.hash_rows64([R[0], Column.from_numpy(rb2)]).to_numpy()
    assert np.count_nonzero(h1 != h2) == 3 and len(np.unique(h1)) >= len(np.unique(np.stack([ra, rb], 1), axis=0)) - 1
    # (3) simulated collision: the exact path must take over
    calls = []
    real = ops.rows_mismatch_count
    monkeypatch.setattr(ops, "rows_mismatch_count", lambda *a: (calls.append(1), real(*a) + 1)[1])
    check_join()
    assert len(calls) >= 2
    # integer key columns group in one partition pass with the rows compared inside the LDS tables (gx_groupby_sum_count_wide:
    # no hash, no certificate); a float key column still goes through the certified row hash -- with the first answer
    # "collision" the exact dense-rank path must give the same groups
    check_groupby()
    before = len(calls)
    check_groupby([L[0], Column.from_numpy(lb.astype(np.float64))])
    assert len(calls) > before


# ------------------------------------------------------------------------------------------------
# compound groupby aggregations: VARIANCE / STD / M2, ARGMIN / ARGMAX
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("vdtype", ["int32", "int64", "float64"])
@pytest.mark.parametrize("case", [c for c in gv.GROUPBY if c["agg"] in ("var", "std", "argmin", "argmax", "min", "max")],
                         ids=lambda c: c["name"])
def test_reference_golden_groupby_compound(gx, case, vdtype):
    Column, ops = gx
    vdtype = case.get("vals_dtype", vdtype)
    keys, km = gv.col(case["keys"], "int32", case.get("keys_valid"))
    vals, vm = gv.col(case["vals"], vdtype, case.get("vals_valid"))
    K, V = Column.from_numpy(keys, km), Column.from_numpy(vals, vm)
    ev = np.array(case["expect_valid"], bool)
    exp = np.array(case["expect"])
    if case["agg"] in ("var", "std"):
        k, var, std, m2, cv = ops.groupby_var_std(K, V, ddof=case.get("ddof", 1))
        np.testing.assert_array_equal(k.to_numpy(), np.array(case["expect_keys"], np.int32))
        r = var if case["agg"] == "var" else std
        np.testing.assert_array_equal(r.valid_numpy() if r.mask is not None and r.null_count else np.ones(r.size, bool), ev)
        assert r.dtype == np.float64
        assert np.all(orc.ulp_diff(r.to_numpy()[ev], exp[ev].astype(np.float64)) <= 1)
    elif case["agg"] in ("min", "max"):
        k, mn, mx, cv = ops.groupby_min_max(K, V)
        o = np.argsort(k.to_numpy(), kind="stable")
        np.testing.assert_array_equal(k.to_numpy()[o], np.array(case["expect_keys"], np.int32))
        np.testing.assert_array_equal(cv.to_numpy()[o] > 0, ev)
        r = (mn if case["agg"] == "min" else mx).to_numpy()[o]
        assert r.dtype == np.dtype(vdtype)                     # MIN / MAX keep the values' type
        np.testing.assert_array_equal(r[ev], exp[ev].astype(vdtype))
    else:
        k, amin, amax, cv = ops.groupby_argmin_argmax(K, V)
        o = np.argsort(k.to_numpy(), kind="stable")
        np.testing.assert_array_equal(k.to_numpy()[o], np.array(case["expect_keys"], np.int32))
        np.testing.assert_array_equal(cv.to_numpy()[o] > 0, ev)
        r = (amin if case["agg"] == "argmin" else amax).to_numpy()[o]
        assert r.dtype == np.int32
        np.testing.assert_array_equal(r[ev], exp[ev])


@pytest.mark.parametrize("vdtype", ["int8", "int64", "uint32", "float32", "float64"])
@pytest.mark.parametrize("nulls", [False, True])
def test_groupby_compound_matches_oracle(gx, vdtype, nulls):
    """700 k rows (the LDS-partitioned sum path), many ties for ARGMIN / ARGMAX (smallest row wins), NaN and
    -0.0 among the float values; integer VAR is bit-exact (int64 sums, the same double formula), float
    VAR is compared relative to the sum of squares (the formula cancels)."""
    Column, ops = gx
    rng = np.random.default_rng(12)
    n = 700_001
    keys = rng.integers(0, 5000, n).astype(np.int64)
    if np.dtype(vdtype).kind == "f":
        vals = (rng.integers(-40, 40, n) / 8).astype(vdtype)
        vals[rng.random(n) < 0.01] = -0.0
    else:
        info = np.iinfo(vdtype)
        vals = rng.integers(max(info.min, -100), min(info.max, 100) + 1, n).astype(vdtype)
    kv = rng.random(n) > 0.05 if nulls else None
    vv = rng.random(n) > 0.2 if nulls else None
    K, V = Column.from_numpy(keys, kv), Column.from_numpy(vals, vv)
    ek, res = orc.groupby_agg(keys, vals, ["var", "std", "m2", "argmin", "argmax", "count_valid"], kv, vv, exact=False)
    k, var, std, m2, cv = ops.groupby_var_std(K, V)
    np.testing.assert_array_equal(k.to_numpy(), ek)
    np.testing.assert_array_equal(cv.to_numpy(), res["count_valid"][0])
    for got, name in ((var, "var"), (std, "std"), (m2, "m2")):
        e, ev = res[name]
        gvalid = got.valid_numpy() if got.null_count else np.ones(got.size, bool)
        np.testing.assert_array_equal(gvalid, ev)
        if np.dtype(vdtype).kind in "iu":
            np.testing.assert_array_equal(got.to_numpy()[ev], e[ev])
        else:
            np.testing.assert_allclose(got.to_numpy()[ev], e[ev], rtol=1e-9, atol=1e-9)
    k2, amin, amax, cv2 = ops.groupby_argmin_argmax(K, V)
    o = np.argsort(k2.to_numpy(), kind="stable")
    np.testing.assert_array_equal(k2.to_numpy()[o], ek)
    ev = res["argmin"][1]
    np.testing.assert_array_equal(amin.to_numpy()[o][ev], res["argmin"][0][ev])
    np.testing.assert_array_equal(amax.to_numpy()[o][ev], res["argmax"][0][ev])


@pytest.mark.parametrize("dtype", ["int64", "int32", "float64", "float32"])
@pytest.mark.parametrize("nulls_equal", [True, False])
def test_single_key_left_join_with_nulls_on_both_sides(gx, dtype, nulls_equal):
    """null_equality::EQUAL: a null left row matches every null build row and must NOT also appear as
    (row, JoinNoMatch) (cpp/src/join/hash_join/hash_join.cu:77-84); UNEQUAL: it matches nothing.  Float keys:
    -0.0 == +0.0 and NaN == NaN whatever the payload (row_operator/common_utils.cuh:215-220)."""
    Column, ops = gx
    rng = np.random.default_rng(31)
    nl, nr = 10_007, 2_003
    if np.dtype(dtype).kind == "f":
        lv = (rng.integers(-4, 5, nl) / 2).astype(dtype)
        rv = (rng.integers(-4, 5, nr) / 2).astype(dtype)
        lv[rng.random(nl) < 0.05] = np.nan
        rv[rng.random(nr) < 0.05] = -np.nan
        lv[rng.random(nl) < 0.05] = -0.0
    else:
        lv = rng.integers(0, 40, nl).astype(dtype)
        rv = rng.integers(0, 40, nr).astype(dtype)
    lm, rm = rng.random(nl) > 0.1, rng.random(nr) > 0.2
    l, r = ops.left_join(Column.from_numpy(lv, lm), Column.from_numpy(rv, rm), nulls_equal)
    gl, gr = orc.canonical_pairs(l.to_numpy(), r.to_numpy())
    el, er = orc.left_join([lv], [rv], [lm], [rm], nulls_equal)
    np.testing.assert_array_equal(gl, el)
    np.testing.assert_array_equal(gr, er)
    l, r = ops.inner_join(Column.from_numpy(lv, lm), Column.from_numpy(rv, rm), nulls_equal)
    gl, gr = orc.canonical_pairs(l.to_numpy(), r.to_numpy())
    el, er = orc.inner_join([lv], [rv], [lm], [rm], nulls_equal)
    np.testing.assert_array_equal(gl, el)
    np.testing.assert_array_equal(gr, er)
