"""At-scale parity of the round-2 additions of the C++ surface against the NumPy oracle:
cudf::rank, top_k, segmented sorts, the SORT path of groupby::aggregate (SUM / PRODUCT / MIN / MAX / COUNT / NTH,
sorted::YES, null_policy::INCLUDE), groupby::scan (SUM / MIN / MAX / COUNT), get_groups, shift, replace_nulls,
hash_join::*_join_match_context and cudf::full_join (the complement kernel).

Inputs are 1e5 .. 4e6 rows with nulls, single-row groups and groups that span many scan chunks / tiles, so the
chunk-carry and multi-workgroup code of gx_segmented_reduce, gx_group_offsets, gx_rank_from_groups, gx_segment_ids,
gx_segmented_shift, gx_segmented_fill_nulls, gx_join_count_rows and gx_join_complement runs under a checker.  The calls
go cudf:: C++ API (cudf_amd/libcudf.so) -> gx_* C ABI -> HIP kernels through tests/cpp/libcudf_test_shim.so; the
oracle functions are pinned to the reference's literal vectors in tests/test_oracle_golden.py.
"""
import ctypes
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cudf_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIND = dict(sum=0, product=2, min=3, max=4, count_valid=5, count_all=6, nth=19)  # cudf::aggregation::Kind
TID = {np.dtype("int8"): 1, np.dtype("int16"): 2, np.dtype("int32"): 3, np.dtype("int64"): 4, np.dtype("uint8"): 5,
       np.dtype("uint16"): 6, np.dtype("uint32"): 7, np.dtype("uint64"): 8, np.dtype("float32"): 9, np.dtype("float64"): 10,
       np.dtype("bool"): 11}
NPT = {v: k for k, v in TID.items()}


class Dev:
    """a host column on the device: data + optional Arrow validity bitmap"""

    def __init__(self, values, valid=None):
        import torch
        from cudf_amd.column import pack_mask
        v = np.ascontiguousarray(values)
        self.n = len(v)
        self.dtype = v.dtype
        self.t = torch.from_numpy(v.view(np.uint8).reshape(-1).copy()).cuda() if v.nbytes else torch.empty(1, dtype=torch.uint8, device="cuda")
        self.m = None
        self.nulls = 0
        if valid is not None:
            valid = np.asarray(valid, bool)
            self.nulls = int((~valid).sum())
            self.m = torch.from_numpy(pack_mask(valid).view(np.int32).copy()).cuda()

    @property
    def p(self):
        return ctypes.c_void_p(self.t.data_ptr())

    @property
    def mp(self):
        return ctypes.c_void_p(self.m.data_ptr()) if self.m is not None else None

    @property
    def tid(self):
        return TID[self.dtype]


class Out:
    def __init__(self, dtype, n, mask=False):
        import torch
        self.dtype = np.dtype(dtype)
        self.t = torch.zeros(max(1, n * self.dtype.itemsize), dtype=torch.uint8, device="cuda")
        self.m = torch.zeros((n + 31) // 32 + 16, dtype=torch.int32, device="cuda") if mask else None

    @property
    def p(self):
        return ctypes.c_void_p(self.t.data_ptr())

    @property
    def mp(self):
        return ctypes.c_void_p(self.m.data_ptr()) if self.m is not None else None

    def get(self, n, dtype=None):
        dt = np.dtype(dtype or self.dtype)
        return self.t[: n * dt.itemsize].cpu().numpy().view(dt).copy()

    def valid(self, n):
        from cudf_amd.column import unpack_mask
        return unpack_mask(self.m.cpu().numpy().view(np.uint32), n)


@pytest.fixture(scope="module")
def shim():
    import torch
    assert torch.cuda.is_available()
    import __graft_entry__ as ge
    path = os.path.join(ROOT, "tests", "cpp", "libcudf_test_shim.so")
    if not os.path.exists(path):
        ge.build()
    import cudf_amd  # noqa: F401  (loads libcudf_amd.so; the shim resolves it through its rpath as well)
    lib = ctypes.CDLL(path)
    lib.shim_last_error.restype = ctypes.c_char_p

    def call(fn, *args):
        rc = getattr(lib, fn)(*args)
        assert rc == 0, f"{fn}: {lib.shim_last_error().decode()}"
    return call


def _rand_valid(rng, n, p_null):
    return rng.random(n) >= p_null


def _ulp_ok(a, b, ulps=1):
    return bool(np.all(orc.ulp_diff(np.asarray(a, np.float64), np.asarray(b, np.float64)) <= ulps))


# ---------------------------------------------------------------------------------------------------------------------
# rank
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("method", [orc.RANK_FIRST, orc.RANK_AVERAGE, orc.RANK_MIN, orc.RANK_MAX, orc.RANK_DENSE])
@pytest.mark.parametrize("percentage", [False, True])
def test_rank_matches_oracle_at_scale(shim, method, percentage):
    rng = np.random.default_rng(100 + method)
    n = 1_500_003
    cases = [
        (rng.integers(0, 5000, n).astype(np.int32), None, True, False, False),            # long tie groups
        (rng.integers(-2**62, 2**62, n, dtype=np.int64), _rand_valid(rng, n, 0.07), True, False, False),  # unique + keep
        (rng.integers(0, 40, n).astype(np.int64), _rand_valid(rng, n, 0.2), False, True, True),   # desc, nulls ranked
        ((rng.integers(0, 300000, n) * 0.25).astype(np.float64), _rand_valid(rng, n, 0.05), False, False, True),  # desc keep
    ]
    for v, valid, asc, null_include, null_before in cases:
        d = Dev(v, valid)
        as_f64 = percentage or method == orc.RANK_AVERAGE
        out = Out(np.float64 if as_f64 else np.int32, n, mask=True)
        nulls = ctypes.c_int(-1)
        shim("shim_rank", d.tid, d.p, d.mp, n, d.nulls, method, 0 if asc else 1, 1 if null_include else 0,
             1 if null_before else 0, 1 if percentage else 0, out.p, out.mp, ctypes.byref(nulls))
        want, wv = orc.rank(v, valid, method, asc, null_include, null_before, percentage)
        got, gv = out.get(n), out.valid(n)
        np.testing.assert_array_equal(gv, wv)
        assert nulls.value == int((~wv).sum())
        if as_f64:
            assert _ulp_ok(got[wv], want[wv], 1)
        else:
            np.testing.assert_array_equal(got[wv], want[wv])


# ---------------------------------------------------------------------------------------------------------------------
# top_k
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("descending", [True, False])
def test_top_k_matches_oracle_at_scale(shim, descending):
    rng = np.random.default_rng(7)
    n = 2_000_000
    for v, valid in [(rng.integers(-1000, 1000, n).astype(np.int32), None),
                     (rng.standard_normal(n), _rand_valid(rng, n, 0.3)),
                     (rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64), None)]:
        d = Dev(v, valid)
        for k in (1, 1000, 333_333, n - 1, n, n + 5):
            ov, oi = Out(v.dtype, n), Out(np.int32, n)
            cnt = ctypes.c_int(-1)
            shim("shim_top_k", d.tid, d.p, d.mp, n, d.nulls, k, 1 if descending else 0, ov.p, oi.p, ctypes.byref(cnt))
            wv, wi, wok = orc.top_k(v, k, descending, valid)
            assert cnt.value == len(wv)
            gvals, gidx = ov.get(cnt.value), oi.get(cnt.value)
            # the contract (top_k.cu:47-74: order and tie choice unspecified): the multiset of the k values, and rows that hold them
            if k >= n:
                np.testing.assert_array_equal(gidx, wi)
                assert gvals.tobytes() == wv.tobytes()
                continue
            assert len(np.unique(gidx)) == len(gidx)
            gok = np.ones(len(gidx), bool) if valid is None else valid[gidx]
            assert int(gok.sum()) == int(wok.sum())  # nulls make the top only once the valid rows are used up
            np.testing.assert_array_equal(np.sort(orc.sortable_bits(gvals[gok])), np.sort(orc.sortable_bits(wv[wok])))
            assert v[gidx][gok].tobytes() == gvals[gok].tobytes()


# ---------------------------------------------------------------------------------------------------------------------
# segmented sort
# ---------------------------------------------------------------------------------------------------------------------
def test_segmented_sorted_order_matches_oracle_at_scale(shim):
    import torch
    rng = np.random.default_rng(11)
    n = 1_200_000
    # segments of very different sizes, empty segments, a head and a tail outside every segment
    cuts = np.sort(rng.choice(np.arange(1000, n - 1000), 4000, replace=False))
    cuts = np.concatenate([cuts[:50], cuts[50:51].repeat(3), cuts[51:], [n - 777]]).astype(np.int32)
    cuts = np.sort(cuts)
    cols = [rng.integers(0, 50, n).astype(np.int32), rng.integers(-10**9, 10**9, n).astype(np.int64)]
    valids = [_rand_valid(rng, n, 0.1), None]
    for asc, nb in (([True, False], [True, True]), ([False, True], [False, True])):
        devs = [Dev(c, v) for c, v in zip(cols, valids)]
        dt = (ctypes.c_int * 2)(*[d.tid for d in devs])
        dp = (ctypes.c_void_p * 2)(*[d.p.value for d in devs])
        vp = (ctypes.c_void_p * 2)(*[(d.mp.value if d.mp is not None else None) for d in devs])
        nl = (ctypes.c_int * 2)(*[d.nulls for d in devs])
        de = (ctypes.c_int * 2)(*[0 if a else 1 for a in asc])
        nbv = (ctypes.c_int * 2)(*[1 if b else 0 for b in nb])
        off = torch.from_numpy(cuts.copy()).cuda()
        out = Out(np.int32, n)
        shim("shim_segmented_sorted_order", 2, dt, dp, vp, nl, n, ctypes.c_void_p(off.data_ptr()), len(cuts), de, nbv, 1, out.p)
        want = orc.segmented_sorted_order(cols, cuts, valids, asc, nb)
        np.testing.assert_array_equal(out.get(n), want)


# ---------------------------------------------------------------------------------------------------------------------
# sort-path groupby::aggregate
# ---------------------------------------------------------------------------------------------------------------------
def _groupby_inputs(rng, n, vdtype, shape):
    if shape == "few_long_groups":      # groups of ~n/7 rows: every group spans many scan chunks
        keys = rng.integers(0, 7, n).astype(np.int32)
    elif shape == "single_row_groups":  # every group is one row
        keys = rng.permutation(n).astype(np.int32)
    else:                               # mixed: a few huge groups among thousands of small ones
        keys = np.where(rng.random(n) < 0.4, rng.integers(0, 3, n), rng.integers(3, n // 20, n)).astype(np.int32)
    if np.dtype(vdtype).kind == "f":
        vals = (rng.random(n) * 2 + 0.5).astype(vdtype)
    else:
        vals = rng.integers(-1000, 1000, n).astype(vdtype)
    return keys, vals


@pytest.mark.parametrize("agg", ["sum", "product", "min", "max", "count_valid", "count_all", "nth"])
@pytest.mark.parametrize("shape", ["few_long_groups", "single_row_groups", "mixed"])
def test_sort_path_groupby_matches_oracle_at_scale(shim, agg, shape):
    rng = np.random.default_rng(zlib.crc32(f"{agg}/{shape}".encode()))
    for vdtype, null_include, keys_sorted in (("int32", False, False), ("float64", True, False), ("int64", False, True)):
        n = 700_001
        keys, vals = _groupby_inputs(rng, n, vdtype, shape)
        kvalid = _rand_valid(rng, n, 0.05)
        vvalid = _rand_valid(rng, n, 0.15)
        if agg == "product" and vdtype == "float64":
            vals = np.where(rng.random(n) < 0.5, 1.0 + rng.random(n) * 1e-3, 1.0 / (1.0 + rng.random(n) * 1e-3))
        if keys_sorted:  # sorted::YES: the caller hands over keys that are already in order
            if not null_include:  # (pre-sorted keys with null rows to drop fall back to sorting, sort_helper.cu:80-86)
                keys, vals, vvalid, kvalid = keys[kvalid], vals[kvalid], vvalid[kvalid], None
            o = orc.sorted_order(keys, kvalid, True, False)
            keys, vals, vvalid = keys[o], vals[o], vvalid[o]
            kvalid = None if kvalid is None else kvalid[o]
        n = len(keys)
        dk, dv = Dev(keys, kvalid), Dev(vals, vvalid)
        n_th = {"few_long_groups": -2, "single_row_groups": 0, "mixed": 1}[shape]
        ok_, okm = Out(np.int32, n, True), None
        ov = Out(np.int64 if vdtype != "float64" else np.float64, n, True)
        kn, vn, g, vt = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        shim("shim_groupby_aggregate", dk.tid, dk.p, dk.mp, dk.nulls, dv.tid, dv.p, dv.mp, dv.nulls, n, KIND[agg], n_th,
             1 if null_include else 0, 1 if keys_sorted else 0, 0 if agg == "nth" else 1, ok_.p, ok_.mp, ctypes.byref(kn), ov.p, ov.mp,
             ctypes.byref(vn), ctypes.byref(g), ctypes.byref(vt))
        wk, wkv, wout, wov = orc.groupby_sort_agg(keys, vals, agg, kvalid, vvalid, null_include, keys_sorted, n_th)
        G = g.value
        assert G == len(wk)
        np.testing.assert_array_equal(ok_.valid(G), wkv)                 # the null-key group comes last under INCLUDE
        np.testing.assert_array_equal(ok_.get(G)[wkv], wk[wkv])         # keys come out SORTED on this path
        got = ov.get(G, NPT[vt.value])
        assert got.dtype == wout.dtype, (got.dtype, wout.dtype)
        gv = ov.valid(G)
        np.testing.assert_array_equal(gv, wov)
        if got.dtype.kind == "f":
            tol = 1 if agg != "product" else None
            if tol:
                assert _ulp_ok(got[wov], wout[wov], 1)
            else:  # a product's rounding depends on the association order (thrust::reduce_by_key promises none)
                np.testing.assert_allclose(got[wov], wout[wov], rtol=1e-11)
        else:
            np.testing.assert_array_equal(got[wov], wout[wov])


# ---------------------------------------------------------------------------------------------------------------------
# groupby::scan (SUM / MIN / MAX / COUNT)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("op", ["sum", "min", "max", "count_valid", "count_all"])
@pytest.mark.parametrize("shape", ["few_long_groups", "mixed"])
def test_groupby_scan_matches_oracle_at_scale(shim, op, shape):
    rng = np.random.default_rng(zlib.crc32(f"{op}/{shape}".encode()))
    n = 900_007
    for vdtype, null_include in (("int32", False), ("float64", True), ("int64", False)):
        keys, vals = _groupby_inputs(rng, n, vdtype, shape)
        kvalid = _rand_valid(rng, n, 0.03)
        vvalid = _rand_valid(rng, n, 0.2)
        dk, dv = Dev(keys, kvalid), Dev(vals, vvalid)
        ok_ = Out(np.int32, n)
        ov = Out(np.int64 if (vdtype != "float64" and op == "sum") else (np.int32 if op.startswith("count") else vdtype), n, True)
        vn, rows, vt = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        shim("shim_groupby_scan", dk.tid, dk.p, dk.mp, dk.nulls, dv.tid, dv.p, dv.mp, dv.nulls, n, KIND[op],
             1 if null_include else 0, 0, ok_.p, ov.p, ov.mp, ctypes.byref(vn), ctypes.byref(rows), ctypes.byref(vt))
        wk, wout, wov = orc.groupby_scan(keys, vals, op, kvalid, vvalid, null_include)
        m = rows.value
        assert m == len(wk)
        sk = ok_.get(m)
        keep = np.ones(m, bool)
        if null_include:
            keep[m - int((~kvalid).sum()):] = False  # the null keys' values are unspecified
        np.testing.assert_array_equal(sk[keep], wk[keep])
        got, gv = ov.get(m, NPT[vt.value]), ov.valid(m)
        assert got.dtype == wout.dtype
        np.testing.assert_array_equal(gv, wov)
        if got.dtype.kind == "f" and op == "sum":
            np.testing.assert_allclose(got[wov], wout[wov], rtol=1e-12)
        else:
            np.testing.assert_array_equal(got[wov], wout[wov])


# ---------------------------------------------------------------------------------------------------------------------
# get_groups / shift / replace_nulls
# ---------------------------------------------------------------------------------------------------------------------
def test_get_groups_matches_oracle_at_scale(shim):
    rng = np.random.default_rng(21)
    n = 1_000_003
    for shape, null_include in (("mixed", False), ("few_long_groups", True), ("single_row_groups", False)):
        keys, vals = _groupby_inputs(rng, n, "int64", shape)
        kvalid = _rand_valid(rng, n, 0.1)
        dk, dv = Dev(keys, kvalid), Dev(vals)
        okk, ovv = Out(np.int32, n), Out(np.int64, n)
        offs = (ctypes.c_int32 * (n + 1))()
        g, rows = ctypes.c_int(), ctypes.c_int()
        shim("shim_groupby_get_groups", dk.tid, dk.p, dk.mp, dk.nulls, dv.tid, dv.p, None, 0, n, 1 if null_include else 0,
             okk.p, ovv.p, offs, ctypes.byref(g), ctypes.byref(rows))
        wk, wv, woff = orc.groupby_get_groups(keys, vals, kvalid, null_include)
        assert rows.value == len(wk) and g.value == len(woff) - 1
        np.testing.assert_array_equal(np.frombuffer(offs, np.int32, g.value + 1), woff)
        nvalid = int(kvalid.sum())
        np.testing.assert_array_equal(okk.get(rows.value)[:nvalid], wk[:nvalid])
        np.testing.assert_array_equal(ovv.get(rows.value), wv)  # stable: values in input order inside a group


@pytest.mark.parametrize("offset", [1, -1, 37, -4099, 3_000_000])
def test_groupby_shift_matches_oracle_at_scale(shim, offset):
    rng = np.random.default_rng(31 + abs(offset))
    n = 800_009
    for vdtype, fill in (("int32", 42), ("float64", None), ("int64", -7)):
        keys, vals = _groupby_inputs(rng, n, vdtype, "mixed")
        vvalid = _rand_valid(rng, n, 0.25)
        dk, dv = Dev(keys), Dev(vals, vvalid)
        okk, ovv = Out(np.int32, n), Out(vdtype, n, True)
        bits = 0 if fill is None else int(np.array([fill], vdtype).view(np.uint32 if np.dtype(vdtype).itemsize == 4 else np.uint64)[0])
        vn, rows = ctypes.c_int(), ctypes.c_int()
        shim("shim_groupby_shift", dk.tid, dk.p, None, 0, dv.tid, dv.p, dv.mp, dv.nulls, n, offset, ctypes.c_ulonglong(bits),
             0 if fill is None else 1, okk.p, ovv.p, ovv.mp, ctypes.byref(vn), ctypes.byref(rows))
        wk, wout, wov = orc.groupby_shift(keys, vals, offset, fill, None, vvalid)
        assert rows.value == n
        np.testing.assert_array_equal(okk.get(n), wk)
        np.testing.assert_array_equal(ovv.valid(n), wov)
        assert vn.value == int((~wov).sum())
        np.testing.assert_array_equal(ovv.get(n)[wov], wout[wov])


@pytest.mark.parametrize("following", [False, True])
def test_groupby_replace_nulls_matches_oracle_at_scale(shim, following):
    rng = np.random.default_rng(41)
    n = 600_011
    for vdtype, shape, p_null in (("int32", "mixed", 0.5), ("float64", "few_long_groups", 0.97), ("int64", "mixed", 0.1)):
        keys, vals = _groupby_inputs(rng, n, vdtype, shape)
        vvalid = _rand_valid(rng, n, p_null)   # 97 % nulls: fills that travel across many scan chunks
        dk, dv = Dev(keys), Dev(vals, vvalid)
        okk, ovv = Out(np.int32, n), Out(vdtype, n, True)
        vn, rows = ctypes.c_int(), ctypes.c_int()
        shim("shim_groupby_replace_nulls", dk.tid, dk.p, None, 0, dv.tid, dv.p, dv.mp, dv.nulls, n, 1 if following else 0,
             okk.p, ovv.p, ovv.mp, ctypes.byref(vn), ctypes.byref(rows))
        wk, wout, wov = orc.groupby_replace_nulls(keys, vals, vvalid, following)
        np.testing.assert_array_equal(okk.get(n), wk)
        np.testing.assert_array_equal(ovv.valid(n), wov)
        assert vn.value == int((~wov).sum())
        assert ovv.get(n)[wov].tobytes() == wout[wov].tobytes()


# ---------------------------------------------------------------------------------------------------------------------
# match contexts and the full-join complement
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("nulls_equal", [True, False])
def test_join_match_context_counts_match_oracle_at_scale(shim, kind, nulls_equal):
    rng = np.random.default_rng(51 + kind)
    nl, nr = 3_000_001, 400_003
    right = rng.integers(0, 150_000, nr).astype(np.int64)      # duplicate build keys: counts > 1
    left = rng.integers(0, 300_000, nl).astype(np.int64)
    lv, rv = _rand_valid(rng, nl, 0.02), _rand_valid(rng, nr, 0.01)
    dl, dr = Dev(left, lv), Dev(right, rv)
    out = Out(np.int32, nl)
    shim("shim_join_match_counts", dl.tid, dl.p, dl.mp, dl.nulls, nl, dr.p, dr.mp, dr.nulls, nr, kind, 1 if nulls_equal else 0, out.p)
    want = orc.join_match_counts(left, right, ("inner", "left", "full")[kind], lv, rv, nulls_equal)
    np.testing.assert_array_equal(out.get(nl), want)


@pytest.mark.parametrize("nulls_equal", [True, False])
def test_full_join_matches_oracle_at_scale(shim, nulls_equal):
    rng = np.random.default_rng(61)
    nl, nr = 1_200_000, 700_001
    right = rng.integers(0, 2_000_000, nr).astype(np.int64)
    left = rng.integers(0, 2_000_000, nl).astype(np.int64)
    lv, rv = _rand_valid(rng, nl, 0.001), _rand_valid(rng, nr, 0.001)
    dl, dr = Dev(left, lv), Dev(right, rv)
    wl, wr = orc.full_join([left], [right], [lv], [rv], nulls_equal)
    cap = len(wl) + 1024
    ol, orr = Out(np.int32, cap), Out(np.int32, cap)
    pairs = ctypes.c_longlong(-1)
    shim("shim_full_join", dl.tid, dl.p, dl.mp, dl.nulls, nl, dr.p, dr.mp, dr.nulls, nr, 1 if nulls_equal else 0,
         ctypes.c_longlong(cap), ol.p, orr.p, ctypes.byref(pairs))
    assert pairs.value == len(wl)
    gl, gr = orc.canonical_pairs(ol.get(pairs.value), orr.get(pairs.value))
    el, er = orc.canonical_pairs(wl, wr)
    np.testing.assert_array_equal(gl, el)
    np.testing.assert_array_equal(gr, er)
