"""GPU: the pandas-like DataFrame wrapper vs pandas itself (the oracle the reference's Python tests
use: python/cudf/cudf/tests/dataframe/methods/test_sort_values.py, reshape/test_merge.py,
groupby/test_reductions.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def DF():
    import cudf          # the name a user of the reference types: an alias package over cudf_amd (cudf/__init__.py)
    import cudf_amd
    assert cudf.DataFrame is cudf_amd.DataFrame
    return cudf.DataFrame


def test_sort_values_single_and_multi_key(DF):
    import pandas as pd
    rng = np.random.default_rng(0)
    n = 200_000
    pdf = pd.DataFrame({"a": rng.integers(0, 50, n).astype(np.int32), "b": rng.random(n), "c": rng.integers(-2**40, 2**40, n)})
    pdf.loc[::17, "b"] = np.nan
    gdf = DF.from_pandas(pdf)
    for by, asc in [("c", True), ("c", False), (["a", "c"], [True, False]), (["a", "b"], True)]:
        exp = pdf.sort_values(by, ascending=asc, kind="stable").reset_index(drop=True)
        got = gdf.sort_values(by, ascending=asc).to_pandas()
        pd.testing.assert_frame_equal(got, exp, check_dtype=False)
    exp = pdf.sort_values("b", na_position="first", kind="stable").reset_index(drop=True)
    pd.testing.assert_frame_equal(gdf.sort_values("b", na_position="first").to_pandas(), exp, check_dtype=False)


def test_sort_values_several_keys_one_word_sort(DF):
    """from 2^18 rows, numeric keys without nulls: DataFrame.sort_values orders the rows with ONE word sort on the tuple
    (ops.sorted_order_table); pandas' stable sort is the oracle, mixed directions, a float key with NaN VALUES (not nulls), ties"""
    import pandas as pd
    import cudf_amd
    from cudf_amd import dataframe as dfm
    rng = np.random.default_rng(12)
    n = 400_003
    assert n >= dfm._TABLE_PATH_MIN_ROWS
    a = rng.integers(0, 40, n).astype(np.int32)
    b = rng.integers(-3, 3, n).astype(np.int8)
    c = rng.integers(-2**50, 2**50, n)
    f = np.round(rng.standard_normal(n), 1)
    v = rng.random(n)
    gdf = DF({"a": a, "b": b, "c": c, "f": f, "v": v})
    pdf = pd.DataFrame({"a": a, "b": b, "c": c, "f": f, "v": v})
    calls = []
    orig = cudf_amd.ops.sorted_order_table
    cudf_amd.ops.sorted_order_table = lambda cols, asc=True: (calls.append(len(cols)), orig(cols, asc))[1]
    try:
        for by, asc in [(["a", "b"], True), (["a", "b", "c"], [True, False, True]), (["b", "f", "a"], [False, True, False]), (["f", "c"], [False, False])]:
            exp = pdf.sort_values(by, ascending=asc, kind="stable").reset_index(drop=True)
            got = gdf.sort_values(by, ascending=asc).to_pandas()
            pd.testing.assert_frame_equal(got, exp, check_dtype=False)
    finally:
        cudf_amd.ops.sorted_order_table = orig
    assert calls == [2, 3, 3, 2]          # the table path was the one that ran
    # NaN values are the greatest of their column in both directions (the reference's comparator, sort_impl.cuh:61-93)
    f2 = f.copy()
    f2[::101] = np.nan
    g2 = DF({"a": a, "f": f2, "v": v})
    got = g2.sort_values(["a", "f"], ascending=[True, True]).to_pandas()
    exp = pd.DataFrame({"a": a, "f": f2, "v": v}).sort_values(["a", "f"], ascending=True, kind="stable", na_position="last").reset_index(drop=True)
    pd.testing.assert_frame_equal(got, exp, check_dtype=False)


@pytest.mark.parametrize("how", ["inner", "left"])
def test_merge_matches_pandas(DF, how):
    import pandas as pd
    rng = np.random.default_rng(1)
    left = pd.DataFrame({"k": rng.integers(0, 5000, 100_000), "x": rng.random(100_000)})
    right = pd.DataFrame({"k": rng.permutation(8000)[:3000], "y": rng.integers(0, 100, 3000).astype(np.int64)})
    exp = left.merge(right, on="k", how=how).sort_values(["k", "x"]).reset_index(drop=True)
    got = DF.from_pandas(left).merge(DF.from_pandas(right), on="k", how=how).to_pandas().sort_values(["k", "x"]).reset_index(drop=True)
    pd.testing.assert_frame_equal(got[exp.columns], exp, check_dtype=False)


def test_groupby_agg_matches_pandas(DF):
    import pandas as pd
    rng = np.random.default_rng(2)
    n = 700_000  # >= 2^19: the LDS-partitioned kernels
    pdf = pd.DataFrame({"k": rng.integers(0, 20_000, n).astype(np.int32), "v": rng.random(n), "w": rng.integers(-1000, 1000, n)})
    exp = pdf.groupby("k").agg(v_sum=("v", "sum"), v_count=("v", "count"), v_mean=("v", "mean"), v_min=("v", "min"),
                               v_max=("v", "max"), w_sum=("w", "sum"), w_min=("w", "min")).reset_index()
    got = DF.from_pandas(pdf).groupby("k").agg({"v": ["sum", "count", "mean", "min", "max"], "w": ["sum", "min"]}).to_pandas()
    np.testing.assert_array_equal(got["v_min"], exp["v_min"])
    np.testing.assert_array_equal(got["v_max"], exp["v_max"])
    np.testing.assert_array_equal(got["w_min"], exp["w_min"])
    np.testing.assert_array_equal(got["k"], exp["k"])
    np.testing.assert_allclose(got["v_sum"], exp["v_sum"], rtol=1e-13)
    np.testing.assert_array_equal(got["v_count"], exp["v_count"])
    np.testing.assert_allclose(got["v_mean"], exp["v_mean"], rtol=1e-13)
    np.testing.assert_array_equal(got["w_sum"], exp["w_sum"])


@pytest.mark.parametrize("nkeys", [1, 2])
@pytest.mark.parametrize("how", ["right", "outer"])
def test_right_and_outer_merge_match_pandas(DF, how, nkeys):
    """merge(how="right" / "outer") (python/cudf/cudf/core/dataframe.py merge; cudf::left_join on the swapped frames / cudf::full_join):
    pandas is the oracle; rows compared after sorting by every column (row order of a join is unspecified)"""
    import pandas as pd
    rng = np.random.default_rng(21 + nkeys)
    nl, nr = 60_000, 9_000
    left = pd.DataFrame({"k": rng.integers(0, 7000, nl), "j": rng.integers(0, 3, nl).astype(np.int32), "x": rng.random(nl)})
    right = pd.DataFrame({"k": rng.integers(3000, 12000, nr), "j": rng.integers(0, 3, nr).astype(np.int32), "y": rng.integers(0, 100, nr).astype(np.int64)})
    on = ["k", "j"][:nkeys]
    if nkeys == 1:
        left, right = left.drop(columns="j"), right.drop(columns="j")
    exp = left.merge(right, on=on, how=how)
    got = DF.from_pandas(left).merge(DF.from_pandas(right), on=on, how=how).to_pandas()
    assert list(got.columns) == list(exp.columns) and len(got) == len(exp)
    cols = list(exp.columns)
    exp = exp.sort_values(cols).reset_index(drop=True)
    got = got.sort_values(cols).reset_index(drop=True)
    pd.testing.assert_frame_equal(got, exp, check_dtype=False)
    assert got["x"].isna().sum() == exp["x"].isna().sum() > 0      # right rows without a partner really occur
    if how == "outer":
        assert got["y"].isna().sum() == exp["y"].isna().sum() > 0


def test_full_join_pairs_match_oracle(DF):
    """ops.full_join (cudf::full_join, join.hpp:240-246): the canonical pair set against the oracle, JoinNoMatch on either side"""
    from cudf_amd import Column, ops
    from oracle import cudf_oracle as orc
    rng = np.random.default_rng(5)
    l = rng.integers(0, 5000, 40_000).astype(np.int64)
    r = rng.integers(2500, 9000, 7_000).astype(np.int64)
    gl, gr = ops.full_join(Column.from_numpy(l), Column.from_numpy(r))
    el, er = orc.full_join([l], [r])
    a = orc.canonical_pairs(gl.to_numpy().astype(np.int64), gr.to_numpy().astype(np.int64))
    b = orc.canonical_pairs(np.asarray(el, np.int64), np.asarray(er, np.int64))
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


@pytest.mark.parametrize("how", ["inner", "left"])
def test_merge_on_two_keys_matches_pandas(DF, how):
    import pandas as pd
    rng = np.random.default_rng(3)
    n = 60_000
    left = pd.DataFrame({"a": rng.integers(0, 40, n).astype(np.int32), "b": rng.integers(0, 300, n), "x": rng.random(n)})
    right = pd.DataFrame({"a": rng.integers(0, 40, 5000).astype(np.int32), "b": rng.integers(0, 300, 5000),
                          "y": rng.integers(0, 100, 5000).astype(np.int64)}).drop_duplicates(["a", "b"])
    exp = left.merge(right, on=["a", "b"], how=how).sort_values(["a", "b", "x"]).reset_index(drop=True)
    got = (DF.from_pandas(left).merge(DF.from_pandas(right), on=["a", "b"], how=how).to_pandas()
           .sort_values(["a", "b", "x"]).reset_index(drop=True))
    pd.testing.assert_frame_equal(got[exp.columns], exp, check_dtype=False)


def test_semi_anti_merge_matches_pandas(DF):
    import pandas as pd
    rng = np.random.default_rng(4)
    left = pd.DataFrame({"k": rng.integers(0, 5000, 50_000), "j": rng.integers(0, 3, 50_000).astype(np.int32), "x": rng.random(50_000)})
    right = pd.DataFrame({"k": rng.permutation(8000)[:3000], "j": rng.integers(0, 3, 3000).astype(np.int32)})
    g = DF.from_pandas(left)
    for on in ("k", ["k", "j"]):
        keys = [on] if isinstance(on, str) else on
        m = left.merge(right[keys].drop_duplicates(), on=keys, how="left", indicator=True)["_merge"].to_numpy() == "both"
        semi = g.merge(DF.from_pandas(right), on=on, how="leftsemi").to_pandas()
        anti = g.merge(DF.from_pandas(right), on=on, how="leftanti").to_pandas()
        pd.testing.assert_frame_equal(semi, left[m].reset_index(drop=True), check_dtype=False)   # left order kept
        pd.testing.assert_frame_equal(anti, left[~m].reset_index(drop=True), check_dtype=False)


def test_groupby_on_two_keys_and_float_key_matches_pandas(DF):
    import pandas as pd
    rng = np.random.default_rng(5)
    n = 300_000
    pdf = pd.DataFrame({"a": rng.integers(0, 30, n).astype(np.int16), "b": rng.integers(-5, 5, n), "f": rng.integers(0, 9, n) / 4,
                        "v": rng.random(n), "w": rng.integers(-1000, 1000, n)})
    for by in (["a", "b"], ["f"], ["b", "f", "a"]):
        exp = pdf.groupby(by).agg(v_sum=("v", "sum"), v_max=("v", "max"), w_sum=("w", "sum"), w_count=("w", "count"),
                                  w_var=("w", "var"), v_std=("v", "std")).reset_index()
        got = DF.from_pandas(pdf).groupby(by).agg({"v": ["sum", "max", "std"], "w": ["sum", "count", "var"]}).to_pandas()
        np.testing.assert_allclose(got["w_var"], exp["w_var"], rtol=1e-9)
        np.testing.assert_allclose(got["v_std"], exp["v_std"], rtol=1e-7)
        for b in by:
            np.testing.assert_array_equal(got[b], exp[b])
        np.testing.assert_allclose(got["v_sum"], exp["v_sum"], rtol=1e-13)
        np.testing.assert_array_equal(got["v_max"], exp["v_max"])
        np.testing.assert_array_equal(got["w_sum"], exp["w_sum"])
        np.testing.assert_array_equal(got["w_count"], exp["w_count"])


def test_groupby_shorthands_match_pandas(DF):
    """GroupBy.sum / count / mean / min / max / var / std over every value column (core/groupby/groupby.py)"""
    import pandas as pd
    rng = np.random.default_rng(31)
    n = 150_000
    pdf = pd.DataFrame({"k": rng.integers(0, 900, n).astype(np.int32), "x": rng.random(n), "y": rng.integers(-50, 50, n).astype(np.int64)})
    g = DF.from_pandas(pdf).groupby("k")
    p = pdf.groupby("k")
    for fn in ("sum", "count", "mean", "min", "max", "var", "std"):
        exp = getattr(p, fn)().reset_index()
        got = getattr(g, fn)().to_pandas()
        assert list(got.columns) == list(exp.columns)
        pd.testing.assert_frame_equal(got, exp, check_dtype=False, rtol=1e-12, atol=1e-12)


def test_import_cudf_from_pandas(DF):
    """`import cudf; cudf.from_pandas(pdf).sort_values(...)` -- the reference's module-level entry point on this path."""
    import cudf
    import pandas as pd
    pdf = pd.DataFrame({"k": np.array([3, 1, 2, 1], np.int64), "v": np.array([0.5, 1.5, 2.5, 3.5])})
    got = cudf.from_pandas(pdf).sort_values("k").to_pandas()
    pd.testing.assert_frame_equal(got, pdf.sort_values("k", kind="stable").reset_index(drop=True), check_dtype=False)
    with pytest.raises(AttributeError):
        cudf.read_parquet  # noqa: B018 -- out of scope: the alias does not pretend
