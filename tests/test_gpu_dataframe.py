"""GPU: the pandas-like DataFrame wrapper vs pandas itself (the oracle the reference's Python tests
use: python/cudf/cudf/tests/dataframe/methods/test_sort_values.py, reshape/test_merge.py,
groupby/test_reductions.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def DF():
    import cudf_amd
    return cudf_amd.DataFrame


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
