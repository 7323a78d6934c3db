"""Golden vectors transcribed (inputs / expected literals only) from the reference's own tests.

Every entry cites the reference test file:line (relative to /root/reference/cpp/tests).  These
pin the CPU oracle (tests/test_oracle_golden.py) and are replayed through the HIP path
(the test_reference_golden_* cases of tests/test_gpu_sort.py, tests/test_gpu_join_groupby.py,
tests/test_gpu_reduce_scan_hash.py and tests/cpp/cudf_api_tests.cpp).  ``N`` marks a null element; masks use 1 = valid.
String key columns of the reference tests are transcribed as small integer codes
("s0"->0, "s1"->1, ...) because only their equality classes matter to the join.
"""
import numpy as np

NaN = float("nan")
Inf = float("inf")
NO_MATCH = -(2**31)  # JoinNoMatch, include/cudf/join/join.hpp:72

# ---------------------------------------------------------------------------------------------
# sort
# ---------------------------------------------------------------------------------------------
SORT = [
    # sort/stable_sort_tests.cpp:88-104  StableSort.SingleColumnNoNull (radix fast path)
    dict(name="stable_single_nonull_signed", dtype="int64", values=[7, 1, -2, 5, 1, 0, 1, -2, 0, 5],
         valid=None, ascending=True, null_before=True, expected=[2, 7, 5, 8, 1, 4, 6, 3, 9, 0],
         compare="indices"),
    dict(name="stable_single_nonull_int32", dtype="int32", values=[7, 1, -2, 5, 1, 0, 1, -2, 0, 5],
         valid=None, ascending=True, null_before=True, expected=[2, 7, 5, 8, 1, 4, 6, 3, 9, 0],
         compare="indices"),
    dict(name="stable_single_nonull_f64", dtype="float64", values=[7, 1, -2, 5, 1, 0, 1, -2, 0, 5],
         valid=None, ascending=True, null_before=True, expected=[2, 7, 5, 8, 1, 4, 6, 3, 9, 0],
         compare="indices"),
    # unsigned: -2 wraps to the maximum
    dict(name="stable_single_nonull_unsigned", dtype="uint32",
         values=[7, 1, 2**32 - 2, 5, 1, 0, 1, 2**32 - 2, 0, 5],
         valid=None, ascending=True, null_before=True, expected=[5, 8, 1, 4, 6, 3, 9, 0, 2, 7],
         compare="indices"),
    dict(name="stable_single_nonull_uint64", dtype="uint64",
         values=[7, 1, 2**64 - 2, 5, 1, 0, 1, 2**64 - 2, 0, 5],
         valid=None, ascending=True, null_before=True, expected=[5, 8, 1, 4, 6, 3, 9, 0, 2, 7],
         compare="indices"),
    # sort/stable_sort_tests.cpp:106-123  StableSort.SingleColumnWithNull -- the reference compares
    # the GATHERED table (run_stable_sort_test :20-31), so null rows are interchangeable.
    dict(name="stable_single_withnull_signed", dtype="int64", values=[7, 1, -2, 5, 1, 0, 1, -2, 0, 5],
         valid=[1, 1, 0, 0, 1, 0, 1, 0, 1, 0], ascending=True, null_before=True,
         expected=[2, 7, 5, 3, 9, 8, 1, 4, 6, 0], compare="gathered"),
    dict(name="stable_single_withnull_unsigned", dtype="uint32",
         values=[7, 1, 2**32 - 2, 5, 1, 0, 1, 2**32 - 2, 0, 5],
         valid=[1, 1, 0, 0, 1, 0, 1, 0, 1, 0], ascending=True, null_before=True,
         expected=[5, 3, 9, 2, 7, 8, 1, 4, 6, 0], compare="gathered"),
    # sort/sort_test.cpp:1068-1083  SortDouble.InfinityAndNan (sorted_order, default order)
    dict(name="double_inf_nan", dtype="float64",
         values=[-0.0, -NaN, -NaN, NaN, Inf, -Inf, 7.0, 5.0, 6.0, NaN, Inf, -Inf, -NaN, -NaN, -0.0],
         valid=None, ascending=True, null_before=True,
         expected=[5, 11, 0, 14, 7, 8, 6, 4, 10, 1, 2, 3, 9, 12, 13], compare="indices"),
    # sort/stable_sort_tests.cpp:278-289  same input through stable_sorted_order
    dict(name="double_inf_nan_f32", dtype="float32",
         values=[-0.0, -NaN, -NaN, NaN, Inf, -Inf, 7.0, 5.0, 6.0, NaN, Inf, -Inf, -NaN, -NaN, -0.0],
         valid=None, ascending=True, null_before=True,
         expected=[5, 11, 0, 14, 7, 8, 6, 4, 10, 1, 2, 3, 9, 12, 13], compare="indices"),
]

# several key columns (the reference's tables carry a strings column between the two numeric ones; its order {a, d, d, e, k} agrees
# with the first numeric column's, so the numeric columns alone give the same expected order)
SORT_TABLE = [
    # sort/sort_test.cpp:148-175  Sort.WithAllValid: col1 ASC, col3 DESC
    dict(name="table_all_valid", cols=[[5, 4, 3, 5, 8], [10, 40, 70, 5, 2]], ascending=[True, False], expected=[2, 1, 0, 3, 4]),
    # sort/stable_sort_tests.cpp:150-171  StableSort.WithAllValid: rows 0 and 3 tie in every column -> input order
    dict(name="table_all_valid_stable_tie", cols=[[5, 4, 3, 5, 8], [10, 40, 70, 10, 2]], ascending=[True, False], expected=[2, 1, 0, 3, 4]),
]

# ---------------------------------------------------------------------------------------------
# hash join.  left/right: list of key columns (each a list, None = null element);
# expected_rows: multiset of (left payload..., right payload...) rows of the reference's gold
# table restricted to the integer payload columns; or expected_pairs: (left_idx, right_idx).
# ---------------------------------------------------------------------------------------------
N = None
JOIN = [
    # join/join_tests.cpp:1163-1237  InnerJoinNoNulls, single key column (both null_equality values)
    dict(name="inner_nonulls_single", dtype="int32",
         left=[[3, 1, 2, 0, 2]], right=[[2, 2, 0, 4, 3]],
         left_payload=[[3, 1, 2, 0, 2], [0, 1, 2, 4, 1]], right_payload=[[2, 2, 0, 4, 3], [1, 0, 1, 2, 1]],
         expected_rows=[(3, 0, 3, 1), (2, 2, 2, 1), (2, 2, 2, 0), (0, 4, 0, 1), (2, 1, 2, 1), (2, 1, 2, 0)],
         nulls_equal=[True, False]),
    # same test, two key columns (second is the string column, coded)
    dict(name="inner_nonulls_multi", dtype="int32",
         left=[[3, 1, 2, 0, 2], [1, 1, 0, 4, 0]], right=[[2, 2, 0, 4, 3], [1, 0, 1, 2, 1]],
         left_payload=[[3, 1, 2, 0, 2], [0, 1, 2, 4, 1]], right_payload=[[2, 2, 0, 4, 3], [1, 0, 1, 2, 1]],
         expected_rows=[(3, 0, 3, 1), (2, 2, 2, 0), (2, 1, 2, 0)],
         nulls_equal=[True, False]),
    # join/join_tests.cpp:1239-1283  InnerJoinWithNulls: left string key has a null at row 2
    dict(name="inner_withnulls", dtype="int32",
         left=[[3, 1, 2, 0, 2], [1, 1, N, 4, 0]], right=[[2, 2, 0, 4, 3], [1, 0, 1, 2, 1]],
         left_payload=[[3, 1, 2, 0, 2], [0, 1, 2, 4, 1]], right_payload=[[2, 2, 0, 4, 3]],
         expected_rows=[(3, 0, 3), (2, 1, 2)],
         nulls_equal=[True]),
    # join/join_tests.cpp:1421-1500  InnerJoinOnNulls: nulls on both sides; EQUAL -> 2 rows, UNEQUAL -> 1
    dict(name="inner_onnulls_equal", dtype="int32",
         left=[[3, 1, 2, 0, 2], [1, 1, N, 4, 0]], right=[[2, 2, 0, 4, 3], [1, N, 1, 2, 1]],
         left_payload=[[3, 1, 2, 0, 2], [0, 1, 2, 4, 1]], right_payload=[[2, 2, 0, 4, 3], [1, 0, 1, 2, 1]],
         expected_rows=[(3, 0, 3, 1), (2, 2, 2, 0)],
         nulls_equal=[True]),
    dict(name="inner_onnulls_unequal", dtype="int32",
         left=[[3, 1, 2, 0, 2], [1, 1, N, 4, 0]], right=[[2, 2, 0, 4, 3], [1, N, 1, 2, 1]],
         left_payload=[[3, 1, 2, 0, 2], [0, 1, 2, 4, 1]], right_payload=[[2, 2, 0, 4, 3], [1, 0, 1, 2, 1]],
         expected_rows=[(3, 0, 3, 1)],
         nulls_equal=[False]),
    # join/join_tests.cpp:1906-1940  EqualValuesInnerJoin
    dict(name="equal_values", dtype="int32",
         left=[[0, 0], [0, 0]], right=[[0, 0], [0, 0]],
         expected_pairs=[(0, 0), (0, 1), (1, 0), (1, 1)], nulls_equal=[True]),
    # join/join_tests.cpp:2010-2038  InnerJoinCornerCase (int64)
    dict(name="corner_case", dtype="int64",
         left=[[4, 1, 3, 2, 2, 2, 2]], right=[[2]],
         expected_pairs=[(3, 0), (4, 0), (5, 0), (6, 0)], nulls_equal=[True]),
    # join/join_tests.cpp:2040-2123  HashJoinSequentialProbes (build t1 once, probe three tables)
    dict(name="sequential_probe_inner", dtype="int32",
         left=[[3, 1, 2, 0, 2], [1, 1, 0, 4, 0]], right=[[2, 2, 0, 4, 3], [1, 0, 1, 2, 1]],
         expected_pairs=[(2, 1), (4, 1), (0, 4)], expected_size=3, nulls_equal=[True]),
    dict(name="sequential_probe_left", dtype="int32", how="left",
         left=[[3, 1, 2, 0, 3], [0, 1, 2, 4, 1]], right=[[2, 2, 0, 4, 3], [1, 0, 1, 2, 1]],
         expected_pairs=[(0, NO_MATCH), (1, NO_MATCH), (2, NO_MATCH), (3, NO_MATCH), (4, 4)],
         expected_size=5, nulls_equal=[True]),
    dict(name="sequential_probe_full", dtype="int32", how="full",
         left=[[3, 1, 2, 0, 3], [0, 1, 2, 4, 1]], right=[[2, 2, 0, 4, 3], [1, 0, 1, 2, 1]],
         expected_pairs=[(0, NO_MATCH), (1, NO_MATCH), (2, NO_MATCH), (3, NO_MATCH), (4, 4),
                         (NO_MATCH, 0), (NO_MATCH, 1), (NO_MATCH, 2), (NO_MATCH, 3)],
         expected_size=9, nulls_equal=[True]),
]

# ---------------------------------------------------------------------------------------------
# semi / anti join (cudf::filtered_join) and distinct_hash_join.  expected = ascending left row
# indices (semi/anti) or the right row per left row (distinct left join, NO_MATCH where none).
# ---------------------------------------------------------------------------------------------
SEMI_ANTI = [
    # join/semi_anti_join_tests.cpp:119-136  TestSimple (result table {0, 1} = left rows 0, 1)
    dict(name="semi_simple", dtype="int32", how="semi", left=[[0, 1, 2]], right=[[0, 1, 3]],
         expected=[0, 1], nulls_equal=True),
    # join/semi_anti_join_tests.cpp:357-388  AntiJoinEmptyTables: empty right -> every left row; empty left -> none
    dict(name="anti_empty_right", dtype="int32", how="anti", left=[[0, 1, 2]], right=[[]],
         expected=[0, 1, 2], nulls_equal=True),
    dict(name="anti_empty_left", dtype="int32", how="anti", left=[[]], right=[[0, 1, 2]],
         expected=[], nulls_equal=True),
    # join/semi_anti_join_tests.cpp:390-421  SemiJoinEmptyTables
    dict(name="semi_empty_right", dtype="int32", how="semi", left=[[0, 1, 2]], right=[[]],
         expected=[], nulls_equal=True),
    dict(name="semi_empty_left", dtype="int32", how="semi", left=[[]], right=[[0, 1, 2]],
         expected=[], nulls_equal=True),
]

DISTINCT_JOIN = [
    # join/distinct_join_tests.cpp:74-93  IntegerInnerJoin: right = 0..2023, left = 0,2,..,4046 -> the
    # 1012 left rows with value < 2024 match right row == value
    dict(name="distinct_integer_inner", dtype="int32", how="inner",
         left=[list(range(0, 4048, 2))], right=[list(range(2024))],
         expected_pairs=[(i, 2 * i) for i in range(1012)]),
    # join/distinct_join_tests.cpp:541-575  PrimitiveLeftJoinNoNulls (two int32 key columns)
    dict(name="distinct_left_nonulls", dtype="int32", how="left",
         left=[[3, 1, 2, 0, 3], [0, 1, 2, 4, 1]], right=[[2, 2, 0, 4, 3], [1, 0, 1, 2, 1]],
         expected=[NO_MATCH, NO_MATCH, NO_MATCH, NO_MATCH, 4]),
    # join/distinct_join_tests.cpp:614-649  PrimitiveLeftJoinWithNulls (left second key null at row 2)
    dict(name="distinct_left_withnulls", dtype="int32", how="left",
         left=[[3, 1, 2, 0, 2], [1, 1, N, 4, 0]], right=[[2, 2, 0, 4, 3], [1, 0, 1, 2, 1]],
         expected=[4, NO_MATCH, NO_MATCH, NO_MATCH, 1]),
    # join/distinct_join_tests.cpp:185-225  PrimitiveInnerJoinNoNulls (three int32 key columns; the test's
    # first table is the build/right side)
    dict(name="distinct_inner_3col", dtype="int32", how="inner",
         left=[[1, 2, 3, 4, 9], [0, 0, 0, 4, 4], [9, 9, 9, 0, 9]], right=[[1, 2, 3, 4, 5], [0, 0, 3, 4, 5], [9, 9, 9, 9, 9]],
         expected_pairs=[(0, 0), (1, 1)]),
]

# ---------------------------------------------------------------------------------------------
# groupby (keys int32; results compared after sorting by key)
# ---------------------------------------------------------------------------------------------
_K_BASIC = [1, 2, 3, 1, 2, 2, 1, 3, 3, 2]
_V_BASIC = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]
_K_NULL = [1, 2, 3, 1, 2, 2, 1, 3, 3, 2, 4]
_KM_NULL = [1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1]
_V_NULL = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 4]
_VM_NULL = [0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 0]

GROUPBY = [
    # groupby/sum_tests.cpp:68-80 basic (integral V -> int64; float V -> V)
    dict(name="sum_basic", keys=_K_BASIC, vals=_V_BASIC, agg="sum",
         expect_keys=[1, 2, 3], expect=[9, 19, 17], expect_valid=[1, 1, 1]),
    # groupby/count_tests.cpp:21-41
    dict(name="count_basic", keys=_K_BASIC, vals=_V_BASIC, agg="count_valid",
         expect_keys=[1, 2, 3], expect=[3, 4, 3], expect_valid=[1, 1, 1]),
    # groupby/mean_tests.cpp:37-56
    dict(name="mean_basic", keys=_K_BASIC, vals=_V_BASIC, agg="mean",
         expect_keys=[1, 2, 3], expect=[3.0, 19.0 / 4, 17.0 / 3], expect_valid=[1, 1, 1]),
    # groupby/sum_tests.cpp:82-94 empty input
    dict(name="sum_empty", keys=[], vals=[], agg="sum", expect_keys=[], expect=[], expect_valid=[]),
    # groupby/sum_tests.cpp:96-108 all keys null -> empty result
    dict(name="sum_zero_valid_keys", keys=[1, 2, 3], keys_valid=[0, 0, 0], vals=[3, 4, 5], agg="sum",
         expect_keys=[], expect=[], expect_valid=[]),
    # groupby/sum_tests.cpp:110-122 all values null -> one group, null sum
    dict(name="sum_zero_valid_values", keys=[1, 1, 1], vals=[3, 4, 5], vals_valid=[0, 0, 0], agg="sum",
         expect_keys=[1], expect=[0], expect_valid=[0]),
    # groupby/sum_tests.cpp:124-145 null keys and values
    dict(name="sum_null_keys_values", keys=_K_NULL, keys_valid=_KM_NULL, vals=_V_NULL, vals_valid=_VM_NULL,
         agg="sum", expect_keys=[1, 2, 3, 4], expect=[9, 14, 10, 0], expect_valid=[1, 1, 1, 0]),
    # groupby/count_tests.cpp:103-132
    dict(name="count_valid_null_keys_values", keys=_K_NULL, keys_valid=_KM_NULL, vals=_V_NULL,
         vals_valid=_VM_NULL, agg="count_valid", expect_keys=[1, 2, 3, 4], expect=[2, 3, 2, 0],
         expect_valid=[1, 1, 1, 1]),
    dict(name="count_all_null_keys_values", keys=_K_NULL, keys_valid=_KM_NULL, vals=_V_NULL,
         vals_valid=_VM_NULL, agg="count_all", expect_keys=[1, 2, 3, 4], expect=[3, 4, 2, 1],
         expect_valid=[1, 1, 1, 1]),
    # groupby/mean_tests.cpp:102-128
    dict(name="mean_null_keys_values", keys=_K_NULL, keys_valid=_KM_NULL, vals=_V_NULL, vals_valid=_VM_NULL,
         agg="mean", expect_keys=[1, 2, 3, 4], expect=[4.5, 14.0 / 3, 5.0, 0.0], expect_valid=[1, 1, 1, 0]),
    # groupby/sum_tests.cpp:167-188 int32 overflow accumulates in int64
    dict(name="sum_overflow_int32", keys=[0, 0], vals=[-2147483648, -2147483648], vals_dtype="int32",
         agg="sum", expect_keys=[0], expect=[-4294967296], expect_valid=[1]),
    # groupby/min_tests.cpp:26-42 basic, :98-120 null keys and values; :497-516 -inf among the values
    dict(name="min_basic", keys=_K_BASIC, vals=_V_BASIC, agg="min",
         expect_keys=[1, 2, 3], expect=[0, 1, 2], expect_valid=[1, 1, 1]),
    dict(name="min_null_keys_values", keys=_K_NULL, keys_valid=_KM_NULL, vals=_V_NULL, vals_valid=_VM_NULL, agg="min",
         expect_keys=[1, 2, 3, 4], expect=[3, 1, 2, 0], expect_valid=[1, 1, 1, 0]),
    dict(name="min_with_infinity", keys=[1, 2, 1, 2], vals=[1.0, 1.0, float("-inf"), 2.0], vals_dtype="float64", agg="min",
         expect_keys=[1, 2], expect=[float("-inf"), 1.0], expect_valid=[1, 1]),
    # groupby/max_tests.cpp:30-46 basic, :102-123 null keys and values; :505-523 +inf among the values
    dict(name="max_basic", keys=_K_BASIC, vals=_V_BASIC, agg="max",
         expect_keys=[1, 2, 3], expect=[6, 9, 8], expect_valid=[1, 1, 1]),
    dict(name="max_null_keys_values", keys=_K_NULL, keys_valid=_KM_NULL, vals=_V_NULL,
         vals_valid=[1, 1, 1, 1, 1, 1, 0, 1, 1, 0, 0], agg="max",
         expect_keys=[1, 2, 3, 4], expect=[3, 5, 8, 0], expect_valid=[1, 1, 1, 0]),
    dict(name="max_with_infinity", keys=[1, 2, 1, 2], vals=[1.0, 1.0, float("inf"), 2.0], vals_dtype="float64", agg="max",
         expect_keys=[1, 2], expect=[float("inf"), 2.0], expect_valid=[1, 1]),
    # groupby/var_tests.cpp:24-42 basic; :92-113 null keys and values; :115-137 ddof = 2
    dict(name="var_basic", keys=_K_BASIC, vals=_V_BASIC, agg="var",
         expect_keys=[1, 2, 3], expect=[9.0, 131.0 / 12, 31.0 / 3], expect_valid=[1, 1, 1]),
    dict(name="var_null_keys_values", keys=_K_NULL, keys_valid=_KM_NULL, vals=[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 3],
         vals_valid=[0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1], agg="var",
         expect_keys=[1, 2, 3, 4], expect=[4.5, 49.0 / 3, 18.0, 0.0], expect_valid=[1, 1, 1, 0]),
    dict(name="var_ddof2", keys=_K_NULL, keys_valid=_KM_NULL, vals=[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 3],
         vals_valid=[0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1], agg="var", ddof=2,
         expect_keys=[1, 2, 3, 4], expect=[0.0, 98.0 / 3, 0.0, 0.0], expect_valid=[0, 1, 0, 0]),
    # groupby/std_tests.cpp:24-42 basic; :92-112 null keys and values; :114-134 ddof = 2
    dict(name="std_basic", keys=_K_BASIC, vals=_V_BASIC, agg="std",
         expect_keys=[1, 2, 3], expect=[3.0, (131.0 / 12) ** 0.5, (31.0 / 3) ** 0.5], expect_valid=[1, 1, 1]),
    dict(name="std_null_keys_values", keys=_K_NULL, keys_valid=_KM_NULL, vals=[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 3],
         vals_valid=[0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1], agg="std",
         expect_keys=[1, 2, 3, 4], expect=[3 / 2 ** 0.5, 7 / 3 ** 0.5, 3 * 2 ** 0.5, 0.0], expect_valid=[1, 1, 1, 0]),
    dict(name="std_ddof2", keys=_K_NULL, keys_valid=_KM_NULL, vals=[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 3],
         vals_valid=[0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1], agg="std", ddof=2,
         expect_keys=[1, 2, 3, 4], expect=[0.0, 7 * (2.0 / 3) ** 0.5, 0.0, 0.0], expect_valid=[0, 1, 0, 0]),
    # groupby/argmin_tests.cpp:23-41 basic, :83-108 null keys and values
    dict(name="argmin_basic", keys=_K_BASIC, vals=[9, 8, 7, 6, 5, 4, 3, 2, 1, 0], agg="argmin",
         expect_keys=[1, 2, 3], expect=[6, 9, 8], expect_valid=[1, 1, 1]),
    dict(name="argmin_null_keys_values", keys=_K_NULL, keys_valid=_KM_NULL, vals=[9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 4],
         vals_valid=[1, 1, 1, 1, 1, 0, 0, 1, 1, 1, 0], agg="argmin",
         expect_keys=[1, 2, 3, 4], expect=[3, 9, 8, 0], expect_valid=[1, 1, 1, 0]),
    # groupby/argmax_tests.cpp:22-40 basic, :82-107 null keys and values (the null key is row 2 here)
    dict(name="argmax_basic", keys=_K_BASIC, vals=[9, 8, 7, 6, 5, 4, 3, 2, 1, 0], agg="argmax",
         expect_keys=[1, 2, 3], expect=[0, 1, 2], expect_valid=[1, 1, 1]),
    dict(name="argmax_null_keys_values", keys=_K_NULL, keys_valid=[1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1],
         vals=[9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 4], vals_valid=[0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0], agg="argmax",
         expect_keys=[1, 2, 3, 4], expect=[3, 4, 7, 0], expect_valid=[1, 1, 1, 0]),
]

GROUPBY_SCAN = [
    # groupby/sum_scan_tests.cpp:33-48 basic
    dict(name="sum_scan_basic", keys=_K_BASIC, vals=_V_BASIC,
         expect_keys=[1, 1, 1, 2, 2, 2, 2, 3, 3, 3], expect=[0, 3, 9, 1, 5, 10, 19, 2, 9, 17],
         expect_valid=[1] * 10),
    # groupby/sum_scan_tests.cpp:118-139 null keys and values
    dict(name="sum_scan_null_keys_values", keys=_K_NULL, keys_valid=_KM_NULL, vals=_V_NULL,
         vals_valid=_VM_NULL, expect_keys=[1, 1, 1, 2, 2, 2, 2, 3, 3, 4],
         expect=[-1, 3, 9, 1, 5, -1, 14, 2, 10, -1], expect_valid=[0, 1, 1, 1, 1, 0, 1, 1, 1, 0]),
]

# ---------------------------------------------------------------------------------------------
# column scan / reduce  (reductions/scan_tests.cpp:164-213; typed over every numeric type)
# ---------------------------------------------------------------------------------------------
_SCAN_IN = [5, 4, 6, 0, 1, 6, 5, 3]
_SCAN_MASK = [1, 1, 1, 0, 1, 1, 1, 1]
SCAN = [
    dict(name="sum_inclusive_nonulls", op="sum", inclusive=True, values=_SCAN_IN, valid=None,
         null_include=False, expect=[5, 9, 15, 15, 16, 22, 27, 30], expect_valid=[1] * 8),   # :164-172
    dict(name="sum_exclusive_nonulls", op="sum", inclusive=False, values=_SCAN_IN, valid=None,
         null_include=False, expect=[0, 5, 9, 15, 15, 16, 22, 27], expect_valid=[1] * 8),    # :175-185
    dict(name="sum_inclusive_nulls_exclude", op="sum", inclusive=True, values=_SCAN_IN, valid=_SCAN_MASK,
         null_include=False, expect=[5, 9, 15, 15, 16, 22, 27, 30], expect_valid=_SCAN_MASK),  # :188-199
    dict(name="sum_inclusive_nulls_include", op="sum", inclusive=True, values=_SCAN_IN, valid=_SCAN_MASK,
         null_include=True, expect=[5, 9, 15, 0, 0, 0, 0, 0], expect_valid=[1, 1, 1, 0, 0, 0, 0, 0]),  # :202-213
]

# ---------------------------------------------------------------------------------------------
# column reduce (reductions/reduction_tests.cpp; the expected values are what the tests' own
# std::accumulate / std::min_element over the literal inputs give; typed over the numeric types)
# ---------------------------------------------------------------------------------------------
REDUCE = [
    # reduction_tests.cpp:330-368  SumReductionTest.Sum
    dict(name="sum", op="sum", values=[6, -14, 13, 64, 0, -13, -20, 45], valid=None, expect=81, expect_valid=True),
    dict(name="sum_nulls", op="sum", values=[6, -14, 13, 64, 0, -13, -20, 45], valid=[1, 1, 0, 0, 1, 1, 1, 1], expect=4,
         expect_valid=True),
    # reduction_tests.cpp:122-200  MinMaxReductionTest.MinMaxReductions
    dict(name="min", op="min", values=[5, 0, -120, -111, 0, 64, 63, 99, 123, -16], valid=None, expect=-120, expect_valid=True),
    dict(name="max", op="max", values=[5, 0, -120, -111, 0, 64, 63, 99, 123, -16], valid=None, expect=123, expect_valid=True),
    dict(name="min_nulls", op="min", values=[5, 0, -120, -111, 0, 64, 63, 99, 123, -16],
         valid=[1, 1, 0, 1, 1, 1, 0, 1, 0, 1], expect=-111, expect_valid=True),
    dict(name="max_nulls", op="max", values=[5, 0, -120, -111, 0, 64, 63, 99, 123, -16],
         valid=[1, 1, 0, 1, 1, 1, 0, 1, 0, 1], expect=99, expect_valid=True),
    # reduction_tests.cpp:999-1013  all_null_output: the result scalar is invalid
    dict(name="sum_all_null", op="sum", values=[1, 2, 3], valid=[0, 0, 0], expect=0, expect_valid=False),
]

# cudf::reduce MEAN / COUNT / ANY / ALL (round 6): literals of reductions/reduction_tests.cpp.  "dtypes": the types the reference's
# typed test runs over; "out": the output type it asks for; init = None: no initial value.
_AA_MASK = [1, 1, 0, 1]
_MEAN = [-3, 2, 1, 0, 5, -3, -2, 28]
_MEAN_MASK = [1, 1, 0, 1, 1, 1, 0, 1]
_CNT = [1, -3, 1, 2, 0, 2, -4, 45]
REDUCE_MORE = [
    # :668-728  ReductionAnyAllTest.AnyAllTrueTrue (int32, float, bool): {1,1,1,1}, init true; then nulls {1,1,0,1} and an INVALID init
    dict(name="any_true", op="any", dtypes=["int32", "float32", "bool"], out="bool", values=[1, 1, 1, 1], valid=None, init=None, init_valid=True, expect=True, expect_valid=True),
    dict(name="all_true", op="all", dtypes=["int32", "float32", "bool"], out="bool", values=[1, 1, 1, 1], valid=None, init=None, init_valid=True, expect=True, expect_valid=True),
    dict(name="any_true_init", op="any", dtypes=["int32", "float32", "bool"], out="bool", values=[1, 1, 1, 1], valid=None, init=1, init_valid=True, expect=True, expect_valid=True),
    dict(name="all_true_init", op="all", dtypes=["int32", "float32", "bool"], out="bool", values=[1, 1, 1, 1], valid=None, init=1, init_valid=True, expect=True, expect_valid=True),
    dict(name="any_true_nulls", op="any", dtypes=["int32", "float32", "bool"], out="bool", values=[1, 1, 1, 1], valid=_AA_MASK, init=None, init_valid=True, expect=True, expect_valid=True),
    dict(name="all_true_nulls", op="all", dtypes=["int32", "float32", "bool"], out="bool", values=[1, 1, 1, 1], valid=_AA_MASK, init=None, init_valid=True, expect=True, expect_valid=True),
    dict(name="any_true_nulls_invalid_init", op="any", dtypes=["int32", "float32", "bool"], out="bool", values=[1, 1, 1, 1], valid=_AA_MASK, init=1, init_valid=False, expect=True, expect_valid=False),
    dict(name="all_true_nulls_invalid_init", op="all", dtypes=["int32", "float32", "bool"], out="bool", values=[1, 1, 1, 1], valid=_AA_MASK, init=1, init_valid=False, expect=True, expect_valid=False),
    # :731-791  AnyAllFalseFalse: {0,0,0,0}, init false
    dict(name="any_false", op="any", dtypes=["int32", "float32", "bool"], out="bool", values=[0, 0, 0, 0], valid=None, init=None, init_valid=True, expect=False, expect_valid=True),
    dict(name="all_false", op="all", dtypes=["int32", "float32", "bool"], out="bool", values=[0, 0, 0, 0], valid=None, init=None, init_valid=True, expect=False, expect_valid=True),
    dict(name="any_false_init", op="any", dtypes=["int32", "float32", "bool"], out="bool", values=[0, 0, 0, 0], valid=None, init=0, init_valid=True, expect=False, expect_valid=True),
    dict(name="all_false_init", op="all", dtypes=["int32", "float32", "bool"], out="bool", values=[0, 0, 0, 0], valid=None, init=0, init_valid=True, expect=False, expect_valid=True),
    dict(name="any_false_nulls", op="any", dtypes=["int32", "float32", "bool"], out="bool", values=[0, 0, 0, 0], valid=_AA_MASK, init=None, init_valid=True, expect=False, expect_valid=True),
    dict(name="all_false_nulls", op="all", dtypes=["int32", "float32", "bool"], out="bool", values=[0, 0, 0, 0], valid=_AA_MASK, init=None, init_valid=True, expect=False, expect_valid=True),
    dict(name="any_false_nulls_invalid_init", op="any", dtypes=["int32", "float32", "bool"], out="bool", values=[0, 0, 0, 0], valid=_AA_MASK, init=0, init_valid=False, expect=False, expect_valid=False),
    # :1145-1178  ReductionEmptyTest.empty_column: an empty column and five all-null rows -> any = false, all = true, both VALID
    dict(name="any_empty", op="any", dtypes=["int32"], out="bool", values=[], valid=None, init=None, init_valid=True, expect=False, expect_valid=True),
    dict(name="all_empty", op="all", dtypes=["int32"], out="bool", values=[], valid=None, init=None, init_valid=True, expect=True, expect_valid=True),
    dict(name="any_all_null", op="any", dtypes=["int32"], out="bool", values=[0] * 5, valid=[0] * 5, init=None, init_valid=True, expect=False, expect_valid=True),
    dict(name="all_all_null", op="all", dtypes=["int32"], out="bool", values=[0] * 5, valid=[0] * 5, init=None, init_valid=True, expect=True, expect_valid=True),
    # :1201-1208  COUNT (null_policy::INCLUDE) of the empty column = 0, of the five all-null rows = 5, both valid
    dict(name="count_all_empty", op="count_all", dtypes=["int32"], out="int32", values=[], valid=None, init=None, init_valid=True, expect=0, expect_valid=True),
    dict(name="count_all_all_null", op="count_all", dtypes=["int32"], out="int32", values=[0] * 5, valid=[0] * 5, init=None, init_valid=True, expect=5, expect_valid=True),
    # :806-840  MultiStepReductionTest.Mean (int16, int32, float, double), FLOAT64 out: 28 / 8; nulls {1, -2 masked}: 29 / 6
    dict(name="mean", op="mean", dtypes=["int16", "int32", "float32", "float64"], out="float64", values=_MEAN, valid=None, init=None, init_valid=True, expect=3.5, expect_valid=True),
    dict(name="mean_nulls", op="mean", dtypes=["int16", "int32", "float32", "float64"], out="float64", values=_MEAN, valid=_MEAN_MASK, init=None, init_valid=True, expect=29 / 6, expect_valid=True),
    # :1759-1785  ReductionTest.Count (typed over the numeric types), size_type out: 8 / 8; null_at(3): 8 / 7
    dict(name="count_all", op="count_all", dtypes=["int8", "int32", "int64", "float32", "float64"], out="int32", values=_CNT, valid=None, init=None, init_valid=True, expect=8, expect_valid=True),
    dict(name="count_valid", op="count_valid", dtypes=["int8", "int32", "int64", "float32", "float64"], out="int32", values=_CNT, valid=None, init=None, init_valid=True, expect=8, expect_valid=True),
    dict(name="count_all_nulls", op="count_all", dtypes=["int8", "int32", "int64", "float32", "float64"], out="int32", values=_CNT, valid=[1, 1, 1, 0, 1, 1, 1, 1], init=None, init_valid=True, expect=8, expect_valid=True),
    dict(name="count_valid_nulls", op="count_valid", dtypes=["int8", "int32", "int64", "float32", "float64"], out="int32", values=_CNT, valid=[1, 1, 1, 0, 1, 1, 1, 1], init=None, init_valid=True, expect=7, expect_valid=True),
]

# Published MurmurHash3_x86_32 known-answer vectors (Appleby's SMHasher reference implementation;
# the reference delegates the body to cuco::murmurhash3_32,
# include/cudf/hashing/detail/murmurhash3_x86_32.cuh:16,45).  (bytes, seed, digest)
MURMUR3_KAT = [
    (b"", 0, 0x00000000),
    (b"", 1, 0x514E28B7),
    (b"", 0xFFFFFFFF, 0x81F16F39),
    (b"\x00\x00\x00\x00", 0, 0x2362F9DE),
    (b"\xff\xff\xff\xff", 0, 0x76293B50),
    (b"\x21\x43\x65\x87", 0, 0xF55B516B),
    (b"\x21\x43\x65\x87", 0x5082EDEE, 0x2362F9DE),
    (b"\x21\x43\x65", 0, 0x7E4A8634),
    (b"\x21\x43", 0, 0xA0F7B07A),
    (b"\x21", 0, 0x72661CF4),
    (b"aaaa", 0x9747B28C, 0x5A97808A),
    (b"Hello, world!", 0x9747B28C, 0x24884CBA),
    (b"The quick brown fox jumps over the lazy dog", 0x9747B28C, 0x2FA826CD),
    (b"abc", 0, 0xB3DD93FA),
    (b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq", 0, 0xEE925B90),
]


def col(values, dtype, valid=None):
    """Build (np.ndarray values, bool mask or None) with None elements turned into nulls."""
    vals = [0 if v is None else v for v in values]
    mask = None
    if any(v is None for v in values):
        mask = np.array([v is not None for v in values], dtype=bool)
    if valid is not None:
        m2 = np.array(valid, dtype=bool)
        mask = m2 if mask is None else (mask & m2)
    dt = np.dtype(dtype)
    if dt.kind == "u":
        arr = np.array(vals, dtype=object).astype(dt) if vals else np.array([], dt)
    else:
        arr = np.array(vals, dtype=dt)
    return arr, mask


# ---------------------------------------------------------------------------------------------------------------------
# sort-path groupby / groupby::scan COUNT / shift / replace_nulls / top_k / segmented sort
# transcribed from the reference's tests; rank vectors are in rank_vectors.json (make_rank_vectors.py)
# ---------------------------------------------------------------------------------------------------------------------
GROUPBY_SORT = [
    # cpp/tests/groupby/product_tests.cpp:24-45 (basic)
    dict(name="product_basic", agg="product", keys=_K_BASIC, vals=_V_BASIC, expect_keys=[1, 2, 3], expect=[0, 180, 112],
         expect_valid=[1, 1, 1]),
    # cpp/tests/groupby/product_tests.cpp:104-128 (null_keys_and_values)
    dict(name="product_null_keys_and_values", agg="product", keys=_K_NULL, keys_valid=_KM_NULL,
         vals=[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 3], vals_valid=_VM_NULL, expect_keys=[1, 2, 3, 4], expect=[18, 36, 16, 3],
         expect_valid=[1, 1, 1, 0]),
    # cpp/tests/groupby/nth_element_tests.cpp:23-48 (basic: n = 0, 1, 2)
    dict(name="nth0", agg="nth", n=0, keys=_K_BASIC, vals=_V_BASIC, expect_keys=[1, 2, 3], expect=[0, 1, 2], expect_valid=[1, 1, 1]),
    dict(name="nth1", agg="nth", n=1, keys=_K_BASIC, vals=_V_BASIC, expect_keys=[1, 2, 3], expect=[3, 4, 7], expect_valid=[1, 1, 1]),
    dict(name="nth2", agg="nth", n=2, keys=_K_BASIC, vals=_V_BASIC, expect_keys=[1, 2, 3], expect=[6, 5, 8], expect_valid=[1, 1, 1]),
    # cpp/tests/groupby/nth_element_tests.cpp:82-107 (negative: n = -1, -2, -3)
    dict(name="nth-1", agg="nth", n=-1, keys=_K_BASIC, vals=_V_BASIC, expect_keys=[1, 2, 3], expect=[6, 9, 8], expect_valid=[1, 1, 1]),
    dict(name="nth-2", agg="nth", n=-2, keys=_K_BASIC, vals=_V_BASIC, expect_keys=[1, 2, 3], expect=[3, 5, 7], expect_valid=[1, 1, 1]),
    dict(name="nth-3", agg="nth", n=-3, keys=_K_BASIC, vals=_V_BASIC, expect_keys=[1, 2, 3], expect=[0, 4, 2], expect_valid=[1, 1, 1]),
    # cpp/tests/groupby/nth_element_tests.cpp:157-176, 178-197 (null keys and values; n = 0, n = 2 out of bounds)
    dict(name="nth0_nulls", agg="nth", n=0, keys=_K_NULL, keys_valid=_KM_NULL, vals=_V_NULL, vals_valid=_VM_NULL,
         expect_keys=[1, 2, 3, 4], expect=[-1, 1, 2, -1], expect_valid=[0, 1, 1, 0]),
    dict(name="nth2_nulls_oob", agg="nth", n=2, keys=_K_NULL, keys_valid=_KM_NULL, vals=_V_NULL, vals_valid=_VM_NULL,
         expect_keys=[1, 2, 3, 4], expect=[6, -1, -1, -1], expect_valid=[1, 0, 0, 0]),
]

GROUPBY_COUNT_SCAN = [
    # cpp/tests/groupby/count_scan_tests.cpp:39-60 (basic: both null policies give the same counts)
    dict(name="basic", keys=_K_BASIC, vals=_V_BASIC, expect_keys=[1, 1, 1, 2, 2, 2, 2, 3, 3, 3],
         expect_valid_count=[1, 2, 3, 1, 2, 3, 4, 1, 2, 3], expect_all_count=[1, 2, 3, 1, 2, 3, 4, 1, 2, 3]),
    # cpp/tests/groupby/count_scan_tests.cpp:100-118 (zero_valid_values)
    dict(name="zero_valid_values", keys=[1, 1, 1], vals=[3, 4, 5], vals_valid=[0, 0, 0], expect_keys=[1, 1, 1],
         expect_valid_count=[0, 0, 0], expect_all_count=[1, 2, 3]),
    # cpp/tests/groupby/count_scan_tests.cpp:120-145 (null_keys_and_values)
    dict(name="null_keys_and_values", keys=_K_NULL, keys_valid=_KM_NULL, vals=_V_NULL, vals_valid=_VM_NULL,
         expect_keys=[1, 1, 1, 2, 2, 2, 2, 3, 3, 4], expect_valid_count=[0, 1, 2, 1, 2, 2, 3, 1, 2, 0],
         expect_all_count=[1, 2, 3, 1, 2, 3, 4, 1, 2, 1]),
]

_SHIFT_K = [1, 2, 1, 2, 2, 1, 1, 2, 1, 2, 1, 2, 1]
_SHIFT_V = [3, 4, 5, 6, 7, 8, 9, 0, 1, 2, 3, 4, 5]
GROUPBY_SHIFT = [
    # cpp/tests/groupby/shift_tests.cpp:64-77 (ForwardShiftWithoutNull_ValidScalar)
    dict(name="forward3_fill42", keys=_SHIFT_K, vals=_SHIFT_V, offset=3, fill=42,
         expect=[42, 42, 42, 3, 5, 8, 9, 42, 42, 42, 4, 6, 7], expect_valid=[1] * 13),
    # cpp/tests/groupby/shift_tests.cpp:79-94 (ForwardShiftWithNull_ValidScalar)
    dict(name="forward3_nulls_fill42", keys=_SHIFT_K, vals=_SHIFT_V, vals_valid=[1, 0, 1, 0, 1, 0, 0, 1, 0, 1, 1, 0, 1], offset=3, fill=42,
         expect=[42, 42, 42, 3, 5, -1, -1, 42, 42, 42, -1, -1, 7], expect_valid=[1, 1, 1, 1, 1, 0, 0, 1, 1, 1, 0, 0, 1]),
    # cpp/tests/groupby/shift_tests.cpp:96-109 (BackwardShiftWithoutNull_NullScalar)
    dict(name="backward1_nullfill", keys=[1, 2, 1, 2, 2, 1, 1], vals=[3, 4, 5, 6, 7, 8, 9], offset=-1, fill=None,
         expect=[5, 8, 9, -1, 6, 7, -1], expect_valid=[1, 1, 1, 0, 1, 1, 0]),
]

GROUPBY_REPLACE_NULLS = [
    # cpp/tests/groupby/replace_nulls_tests.cpp:40-56 (PrecedingFill)
    dict(name="preceding", keys=[0, 1, 0, 1, 0, 1], vals=[42, 7, 24, 10, 1, 1000], vals_valid=[1, 1, 1, 0, 0, 0], following=False,
         expect_keys=[0, 0, 0, 1, 1, 1], expect=[42, 24, 24, 7, 7, 7], expect_valid=[1, 1, 1, 1, 1, 1]),
    # cpp/tests/groupby/replace_nulls_tests.cpp:58-75 (FollowingFill)
    dict(name="following", keys=[0, 0, 1, 1, 0, 1, 1, 1], vals=[2, 4, 8, 16, 32, 64, 128, 256], vals_valid=[1, 0, 1, 0, 1, 0, 1, 1],
         following=True, expect_keys=[0, 0, 0, 1, 1, 1, 1, 1], expect=[2, 32, 32, 8, 128, 128, 128, 256], expect_valid=[1] * 8),
    # cpp/tests/groupby/replace_nulls_tests.cpp:77-93 (PrecedingFillLeadingNulls)
    dict(name="preceding_leading_nulls", keys=[0, 1, 0, 1, 0, 1], vals=[42, 7, 24, 10, 1, 1000], vals_valid=[0, 0, 1, 0, 0, 0],
         following=False, expect_keys=[0, 0, 0, 1, 1, 1], expect=[-1, 24, 24, -1, -1, -1], expect_valid=[0, 1, 1, 0, 0, 0]),
    # cpp/tests/groupby/replace_nulls_tests.cpp:95-112 (FollowingFillTrailingNulls)
    dict(name="following_trailing_nulls", keys=[0, 0, 1, 1, 0, 1, 1, 1], vals=[2, 4, 8, 16, 32, 64, 128, 256],
         vals_valid=[1, 0, 0, 0, 0, 1, 0, 0], following=True, expect_keys=[0, 0, 0, 1, 1, 1, 1, 1],
         expect=[2, -1, -1, 64, 64, 64, -1, -1], expect_valid=[1, 0, 0, 1, 1, 1, 0, 0]),
]

# cpp/tests/sort/segmented_sort_tests.cpp:70-104 (NoNull: col1, col2, segments) and :141-157 (partial offsets)
_SEG_C1 = [10, 36, 14, 32, 49, 23, 10, 34, 12, 45, 12, 37, 43, 26, 21, 16]
_SEG_C2 = [10, 63, 41, 23, 94, 32, 10, 43, 21, 54, 22, 73, 34, 62, 12, 61]
_SEG_OFF = [0, 3, 5, 5, 5, 6, 11, 13, 14, 16]
SEGMENTED_SORT = [
    dict(name="asc", cols=[_SEG_C1], offsets=_SEG_OFF, ascending=[True],
         expect_col0=[10, 14, 36, 32, 49, 23, 10, 12, 12, 34, 45, 37, 43, 26, 16, 21]),
    dict(name="desc", cols=[_SEG_C1], offsets=_SEG_OFF, ascending=[False],
         expect_col0=[36, 14, 10, 49, 32, 23, 45, 34, 12, 12, 10, 43, 37, 26, 21, 16]),
    dict(name="two_cols_asc_desc", cols=[_SEG_C1, _SEG_C2], offsets=_SEG_OFF, ascending=[True, False],
         expect_col1=[10, 41, 63, 23, 94, 32, 10, 22, 21, 43, 54, 73, 34, 62, 61, 12]),
    dict(name="partial_offsets", cols=[_SEG_C1], offsets=[3, 7], ascending=[True],
         expect_order=[0, 1, 2, 6, 5, 3, 4, 7, 8, 9, 10, 11, 12, 13, 14, 15]),
]

# ---------------------------------------------------------------------------------------------
# cudf::hash_partition -- the CONTRACT cases of partitioning/hash_partition_test.cpp (what the returned table and
# offsets vector must look like).  String columns are transcribed as small integer codes (only their presence as a
# non-key / key column matters to these cases).  "rows" / "cols" / "noffsets" are the EXPECT_EQ literals of the cases.
# ---------------------------------------------------------------------------------------------
_HP_FLOATS = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0]
_HP_INTS = [1, 2, 3, 4, 5, 6, 7, 8]
_HP_STR_CODES = [0, 1, 2, 3, 4, 5, 6, 7]          # {"a","bb","ccc","d","ee","fff","gg","h"}
HASH_PARTITION = [
    # hash_partition_test.cpp:49-60  InvalidColumnsToHash -> std::out_of_range
    dict(name="invalid_columns_to_hash", cols=[("float32", _HP_FLOATS), ("int16", _HP_INTS), ("int32", _HP_STR_CODES)],
         keys=[-1], parts=3, throws="std::out_of_range"),
    # :73-89  ZeroPartitions -> empty table with the input's columns, num_partitions + 1 = 1 offset
    dict(name="zero_partitions", cols=[("float32", _HP_FLOATS), ("int16", _HP_INTS), ("int32", _HP_STR_CODES)],
         keys=[2], parts=0, rows=0, ncols=3, noffsets=1),
    # :91-107  ZeroRows
    dict(name="zero_rows", cols=[("float32", []), ("int16", []), ("int32", [])], keys=[2], parts=3, rows=0, ncols=3, noffsets=4),
    # :109-122  ZeroColumns
    dict(name="zero_columns", cols=[], keys=[], parts=3, rows=0, ncols=0, noffsets=4),
    # :124-141  ZeroColumnsNonEmptyTable: nothing to hash -> an EMPTY result of the input's types
    dict(name="zero_columns_nonempty_table", cols=[("float32", _HP_FLOATS), ("int16", _HP_INTS), ("int32", _HP_STR_CODES)],
         keys=[], parts=3, rows=0, ncols=3, noffsets=4),
    # :264-286  MixedColumnTypes: offsets of size num_partitions + 1, same shape as the input, deterministic
    dict(name="mixed_column_types", cols=[("float32", _HP_FLOATS), ("int16", _HP_INTS), ("int32", _HP_STR_CODES)],
         keys=[0, 2], parts=3, rows=8, ncols=3, noffsets=4, deterministic=True),
    # :343-367  CustomSeedValue
    dict(name="custom_seed", cols=[("float32", _HP_FLOATS), ("int16", _HP_INTS), ("int32", _HP_STR_CODES)],
         keys=[0, 2], parts=3, seed=12345, rows=8, ncols=3, noffsets=4, deterministic=True),
]
# :302-326  ColumnsToHash: two tables that share the hashed column get the same offsets and the same hashed column
HASH_PARTITION_COLUMNS_TO_HASH = dict(to_hash=[1, 2, 3, 4, 5, 6], first=[7, 8, 9, 10, 11, 12], second=[13, 14, 15, 16, 17, 18], parts=3)
# :62-71  InvalidKeyRows: 8 input rows, 9 key rows -> std::invalid_argument
HASH_PARTITION_INVALID_KEY_ROWS = dict(input=_HP_FLOATS, keys=[1, 2, 3, 4, 5, 6, 7, 8, 9], parts=3)
# :173-188 MorePartitionsThanSharedMemory (48 * 1024 rows of `true`), :190-209 LargePartitionCountCorrectness (iota 1000),
# :237-262 LargePartitionCountWithNulls (200 rows, valid = i % 4 != 0 -> 50 nulls survive).  The partition count is
# "shared memory per block / 4 + 10" there; 160 KiB of LDS per CU gives 40970 here.
HASH_PARTITION_LARGE_P = 160 * 1024 // 4 + 10
# :388-472  HashPartitionFixedWidth: (cols, rows, partitions, nulls): MorePartitionsThanRows, LargeInput, HasNulls --
# hash_partition(input, all columns) must give the offsets of hash_partition(input, {murmurhash3_x86_32(input)}, HASH_IDENTITY)
HASH_PARTITION_FIXED_WIDTH = [(5, 10, 50, False), (10, 1000, 10, False), (10, 1000, 10, True)]

# cudf::partition, partitioning/partition_test.cpp (typed over every fixed-width value type x every integral map type but bool,
# :28-33).  The fixed-width column of each case; "expected" is the reference's expected table, compared per partition as a set
# (:85-108 expect_equal_partitions).  :126-140 Identity, :171-189 Reverse, :191-209 SinglePartition, :211-232 EmptyPartitions.
PARTITION_BY_MAP = [
    dict(name="Identity", values=[0, 1, 2, 3, 4, 5], map=[0, 1, 2, 3, 4, 5], parts=6, offsets=[0, 1, 2, 3, 4, 5, 6], expected=[0, 1, 2, 3, 4, 5]),
    dict(name="Reverse", values=[0, 1, 3, 7, 5, 13], map=[5, 4, 3, 2, 1, 0], parts=6, offsets=[0, 1, 2, 3, 4, 5, 6], expected=[13, 5, 7, 3, 1, 0]),
    dict(name="SinglePartition", values=[0, 1, 3, 7, 5, 13], map=[0, 0, 0, 0, 0, 0], parts=1, offsets=[0, 6], expected=[13, 5, 7, 3, 1, 0]),
    dict(name="EmptyPartitions", values=[0, 1, 3, 7, 5, 13], map=[2, 2, 0, 0, 4, 4], parts=5, offsets=[0, 2, 2, 4, 4, 6], expected=[3, 7, 0, 1, 5, 13]),
]

# ---------------------------------------------------------------------------------------------
# cudf::reduce with an initial value (reductions/reduction_tests.cpp).  expect = what the test's own std::accumulate
# over the literals gives; "init_valid": False = init_scalar->set_valid_async(false) -> the result is invalid.
# ---------------------------------------------------------------------------------------------
_MM = [5, 0, -120, -111, 0, 64, 63, 99, 123, -16]
_MM_MASK = [1, 1, 0, 1, 1, 1, 0, 1, 0, 1]
_SUM = [6, -14, 13, 64, 0, -13, -20, 45]
_SUM_MASK = [1, 1, 0, 0, 1, 1, 1, 1]
_PROD = [5, -1, 1, 0, 3, 2, 4]
_PROD_MASK = [1, 1, 0, 0, 1, 1, 1]
REDUCE_INIT = [
    # reduction_tests.cpp:122-161  MinMaxReductions, init 100, no nulls (typed int32 / int64 / float / double)
    dict(name="min_init", op="min", values=_MM, valid=None, init=100, init_valid=True, expect=-120, expect_valid=True),
    dict(name="max_init", op="max", values=_MM, valid=None, init=100, init_valid=True, expect=123, expect_valid=True),
    # :171-205  with nulls {-120, 63, 123 masked}: min(100, -111) / max(100, 99)
    dict(name="min_init_nulls", op="min", values=_MM, valid=_MM_MASK, init=100, init_valid=True, expect=-111, expect_valid=True),
    dict(name="max_init_nulls", op="max", values=_MM, valid=_MM_MASK, init=100, init_valid=True, expect=100, expect_valid=True),
    # :215-235  all rows null + invalid init -> invalid
    dict(name="min_init_all_null", op="min", values=_MM, valid=[0] * 10, init=100, init_valid=False, expect=0, expect_valid=False),
    dict(name="max_init_all_null", op="max", values=_MM, valid=[0] * 10, init=100, init_valid=False, expect=0, expect_valid=False),
    # :330-352  Sum, init 100: 81 + 100
    dict(name="sum_init", op="sum", values=_SUM, valid=None, init=100, init_valid=True, expect=181, expect_valid=True),
    # :354-368  nulls + INVALID init -> invalid result
    dict(name="sum_init_invalid", op="sum", values=_SUM, valid=_SUM_MASK, init=100, init_valid=False, expect=0, expect_valid=False),
    # :372-405  Product, init 4: 5 * -1 * 1 * 0 * ... = 0; with the zero masked: 4 * (5 * -1 * 3 * 2 * 4) = -480 (our extra case
    # on the same literals -- the reference's null case uses the invalid init, below)
    dict(name="product_init", op="product", values=_PROD, valid=None, init=4, init_valid=True, expect=0, expect_valid=True),
    dict(name="product_init_nulls_valid_init", op="product", values=_PROD, valid=_PROD_MASK, init=4, init_valid=True, expect=-480,
         expect_valid=True),
    # :407-421  nulls + INVALID init -> invalid
    dict(name="product_init_invalid", op="product", values=_PROD, valid=_PROD_MASK, init=4, init_valid=False, expect=0, expect_valid=False),
]
