"""Thin pandas-like wrapper over the hot path (the cudf.DataFrame surface of the reference, reduced
to the methods that land on it): sort_values, merge, groupby(...).agg.

reference: python/cudf/cudf/core/dataframe.py (sort_values -> core/_internals/sorting.py ->
pylibcudf.sorting.sorted_order + gather; merge -> core/join/join.py -> pylibcudf.join.inner_join /
left_join / full_join / left_semi_join / left_anti_join + gather; groupby(...).agg -> core/groupby/groupby.py -> pylibcudf.groupby.aggregate).
Columns are fixed-width numeric; data moves host<->device only in from_pandas / to_pandas.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib as L
from . import ops
from .column import Column, bitmask_words, ptr, stream_ptr

_lib = L.lib


class DataFrame:
    def __init__(self, data: Optional[Dict[str, Union[Column, np.ndarray, Sequence]]] = None):
        self._cols: Dict[str, Column] = {}
        for name, v in (data or {}).items():
            self[name] = v

    # ---- construction / export
    @classmethod
    def from_pandas(cls, pdf) -> "DataFrame":
        df = cls()
        for name in pdf.columns:
            s = pdf[name]
            if s.isna().any():
                valid = ~s.isna().to_numpy()
                vals = s.fillna(0).to_numpy()
                df._cols[str(name)] = Column.from_numpy(vals, valid)
            else:
                df._cols[str(name)] = Column.from_numpy(s.to_numpy())
        return df

    def to_pandas(self):
        import pandas as pd
        out = {}
        for name, c in self._cols.items():
            v = c.to_numpy()
            if c.has_nulls():
                m = c.valid_numpy()
                v = v.astype(np.float64) if v.dtype.kind != "f" else v.copy()
                v[~m] = np.nan
            out[name] = v
        return pd.DataFrame(out)

    # ---- dict-like
    def __setitem__(self, name: str, value):
        col = value if isinstance(value, Column) else Column.from_numpy(np.asarray(value))
        if self._cols and len(col) != len(self):
            raise ValueError("Length of values does not match length of index")
        self._cols[name] = col

    def __getitem__(self, name: str) -> Column:
        return self._cols[name]

    def __len__(self) -> int:
        return next(iter(self._cols.values())).size if self._cols else 0

    @property
    def columns(self) -> List[str]:
        return list(self._cols)

    def _take(self, gather_map: Column, nullify: bool = False) -> "DataFrame":
        out = DataFrame()
        for name, c in self._cols.items():
            out._cols[name] = ops.gather(c, gather_map, nullify_out_of_bounds=nullify)
        return out

    # ---- the hot path
    def sort_values(self, by: Union[str, Sequence[str]], ascending: Union[bool, Sequence[bool]] = True,
                    na_position: str = "last") -> "DataFrame":
        """Stable sort by one or more key columns (pandas semantics: NaN/nulls last by default).
        Several keys = LSD over the key columns with stable sorts, as cudf::sorted_order does."""
        keys = [by] if isinstance(by, str) else list(by)
        asc = [ascending] * len(keys) if isinstance(ascending, bool) else list(ascending)
        if len(asc) != len(keys):
            raise ValueError("Length of ascending must match the number of sort keys")
        if na_position not in ("first", "last"):
            raise ValueError("invalid na_position")
        order = _table_order([self._cols[k] for k in keys], asc)   # several numeric keys without nulls: one word sort on the tuple
        if order is not None:
            return self._take(order)
        for name, a in reversed(list(zip(keys, asc))):
            col = self._cols[name] if order is None else ops.gather(self._cols[name], order)
            # NullOrder mapping of core/_internals/sorting.py:83-90: null_before = asc ^ (na == "last")
            null_before = a ^ (na_position == "last")
            perm = ops.sorted_order(col, ascending=a, null_before=null_before)
            order = perm if order is None else ops.gather(order, perm)
        return self._take(order)

    def merge(self, right: "DataFrame", on: Union[str, Sequence[str]], how: str = "inner",
              suffixes=("_x", "_y")) -> "DataFrame":
        """Equi-join on one or several key columns; how in {"inner", "left", "right", "outer", "leftsemi", "leftanti"}
        (python/cudf/cudf/core/dataframe.py `merge`; null keys match nothing, as pandas' NaN keys do not on this path).
        Row order is unspecified for inner / left / right / outer (as in cudf); leftsemi / leftanti keep the left order.
        Columns: the keys, then the left frame's other columns, then the right frame's (suffixes where names collide)."""
        if how not in ("inner", "left", "right", "outer", "leftsemi", "leftanti"):
            raise NotImplementedError(f"merge(how={how!r}) is not on this path yet")
        on = [on] if isinstance(on, str) else list(on)
        lt, rt = [self._cols[k] for k in on], [right._cols[k] for k in on]
        if how in ("leftsemi", "leftanti"):
            if len(on) == 1:
                lk, rk = lt[0], rt[0]
            else:
                lk, rk = ops.encode_rows([lt, rt], nulls_equal=False)
            fn = ops.left_semi_join if how == "leftsemi" else ops.left_anti_join
            return self._take(fn(lk, rk, nulls_equal=False))
        if how == "inner":
            li, ri = ops.inner_join_tables(lt, rt, nulls_equal=False)
        elif how == "left":
            li, ri = ops.left_join_tables(lt, rt, nulls_equal=False)
        elif how == "right":   # the left join of the swapped frames: every right row appears, li = JoinNoMatch where it has no partner
            ri, li = ops.left_join_tables(rt, lt, nulls_equal=False)
        else:                  # outer: the left join's pairs, then (JoinNoMatch, r) for every right row without a partner
            li, ri = ops.left_join_tables(lt, rt, nulls_equal=False)
            n0 = li.size           # pairs of the left join; the unmatched right rows come behind them
            li, ri = ops._append_unmatched_right(li, ri, len(right))
        null_l, null_r = how in ("right", "outer"), how in ("left", "outer")
        out = DataFrame()
        for k, lc, rc in zip(on, lt, rt):
            if how == "right":
                out._cols[k] = ops.gather(rc, ri)
            elif how == "outer":
                # the key of a pair is the left row's where there is one, else the right row's: unmatched right rows sit BEHIND the
                # left join's pairs (one per left row or match), so the column is two gathers side by side
                head = ops.gather(lc, _head(li, n0))
                tail = ops.gather(rc, _tail(ri, n0))
                out._cols[k] = ops.concat_columns(head, tail)
            else:
                out._cols[k] = ops.gather(lc, li)
        for name, c in self._cols.items():
            if name not in on:
                out._cols[name + (suffixes[0] if name in right._cols else "")] = ops.gather(c, li, nullify_out_of_bounds=null_l)
        for name, c in right._cols.items():
            if name not in on:
                out._cols[name + (suffixes[1] if name in self._cols else "")] = ops.gather(c, ri, nullify_out_of_bounds=null_r)
        return out

    def groupby(self, by: Union[str, Sequence[str]]) -> "GroupBy":
        return GroupBy(self, by)


def _head(col: Column, n: int) -> Column:
    return ops._slice_rows(col, 0, n)


def _tail(col: Column, n: int) -> Column:
    return ops._slice_rows(col, n, col.size)


def _valid_from_counts(counts: Column):
    """(validity words, null count) of per-group results from COUNT_VALID: a group without a valid value is null
    (src/groupby/hash/output_utils.cu:68-70); None when every group has one"""
    n = counts.size
    words = torch.zeros(bitmask_words(n), dtype=torch.int32, device="cuda")
    nulls = torch.zeros(1, dtype=torch.int64, device="cuda")
    L.check(_lib.gx_valid_from_counts(counts.data_ptr, n, ptr(words), ptr(nulls), stream_ptr()), "gx_valid_from_counts")
    k = int(nulls.item())
    return (words, k) if k else (None, 0)


_TABLE_PATH_MIN_ROWS = 1 << 18   # as cudf::sorted_order(table_view) of this tree (cudf_amd/cpp/src/sorting.cpp)


def _table_order(cols: Sequence[Column], asc: Sequence[bool]) -> Optional[Column]:
    """2 - 8 numeric key columns without nulls from 2^18 rows: the lexicographic stable order in ONE word sort on a nested rank of the
    tuple (ops.sorted_order_table -> gx_sorted_order_table; 2 x int64 at 1e9 rows: 32 ms against ~90 for the per-column loop); None
    where that path does not apply.  Same result as the loop: without nulls na_position has nothing to place, and NaN is the greatest
    value of a column on both paths (sort_impl.cuh:61-93)."""
    if not 2 <= len(cols) <= 8 or cols[0].size < _TABLE_PATH_MIN_ROWS:
        return None
    for c in cols:
        if c.has_nulls() or c.dtype.kind not in "iufb" or c.dtype.itemsize not in (1, 2, 4, 8) or (c.dtype.kind == "f" and c.dtype.itemsize == 2):
            return None
    return ops.sorted_order_table(list(cols), list(asc))


def _lexicographic_order(cols: Sequence[Column]) -> Column:
    """stable order of the rows by (cols[0], cols[1], ...): one word sort where _table_order applies, else LSD over the columns with
    the stable radix argsort"""
    order = _table_order(cols, [True] * len(cols))
    if order is not None:
        return order
    for c in reversed(list(cols)):
        if order is None:
            order = ops.sorted_order(c)
        else:
            order = ops.gather(order, ops.sorted_order(ops.gather(c, order)))
    return order


class GroupBy:
    _SUPPORTED = ("sum", "count", "mean", "min", "max", "var", "std")

    def __init__(self, df: DataFrame, by: Union[str, Sequence[str]]):
        self._df, self._by = df, ([by] if isinstance(by, str) else list(by))

    def _all(self, fn: str) -> DataFrame:
        """fn over every column that is not a key (python/cudf/cudf/core/groupby/groupby.py: GroupBy.sum / mean / ... = agg(fn))"""
        vals = [c for c in self._df.columns if c not in self._by]
        if not vals:
            raise ValueError("groupby: no value columns to aggregate")
        return self.agg({c: fn for c in vals})

    def sum(self) -> DataFrame:
        return self._all("sum")

    def count(self) -> DataFrame:
        return self._all("count")

    def mean(self) -> DataFrame:
        return self._all("mean")

    def min(self) -> DataFrame:
        return self._all("min")

    def max(self) -> DataFrame:
        return self._all("max")

    def var(self) -> DataFrame:
        return self._all("var")

    def std(self) -> DataFrame:
        return self._all("std")

    def agg(self, spec: Dict[str, Union[str, Sequence[str]]], _exact: bool = False) -> DataFrame:
        """{value column: "sum" | "count" | "mean" | "min" | "max" | "var" | "std" | [..]} -> one row per group, sorted by key
        (pandas' default sort=True).  Null keys are dropped (dropna=True), null values are skipped."""
        by = self._by
        k0 = self._df[by[0]]
        rep = rk = None
        if len(by) == 1 and k0.dtype.kind in "iu" and k0.dtype.itemsize in (4, 8):
            keys = k0
        elif _exact or len(by) > 8:  # exact encoding: dense row ids, one radix sort per key column
            keys, rep, _ = ops.groupby_keys_tables([self._df[b] for b in by])
        else:  # several key columns / floats / narrow types: one 8-byte row key (packed values or row hash)
            rk = ops.RowKeys([self._df[b] for b in by])
            keys = rk.col
        out = DataFrame()
        first = True
        for name, fns in spec.items():
            fns = [fns] if isinstance(fns, str) else list(fns)
            for f in fns:
                if f not in self._SUPPORTED:
                    raise NotImplementedError(f"groupby aggregation {f!r} is not on this path yet")
            k, s, cv, _ = ops.groupby_sum_count(keys, self._df[name])
            order = ops.sorted_order(k)
            if first:
                kk = ops.gather(k, order)
                if rk is not None:
                    kc = rk.key_columns(kk)
                    if kc is None:  # two different rows shared a 64-bit hash: start over with the exact encoding
                        return self.agg(spec, _exact=True)
                    for b, c in zip(by, kc):
                        out._cols[b] = c
                elif rep is None:
                    out._cols[by[0]] = kk
                else:
                    rows = ops.gather(rep, kk)
                    for b in by:
                        out._cols[b] = ops.gather(self._df[b], rows)
                first = False
            s, cv = ops.gather(s, order), ops.gather(cv, order)
            mn = mx = None
            if any(f in ("min", "max") for f in fns):
                k2, mn, mx, cv2 = ops.groupby_min_max(keys, self._df[name])
                o2 = ops.sorted_order(k2)
                mn, mx, cv2 = ops.gather(mn, o2), ops.gather(mx, o2), ops.gather(cv2, o2)
                mn.mask, mn.null_count = _valid_from_counts(cv2)    # a group without a valid value is null
                mx.mask, mx.null_count = mn.mask, mn.null_count
            var = std = None
            if any(f in ("var", "std") for f in fns):  # ddof = 1, groups come back in ascending key order
                _, var, std, _, _ = ops.groupby_var_std(keys, self._df[name])
            for f in fns:
                label = name if len(fns) == 1 and len(spec) >= 1 and all(isinstance(v, str) for v in spec.values()) else f"{name}_{f}"
                if f == "sum":
                    out._cols[label] = s
                elif f == "count":
                    out._cols[label] = cv
                elif f == "min":
                    out._cols[label] = mn
                elif f == "max":
                    out._cols[label] = mx
                elif f == "var":
                    out._cols[label] = var
                elif f == "std":
                    out._cols[label] = std
                else:  # MEAN = SUM / COUNT_VALID in double, on the device (hash_compound_agg_finalizer.cu:92-133)
                    m = Column.empty(np.float64, s.size)
                    L.check(_lib.gx_mean_from_sum(s.gx, s.data_ptr, cv.data_ptr, s.size, m.data_ptr, stream_ptr()), "gx_mean_from_sum")
                    m.mask, m.null_count = _valid_from_counts(cv)
                    out._cols[label] = m
        if (rep is not None or rk is not None) and len(out._cols):
            # results are in row-key order, pandas sorts by the key columns: order the (few) groups once
            perm = _lexicographic_order([out._cols[b] for b in by])
            for name in list(out._cols):
                out._cols[name] = ops.gather(out._cols[name], perm)
        return out
