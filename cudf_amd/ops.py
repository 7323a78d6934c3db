"""Host-side mirror of the libcudf operator interface for the hot path, over the C ABI.

Function names / argument meaning / error behaviour follow pylibcudf's thin wrappers
(python/pylibcudf/pylibcudf/{sorting.pyx:37-613, join.pyx:63-108, groupby.pyx:89,
reduce.pyx:48-157, copying.pyx gather}) which bind the C++ API named in SURVEY.md section 8b.
Every function here only validates, allocates (torch as the device allocator) and calls
``libcudf_amd.so``; there is no CPU implementation to fall back to.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib as L
from .column import Column, device_bytes, gx_dtype, ptr, stream_ptr, bitmask_words

JOIN_NO_MATCH = -(2**31)  # cudf::JoinNoMatch

_lib = L.lib


def _query(fn, *args):
    """Run the tmp==NULL size query of a gx_* entry point; returns required bytes."""
    nbytes = ctypes.c_size_t(0)
    L.check(fn(*args, None, ctypes.byref(nbytes), stream_ptr()), fn.__name__ + " (size query)")
    return nbytes.value


def _run(fn, *args):
    nbytes = _query(fn, *args)
    tmp = device_bytes(nbytes)
    nb = ctypes.c_size_t(nbytes)
    L.check(fn(*args, ptr(tmp), ctypes.byref(nb), stream_ptr()), fn.__name__)
    return tmp


def _check_sort_status(tmp: torch.Tensor):
    """Fail loudly if the device-side look-back protocol reported a timeout (never expected).  Only status 5 is a fault -- 3 means a
    bookkeeping mismatch of the hybrid path after which the LSD passes produced a correct output (include/cudf_amd/gx.h; the C++
    face draws the same line, cudf_amd/cpp/src/device_faults.cpp)."""
    st = ctypes.c_int(0)
    L.check(_lib.gx_sort_status(ptr(tmp), ctypes.byref(st), stream_ptr()), "gx_sort_status")
    if st.value == 5:
        raise L.GxError(f"radix sort look-back timed out (status {st.value})")


def _run_sort(fn, *args):
    """a sort whose status word this caller reads: opt into the recoverable fault form for this call (default: the kernel traps)"""
    _lib.gx_sort_set_fault_mode(1)
    try:
        return _run(fn, *args)
    finally:
        _lib.gx_sort_set_fault_mode(0)


def _dev_i64(value: int = 0) -> torch.Tensor:
    return torch.full((1,), value, dtype=torch.int64, device="cuda")


# ------------------------------------------------------------------------------------------------
# sorting  (cudf::sort / sorted_order / stable_sorted_order / sort_by_key: sorting.hpp:44-163)
# ------------------------------------------------------------------------------------------------

def sort(col: Column, ascending: bool = True, null_before: bool = True) -> Column:
    """cudf::sort of a one-column table (src/sort/sort.cu:52-67).  No nulls: keys-only radix sort;
    with nulls: sort_by_key(input, input) like the reference's general path (:31-50)."""
    if col.has_nulls():
        order = sorted_order(col, ascending, null_before)
        return gather(col, order)
    out = Column.empty(col.dtype, col.size)
    tmp = _run_sort(_lib.gx_sort_keys, col.gx, col.data_ptr, out.data_ptr, col.size, int(not ascending))
    _check_sort_status(tmp)
    return out


def sorted_order(col: Column, ascending: bool = True, null_before: bool = True) -> Column:
    """cudf::sorted_order == stable_sorted_order for one column (always stable here)."""
    out = Column.empty(np.int32, col.size)
    valid = col.mask_ptr if col.has_nulls() else None
    tmp = (_run if valid else _run_sort)(_lib.gx_sorted_order, col.gx, col.data_ptr, valid, col.size, col.null_count if valid else 0,
                                         int(not ascending), int(null_before), out.data_ptr)
    if not valid:
        _check_sort_status(tmp)  # plan header sits at the start of the scratch on the radix path
    return out


stable_sorted_order = sorted_order


def sorted_order_table(cols: Sequence[Column], ascending: Union[bool, Sequence[bool]] = True) -> Column:
    """cudf::sorted_order / stable_sorted_order of a table of numeric columns WITHOUT nulls (src/sort/sort_impl.cuh:61-93): the
    lexicographic order of the rows, stable; NaN equivalent and greatest in every column.  One word sort on a nested rank of the tuple
    (gx_sorted_order_table, cudf_amd/csrc/gx_order.hip) instead of one argsort + two random gathers per column."""
    cols = list(cols)
    if not 1 <= len(cols) <= 8:
        raise ValueError("sorted_order_table: 1 to 8 key columns")
    n = cols[0].size
    asc = [ascending] * len(cols) if isinstance(ascending, bool) else list(ascending)
    if len(asc) != len(cols):
        raise ValueError("Mismatch between number of columns and column order.")  # sort_impl.cuh:43-46
    for c in cols:
        if c.size != n:
            raise ValueError("sorted_order_table: columns of different lengths")
        if c.has_nulls():
            raise NotImplementedError("sorted_order_table: columns with nulls take the per-column path (DataFrame.sort_values)")
    out = Column.empty(np.int32, n)
    if n == 0:
        return out
    k = len(cols)
    dtypes = (ctypes.c_int * k)(*[c.gx for c in cols])
    datas = (ctypes.c_void_p * k)(*[c.data_ptr for c in cols])
    desc = (ctypes.c_int * k)(*[int(not a) for a in asc])
    tmp = _run_sort(_lib.gx_sorted_order_table, k, dtypes, datas, desc, n, out.data_ptr)
    _check_sort_status(tmp)
    return out


def sort_by_key(values: Sequence[Column], keys: Column, ascending: bool = True,
                null_before: bool = True) -> List[Column]:
    """cudf::sort_by_key: gather(values, sorted_order(keys)) (src/sort/sort.cu:31-50)."""
    for v in values:
        if v.size != keys.size:
            raise RuntimeError("Mismatch in number of rows for values and keys")  # sort.cu:39
    order = sorted_order(keys, ascending, null_before)
    return [gather(v, order) for v in values]


def gather(col: Column, gather_map: Column, nullify_out_of_bounds: bool = False) -> Column:
    """cudf::gather for a fixed-width column (copying.hpp; detail/gather.cuh:108-131,506-577)."""
    if gather_map.dtype != np.int32:
        raise TypeError("gather map must be INT32")
    n = gather_map.size
    need_mask = col.mask is not None or nullify_out_of_bounds
    out = Column.empty(col.dtype, n, nullable=need_mask)
    L.check(_lib.gx_gather(col.dtype.itemsize, col.data_ptr, col.mask_ptr, col.size, gather_map.data_ptr, n,
                           int(nullify_out_of_bounds), out.data_ptr, out.mask_ptr, stream_ptr()), "gx_gather")
    if need_mask:
        out.null_count = n - bitmask_count(out.mask, n)
    return out


def gather_global_rows(rows: Column, idx: Column, seg_counts: Sequence[int], seg_bases: Sequence[int]) -> Column:
    """out[j] = rows[idx[j]] + seg_bases[segment of idx[j]] as int64: received int32 local rows -> global row ids of
    the join pairs `idx` selects (segment k = the seg_counts[k] entries received from rank k)."""
    n = idx.size
    out = Column.empty(np.int64, n)
    cnt = (ctypes.c_int64 * len(seg_counts))(*[int(x) for x in seg_counts])
    bas = (ctypes.c_int64 * len(seg_bases))(*[int(x) for x in seg_bases])
    L.check(_lib.gx_gather_global_rows(rows.data_ptr, rows.size, idx.data_ptr, n, len(seg_counts), cnt, bas, out.data_ptr,
                                       stream_ptr()), "gx_gather_global_rows")
    return out


def bitmask_count(mask: torch.Tensor, nbits: int) -> int:
    cnt = _dev_i64()
    L.check(_lib.gx_bitmask_count(ptr(mask), 0, nbits, ptr(cnt), stream_ptr()), "gx_bitmask_count")
    return int(cnt.item())


# ------------------------------------------------------------------------------------------------
# hashing / partitioning
# ------------------------------------------------------------------------------------------------

def murmurhash3_x86_32(cols: Sequence[Column], seed: int = 0) -> Column:
    """cudf::hashing::murmurhash3_x86_32 row hash of fixed-width columns."""
    n = cols[0].size
    out = Column.empty(np.uint32, n)
    for k, c in enumerate(cols):
        L.check(_lib.gx_murmur3_32(c.gx, c.data_ptr, c.mask_ptr if c.has_nulls() else None, n, seed, int(k > 0),
                                   out.data_ptr, stream_ptr()), "gx_murmur3_32")
    return out


def identity_hash(cols: Sequence[Column]) -> Column:
    """Row hash over IdentityHash (cudf::hash_partition(..., hash_id::HASH_IDENTITY), partitioning.cu:852-889): the element cast to
    uint32, columns folded with hash_combine, null -> UINT32_MAX."""
    n = cols[0].size
    out = Column.empty(np.uint32, n)
    for k, c in enumerate(cols):
        L.check(_lib.gx_identity_hash_32(c.gx, c.data_ptr, c.mask_ptr if c.has_nulls() else None, n, int(k > 0), out.data_ptr, stream_ptr()),
                "gx_identity_hash_32")
    return out


def hash_partition_map(key_cols: Sequence[Column], num_partitions: int, seed: int = 0, hash_function: str = "murmur3") -> Tuple[Column, np.ndarray]:
    """cudf::hash_partition in index form: (gather map, offsets[num_partitions+1]).  hash_function: "murmur3" (HASH_MURMUR3) or
    "identity" (HASH_IDENTITY)."""
    if hash_function not in ("murmur3", "identity"):
        raise ValueError("Unsupported hash function in hash_partition")
    h = identity_hash(key_cols) if hash_function == "identity" else murmurhash3_x86_32(key_cols, seed)
    n = h.size
    out_map = Column.empty(np.int32, n)
    offs = torch.empty(num_partitions + 1, dtype=torch.int32, device="cuda")
    _run(_lib.gx_hash_partition_map, h.data_ptr, n, num_partitions, out_map.data_ptr, ptr(offs))
    return out_map, offs.cpu().numpy()


def partition_rows(col: Column, nparts: int, splitters: Optional[Sequence] = None, want_rows: bool = True):
    """One-pass partition of a key column into `nparts` groups (gx_partition_rows): by a hash that is independent of
    the join table's slot bits (splitters None; any nparts <= 16), or by range (nparts - 1 ascending splitters:
    destination = number of splitters <= key).  Returns (grouped keys, int32 row indices or None, nparts + 1 offsets).
    What a rank runs before the all-to-all of the distributed sort / join / groupby
    (cudf::hash_partition, cpp/src/partitioning/partitioning.cu:568-660)."""
    n = col.size
    out = Column.empty(col.dtype, n)
    rows = Column.empty(np.int32, n) if want_rows else None
    offs = torch.zeros(nparts + 1, dtype=torch.int64, device="cuda")
    mode = 0 if splitters is None else 1
    sp = None
    if splitters is not None:
        if len(splitters) != nparts - 1:
            raise ValueError("range partition needs nparts - 1 splitters")
        sp_np = np.asarray(list(splitters) + [0], dtype=col.dtype)   # host array the call reads synchronously
        sp = sp_np.ctypes.data_as(ctypes.c_void_p)
    _run(_lib.gx_partition_rows, col.gx, col.data_ptr, n, mode, nparts, sp, out.data_ptr, rows.data_ptr if rows is not None else None,
         ptr(offs))
    return out, rows, [int(x) for x in offs.cpu()]


def merge_sum_count(keys: Column, sums: Column, counts: Column):
    """Partial (key, sum, count) rows -> one row per distinct key, keys ascending: sorted_order of the keys, run heads
    and labels once, then two segmented reductions over the same grouping (the sort-based aggregation path:
    cpp/src/groupby/sort/aggregate.cpp:94-142).  Float sums in double-double (order independent)."""
    n = keys.size
    if n == 0:
        return Column.empty(keys.dtype, 0), Column.empty(sums.dtype, 0), Column.empty(np.int64, 0)
    order = sorted_order(keys)
    sk, ss, sc = gather(keys, order), gather(sums, order), gather(counts, order)
    heads = torch.empty(n, dtype=torch.uint8, device="cuda")
    L.check(_lib.gx_group_heads(sk.gx, sk.data_ptr, None, None, n, 0, ptr(heads), stream_ptr()), "gx_group_heads")
    labels = torch.empty(n, dtype=torch.int32, device="cuda")
    offsets = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    ng = _dev_i64()
    _run(_lib.gx_group_offsets, ptr(heads), n, ptr(labels), ptr(offsets), None, ptr(ng))
    g = int(ng.item())
    out_s = Column.empty(sums.dtype if sums.dtype.kind == "f" else np.int64, g)
    out_c = Column.empty(np.int64, g)
    _run(_lib.gx_segmented_reduce, ss.gx, ss.data_ptr, None, ptr(heads), ptr(labels), n, L.OP_SUM, out_s.data_ptr, None)
    _run(_lib.gx_segmented_reduce, sc.gx, sc.data_ptr, None, ptr(heads), ptr(labels), n, L.OP_SUM, out_c.data_ptr, None)
    starts = Column(offsets[:g].contiguous().view(torch.uint8), np.int32, g)
    return gather(sk, starts), out_s, out_c


# ------------------------------------------------------------------------------------------------
# hash join  (cudf::hash_join / cudf::inner_join: join/hash_join.hpp:70-444, join/join.hpp:160-166)
# ------------------------------------------------------------------------------------------------

def _join_key(col: Column) -> Column:
    if col.dtype.itemsize not in (4, 8):
        raise TypeError("join key must be a 4- or 8-byte fixed-width column")  # data_type_error
    if col.dtype.kind == "f":
        # row equality of the reference: -0.0 == +0.0 and NaN == NaN whatever the payload
        # (detail/row_operator/common_utils.cuh:215-220); gx_pack_keys normalises both, the table compares bits
        key = pack_keys([col])
        key.mask, key.null_count = col.mask, col.null_count
        return key
    return col


class HashJoin:
    """cudf::hash_join: build once on `right`, probe many (hash_join.hpp:95-125).  The object
    views the build column: keep it alive (hash_join.hpp:83-84)."""

    def __init__(self, right: Column, nulls_equal: bool = True, load_factor: float = 0.5):
        if not (0.0 < load_factor <= 1.0):
            raise ValueError("Invalid load factor: must be greater than 0 and less than or equal to 1.")
        self.build_dtype = right.dtype
        self.build = right = _join_key(right)
        self.nulls_equal = nulls_equal
        self.key_size = right.dtype.itemsize
        self.load_factor = load_factor
        self.table_bytes = _lib.gx_join_table_bytes(self.key_size, right.size, load_factor)
        self.table = device_bytes(self.table_bytes)
        valid = right.mask_ptr if right.has_nulls() else None
        if (valid is None and right.size >= (1 << 20)
                and _lib.gx_join_partition_bits(self.key_size, self.table_bytes) > 0):
            # large build side: partition the rows first so the inserts hit L2-resident sub-tables
            _run(_lib.gx_join_build_partitioned, self.key_size, right.data_ptr, right.size, ptr(self.table),
                 self.table_bytes, load_factor)
        else:
            L.check(_lib.gx_join_build(self.key_size, right.data_ptr, valid, right.size, ptr(self.table),
                                       self.table_bytes, load_factor, stream_ptr()), "gx_join_build")

    def _check(self, left: Column) -> Column:
        """type check (hash_join.cu:56-58); returns the probe column in the table's key space"""
        if left.dtype != self.build_dtype:
            raise TypeError("Mismatch in joining column data types")
        return _join_key(left) if left.size else left

    def inner_join_size(self, left: Column) -> int:
        left = self._check(left)
        if left.size == 0 or self.build.size == 0:
            return 0
        cnt = _dev_i64()
        valid = left.mask_ptr if left.has_nulls() else None
        L.check(_lib.gx_join_count(self.key_size, left.data_ptr, valid, left.size, ptr(self.table),
                                   self.table_bytes, ptr(cnt), stream_ptr()), "gx_join_count")
        total = int(cnt.item())
        if self.nulls_equal:
            total += left.null_count * self.build.null_count if left.has_nulls() and self.build.has_nulls() else 0
        return total

    PARTITIONED_MIN_ROWS = 1 << 22

    def _probe(self, left: Column, capacity: int, left_outer: int):
        """left_outer: gx_join_probe's flags (bit 0 left outer, bit 1 null probe rows emit nothing)"""
        lo = Column.empty(np.int32, capacity)
        ro = Column.empty(np.int32, capacity)
        cur = _dev_i64()
        valid = left.mask_ptr if left.has_nulls() else None
        if (valid is None and left.size >= self.PARTITIONED_MIN_ROWS
                and _lib.gx_join_partition_bits(self.key_size, self.table_bytes) > 0):
            # large probe against a table far beyond the L2s: partition the probe rows first
            _run(_lib.gx_join_probe_partitioned, self.key_size, left.data_ptr, left.size, ptr(self.table),
                 self.table_bytes, int(left_outer) & 1, lo.data_ptr, ro.data_ptr, capacity, ptr(cur))
            return lo, ro, int(cur.item())
        L.check(_lib.gx_join_probe(self.key_size, left.data_ptr, valid, left.size, ptr(self.table), self.table_bytes,
                                   int(left_outer), lo.data_ptr, ro.data_ptr, capacity, ptr(cur), stream_ptr()),
                "gx_join_probe")
        return lo, ro, int(cur.item())

    def inner_join(self, left: Column, output_size: Optional[int] = None) -> Tuple[Column, Column]:
        """(left_indices, right_indices), order unspecified (join.hpp:131-134)."""
        left = self._check(left)
        if left.size == 0 or self.build.size == 0:  # trivial joins (hash_join.cu:32-45)
            return Column.empty(np.int32, 0), Column.empty(np.int32, 0)
        # optimistic single pass: with distinct build keys (the common PK-FK case) matches <= probe
        # rows; 288 GB of HBM makes that allocation cheaper than the reference's count pass
        capacity = output_size if output_size is not None else left.size
        lo, ro, total = self._probe(left, capacity, False)
        if total > capacity:  # duplicate build keys blew the guess: size is now known exactly
            lo, ro, total = self._probe(left, total, False)
        lo.size = ro.size = total
        if self.nulls_equal and left.has_nulls() and self.build.has_nulls():
            lo, ro = _append_null_cross(lo, ro, left, self.build)
        return lo, ro

    def left_join(self, left: Column) -> Tuple[Column, Column]:
        left = self._check(left)
        if left.size == 0:
            return Column.empty(np.int32, 0), Column.empty(np.int32, 0)
        # null == null (hash_join.cu:77-84): a null left row matches every null build row instead of (i, JoinNoMatch)
        cross = self.nulls_equal and left.has_nulls() and self.build.has_nulls()
        flags = 3 if cross else 1
        lo, ro, total = self._probe(left, left.size, flags)
        if total > left.size:
            lo, ro, total = self._probe(left, total, flags)
        lo.size = ro.size = total
        if cross:
            lo, ro = _append_null_cross(lo, ro, left, self.build)
        return lo, ro


    def lookup(self, left: Column) -> Column:
        """cudf::distinct_hash_join::left_join (distinct_hash_join.hpp:96-116): for DISTINCT build keys,
        the build row matching each left row, in left order, JoinNoMatch (INT32_MIN) where there is none.
        (With duplicate build keys it returns one of the matching rows.)"""
        left = self._check(left)
        out = Column.empty(np.int32, left.size)
        if left.size == 0:
            return out
        if self.build.size == 0:   # nothing to look up: JoinNoMatch everywhere (a host-built column: no kernel of ours or of torch's)
            return Column.from_numpy(np.full(left.size, -(2 ** 31), np.int32))
        valid = left.mask_ptr if left.has_nulls() else None
        L.check(_lib.gx_join_lookup(self.key_size, left.data_ptr, valid, left.size, ptr(self.table), self.table_bytes,
                                    out.data_ptr, stream_ptr()), "gx_join_lookup")
        if self.nulls_equal and left.has_nulls() and self.build.has_nulls():
            # null == null: every null left row matches the (single, keys are distinct) null build row.  A rare path, patched on the
            # host like the C++ face does (cpp/src/join.cpp: the null x null pairs are composed on the host and copied)
            rv = int(np.nonzero(~self.build.valid_numpy())[0][0])
            o = out.to_numpy()
            o[~left.valid_numpy()] = rv
            out = Column.from_numpy(o)
        return out

    def _filter(self, left: Column, anti: bool) -> Column:
        left = self._check(left)
        out = Column.empty(np.int32, left.size)
        if left.size == 0:
            return out
        if self.build.size == 0:  # nothing can match
            if anti:
                L.check(_lib.gx_sequence_i32(out.data_ptr, left.size, 0, stream_ptr()), "gx_sequence_i32")
            else:
                out.size = 0
            return out
        cnt = _dev_i64()
        valid = left.mask_ptr if left.has_nulls() else None
        null_matches = int(self.nulls_equal and self.build.has_nulls())
        _run(_lib.gx_join_filter, self.key_size, left.data_ptr, valid, left.size, ptr(self.table), self.table_bytes,
             int(anti), null_matches, out.data_ptr, ptr(cnt))
        out.size = int(cnt.item())
        return out

    def semi_join(self, left: Column) -> Column:
        """cudf::filtered_join::semi_join (filtered_join.hpp:96-116): ascending left rows with a match."""
        return self._filter(left, False)

    def anti_join(self, left: Column) -> Column:
        """cudf::filtered_join::anti_join (filtered_join.hpp:118-139): ascending left rows without a match."""
        return self._filter(left, True)


def _append_null_cross(lo: Column, ro: Column, left: Column, right: Column):
    """null_equality::EQUAL for a single nullable key: every null left row matches every null
    right row.  Rare path: the cross product is composed on the HOST from the two validity masks and copied behind the probe's
    pairs -- what the C++ face does (cudf_amd/cpp/src/join.cpp, `cross`); no torch kernel takes part."""
    lv = np.nonzero(~left.valid_numpy())[0].astype(np.int32)
    rv = np.nonzero(~right.valid_numpy())[0].astype(np.int32)
    cross = len(lv) * len(rv)
    n = lo.size + cross
    outs = []
    for src, add in ((lo, np.repeat(lv, len(rv))), (ro, np.tile(rv, len(lv)))):
        out = Column.empty(np.int32, n)
        if lo.size:
            L.check(_lib.gx_copy_bytes(src.data_ptr, out.data_ptr, lo.size * 4, stream_ptr()), "gx_copy_bytes")
        if cross:
            tail = Column.from_numpy(np.ascontiguousarray(add))
            L.check(_lib.gx_copy_bytes(tail.data_ptr, ctypes.c_void_p(out.data.data_ptr() + lo.size * 4), cross * 4, stream_ptr()), "gx_copy_bytes")
            torch.cuda.current_stream().synchronize()   # `tail` may be recycled once the copy has run
        outs.append(out)
    return outs[0], outs[1]


def inner_join(left: Column, right: Column, nulls_equal: bool = True) -> Tuple[Column, Column]:
    """cudf::inner_join (src/join/join.cu:27-60): build on the smaller side, swap the pair back."""
    if left.dtype != right.dtype:
        raise TypeError("Mismatch in joining column data types")
    if right.size > left.size:
        hj = HashJoin(left, nulls_equal)
        r, l = hj.inner_join(right)
        return l, r
    return HashJoin(right, nulls_equal).inner_join(left)


def left_semi_join(left: Column, right: Column, nulls_equal: bool = True) -> Column:
    if left.dtype != right.dtype:
        raise TypeError("Mismatch in joining column data types")
    return HashJoin(right, nulls_equal).semi_join(left)


def left_anti_join(left: Column, right: Column, nulls_equal: bool = True) -> Column:
    if left.dtype != right.dtype:
        raise TypeError("Mismatch in joining column data types")
    return HashJoin(right, nulls_equal).anti_join(left)


def left_join(left: Column, right: Column, nulls_equal: bool = True) -> Tuple[Column, Column]:
    if left.dtype != right.dtype:
        raise TypeError("Mismatch in joining column data types")
    return HashJoin(right, nulls_equal).left_join(left)


def _append_unmatched_right(lo: Column, ro: Column, right_rows: int) -> Tuple[Column, Column]:
    """cudf::full_join's complement step (src/join/join_utils.cu:86-157; gx_join_complement): behind the left-join pairs, one
    (JoinNoMatch, r) pair for every right row r that occurs in no pair.  Returns the grown pair arrays."""
    n0 = lo.size
    ol, orr = Column.empty(np.int32, n0 + right_rows), Column.empty(np.int32, n0 + right_rows)
    if n0:
        L.check(_lib.gx_copy_bytes(lo.data_ptr, ol.data_ptr, n0 * 4, stream_ptr()), "gx_copy_bytes")
        L.check(_lib.gx_copy_bytes(ro.data_ptr, orr.data_ptr, n0 * 4, stream_ptr()), "gx_copy_bytes")
    cur = _dev_i64(n0)
    _run(_lib.gx_join_complement, ro.data_ptr, n0, right_rows, ol.data_ptr, orr.data_ptr, n0 + right_rows, ptr(cur))
    ol.size = orr.size = int(cur.item())
    return ol, orr


def full_join(left: Column, right: Column, nulls_equal: bool = True) -> Tuple[Column, Column]:
    """cudf::full_join (join.hpp:240-246; src/join/hash_join/hash_join.cu:168-199): the left join's pairs, then every right row without
    a partner as (JoinNoMatch, r)."""
    lo, ro = left_join(left, right, nulls_equal)
    return _append_unmatched_right(lo, ro, right.size)


# ------------------------------------------------------------------------------------------------
# multi-column keys: rows -> one fixed-width key  (gx_pack_keys / gx_dense_rank; the reference compares
# whole rows inside its hash tables: detail/row_operator/primitive_row_operators.cuh:207-274)
# ------------------------------------------------------------------------------------------------

def pack_keys(cols: Sequence[Column]) -> Column:
    """Concatenate columns whose widths sum to <= 8 bytes into one uint64 key column (no nulls)."""
    n = cols[0].size
    out = Column.empty(np.uint64, n)
    ptrs = (ctypes.c_void_p * len(cols))(*[c.data_ptr.value or 0 for c in cols])
    dts = (ctypes.c_int * len(cols))(*[c.gx for c in cols])
    L.check(_lib.gx_pack_keys(len(cols), ptrs, dts, n, out.data_ptr, stream_ptr()), "gx_pack_keys")
    return out


def _col_ptrs(cols: Sequence[Column]):
    return ((ctypes.c_void_p * len(cols))(*[c.data_ptr.value or 0 for c in cols]), (ctypes.c_int * len(cols))(*[c.gx for c in cols]))


def hash_rows64(cols: Sequence[Column], seed: int = 0) -> Column:
    """One uint64 hash per row of the key columns (floats normalised: -0.0 == +0.0, NaN == NaN).  Equal rows hash
    equal; a result obtained through the hashes is certified by rows_mismatch_count == 0."""
    n = cols[0].size
    out = Column.empty(np.uint64, n)
    ptrs, dts = _col_ptrs(cols)
    L.check(_lib.gx_hash_rows64(len(cols), ptrs, dts, n, seed, out.data_ptr, stream_ptr()), "gx_hash_rows64")
    return out


def rows_mismatch_count(lcols: Sequence[Column], rcols: Sequence[Column], lidx: Optional[Column], ridx: Optional[Column],
                        npairs: int) -> int:
    """Number of pairs (lidx[i], ridx[i]) -- None = row i, negative = no row -- whose rows differ in some column."""
    lp, dts = _col_ptrs(lcols)
    rp, _ = _col_ptrs(rcols)
    cnt = _dev_i64()
    L.check(_lib.gx_rows_mismatch_count(len(lcols), lp, rp, dts, lidx.data_ptr if lidx is not None else None,
                                        ridx.data_ptr if ridx is not None else None, npairs, ptr(cnt), stream_ptr()),
            "gx_rows_mismatch_count")
    return int(cnt.item())


class RowKeys:
    """ONE 8-byte key per row of a multi-column key table, for the single-key groupby kernels, without sorting:
    the packed column values when their widths sum to <= 8 bytes (exact), else a 64-bit row hash.  Rows holding a
    null key get a null key (they belong to no group: null_policy::EXCLUDE).  key_columns() turns the distinct keys
    of a result back into key columns and -- for hashed keys -- certifies that no two different rows shared a hash."""

    def __init__(self, cols: Sequence[Column]):
        self.cols = list(cols)
        n = self.cols[0].size
        self.bare = [Column(c.data, c.dtype, c.size) for c in self.cols]
        self.exact = sum(c.dtype.itemsize for c in self.cols) <= 8
        self.col = pack_keys(self.bare) if self.exact else hash_rows64(self.bare)
        self.col.mask, self.col.null_count = _and_masks(self.cols, n)

    def key_columns(self, distinct: Column) -> Optional[List[Column]]:
        """Key columns of the groups whose row keys are `distinct` (in that order).  Packed keys are unpacked.  Hashed
        keys: per key column one streaming groupby MIN + MAX by hash; MIN == MAX in every group and column means every
        row of a group carries the same key values (the values returned), i.e. no 64-bit collision; None otherwise
        (the caller then encodes through gx_dense_rank)."""
        g = distinct.size
        if self.exact:
            outs = [Column.empty(c.dtype, g) for c in self.cols]
            ptrs = (ctypes.c_void_p * len(outs))(*[c.data_ptr.value or 0 for c in outs])
            dts = (ctypes.c_int * len(outs))(*[c.gx for c in outs])
            L.check(_lib.gx_unpack_keys(len(outs), ptrs, dts, g, distinct.data_ptr, stream_ptr()), "gx_unpack_keys")
            return outs
        if g == 0:
            return [Column.empty(c.dtype, 0) for c in self.cols]
        mns, mxs = [], []
        for c in self.bare:
            hk, mn, mx, _ = groupby_min_max(self.col, c, max_groups_hint=g)
            if hk.size != g:
                return None
            o = sorted_order(hk)                                   # ascending-hash order
            mns.append(gather(mn, o))
            mxs.append(gather(mx, o))
        if rows_mismatch_count(mns, mxs, None, None, g) != 0:
            return None
        back = sorted_order(sorted_order(distinct))                # ascending-hash position of every entry of `distinct`
        return [gather(m, back) for m in mns]


def dense_rank(col: Column):
    """(ids int32 column, first row of every id, number of ids): equal values share an id, null == null
    has its own id (last), ids ascend with the value."""
    n = col.size
    ids, rep = Column.empty(np.int32, n), Column.empty(np.int32, n)
    ng = _dev_i64()
    _run(_lib.gx_dense_rank, col.gx, col.data_ptr, col.mask_ptr if col.has_nulls() else None, n,
         col.null_count if col.has_nulls() else 0, ids.data_ptr, rep.data_ptr, ptr(ng))
    g = int(ng.item())
    rep.size = g
    return ids, rep, g


def _and_masks(cols: Sequence[Column], n: int):
    """(mask words tensor or None, null count) of the AND of the columns' validity."""
    masks = [c for c in cols if c.has_nulls()]
    if not masks:
        return None, 0
    out = torch.zeros(bitmask_words(n), dtype=torch.int32, device="cuda")
    arr = (ctypes.c_void_p * len(masks))(*[c.mask_ptr.value for c in masks])
    cnt = _dev_i64()
    L.check(_lib.gx_bitmask_and(arr, len(masks), n, ptr(out), ptr(cnt), stream_ptr()), "gx_bitmask_and")
    return out, n - int(cnt.item())


def concat_columns(a: Column, b: Column) -> Column:
    """cudf::concatenate of two columns of one type (data memcpy + gx_bitmask_copy for the validity)."""
    if a.dtype != b.dtype:
        raise TypeError("Mismatch in joining column data types")
    n = a.size + b.size
    if n > 2**31 - 1:
        raise OverflowError("concatenated key columns exceed cudf::size_type")
    w = a.dtype.itemsize
    out = Column.empty(a.dtype, n, nullable=a.has_nulls() or b.has_nulls())
    out.data[: a.size * w].copy_(a.data[: a.size * w])
    out.data[a.size * w: n * w].copy_(b.data[: b.size * w])
    if out.mask is not None:
        L.check(_lib.gx_bitmask_copy(ptr(out.mask), 0, a.mask_ptr if a.has_nulls() else None, 0, a.size, stream_ptr()),
                "gx_bitmask_copy")
        L.check(_lib.gx_bitmask_copy(ptr(out.mask), a.size, b.mask_ptr if b.has_nulls() else None, 0, b.size,
                                     stream_ptr()), "gx_bitmask_copy")
        out.null_count = a.null_count + b.null_count
    return out


def _slice_rows(col: Column, begin: int, end: int) -> Column:
    """Rows [begin, end) of a column WITHOUT nulls as a new column sharing no storage."""
    w = col.dtype.itemsize
    out = Column.empty(col.dtype, end - begin)
    out.data[: (end - begin) * w].copy_(col.data[begin * w: end * w])
    return out


def encode_rows(tables: Sequence[Sequence[Column]], nulls_equal: bool = True, dense: bool = False):
    """Encode the rows of 1 or 2 tables (same schema) into ONE key column per table such that two rows
    get equal keys iff they compare equal column by column (null == null, NaN == NaN, -0.0 == +0.0).
    Rows holding a null are marked null in the result when nulls_equal is False (they can then match
    nothing).  dense=True (one table): keys are dense int32 ids; also returns (first row per id, #ids).

    Path 1 (widths sum to <= 8 bytes, no nulls): gx_pack_keys, exact, one streaming pass.
    Path 2: every column -> dense ids over the CONCATENATION of the tables (so ids agree across them),
    then (ids so far, next column's ids) pairs are packed and ranked again."""
    ncols = len(tables[0])
    for t in tables[1:]:
        if len(t) != ncols:
            raise ValueError("Mismatch in number of columns to be joined on")
        for a, b in zip(tables[0], t):
            if a.dtype != b.dtype:
                raise TypeError("Mismatch in joining column data types")
    sizes = [t[0].size for t in tables]
    any_nulls = any(c.has_nulls() for t in tables for c in t)
    width = sum(c.dtype.itemsize for c in tables[0])
    if width <= 8 and not any_nulls and not dense:
        return [pack_keys(t) for t in tables]
    # ---- one tall table
    if len(tables) == 2:
        cols = [concat_columns(a, b) for a, b in zip(tables[0], tables[1])]
    else:
        cols = list(tables[0])
    n = cols[0].size
    if width <= 8 and not any_nulls:
        ids, rep, g = dense_rank(pack_keys(cols))
    else:
        def as32(c):  # a column the pair-packing can take as it is: 4 bytes, no nulls, bitwise equality
            return c if (c.dtype.itemsize == 4 and not c.has_nulls() and c.dtype.kind in "iu") else dense_rank(c)[0]
        cur = None
        ids = rep = None
        g = 0
        for k, c in enumerate(cols):
            if cur is None:
                if ncols == 1:
                    ids, rep, g = dense_rank(c)
                cur = as32(c) if ncols > 1 else ids
                continue
            pair = pack_keys([cur, as32(c)])
            if k == ncols - 1 and not dense:
                ids = pair           # the last packing need not be dense
            else:
                ids, rep, g = dense_rank(pair)
                cur = ids
    mask, nulls = (None, 0) if nulls_equal else _and_masks(cols, n)
    outs = []
    begin = 0
    for sz in sizes:
        o = _slice_rows(ids, begin, begin + sz) if len(sizes) > 1 else ids
        if mask is not None:
            words = torch.zeros(bitmask_words(sz), dtype=torch.int32, device="cuda")
            L.check(_lib.gx_bitmask_copy(ptr(words), 0, ptr(mask), begin, sz, stream_ptr()), "gx_bitmask_copy")
            cnt = _dev_i64()
            L.check(_lib.gx_bitmask_count(ptr(words), 0, sz, ptr(cnt), stream_ptr()), "gx_bitmask_count")
            o.mask, o.null_count = words, sz - int(cnt.item())
        outs.append(o)
        begin += sz
    if dense:
        return outs[0], rep, g
    return outs


def _hashed_join(left: Sequence[Column], right: Sequence[Column], nulls_equal: bool, join):
    """Rows wider than 8 bytes: join on a 64-bit row hash (one pass over the key columns per side), then compare the
    real columns of every emitted pair.  Equal rows always hash equal, so no pair is missing; a pair that fails the
    comparison is a 64-bit collision and sends the call to the exact encoding (None).  With nulls the fast path
    needs null != null (a row holding a null matches nothing: its hash is marked null)."""
    if len(left) != len(right):
        raise ValueError("Mismatch in number of columns to be joined on")
    for a, b in zip(left, right):
        if a.dtype != b.dtype:
            raise TypeError("Mismatch in joining column data types")
    width = sum(c.dtype.itemsize for c in left)
    any_nulls = any(c.has_nulls() for c in list(left) + list(right))
    if width <= 8 or (any_nulls and nulls_equal):
        return None
    keys = []
    for t in (left, right):
        bare = [Column(c.data, c.dtype, c.size) for c in t]
        k = hash_rows64(bare)
        k.mask, k.null_count = _and_masks(t, t[0].size)
        keys.append((k, bare))
    (lk, lbare), (rk, rbare) = keys
    l, r = join(lk, rk, False)
    if rows_mismatch_count(lbare, rbare, l, r, l.size) != 0:
        return None
    return l, r


def inner_join_tables(left: Sequence[Column], right: Sequence[Column], nulls_equal: bool = True):
    """cudf::inner_join on key TABLES (join.hpp:160-166): rows are encoded, then the single-key join."""
    if len(left) == 1 and len(right) == 1:
        return inner_join(left[0], right[0], nulls_equal)
    res = _hashed_join(left, right, nulls_equal, inner_join)
    if res is not None:
        return res
    lk, rk = encode_rows([left, right], nulls_equal)
    return inner_join(lk, rk, nulls_equal)


def left_join_tables(left: Sequence[Column], right: Sequence[Column], nulls_equal: bool = True):
    if len(left) == 1 and len(right) == 1:
        return left_join(left[0], right[0], nulls_equal)
    res = _hashed_join(left, right, nulls_equal, left_join)
    if res is not None:
        return res
    lk, rk = encode_rows([left, right], nulls_equal)
    return left_join(lk, rk, nulls_equal)


def full_join_tables(left: Sequence[Column], right: Sequence[Column], nulls_equal: bool = True):
    """cudf::full_join on key tables: the left join of the tables + the unmatched right rows"""
    lo, ro = left_join_tables(left, right, nulls_equal)
    return _append_unmatched_right(lo, ro, right[0].size)


def groupby_keys_tables(keys: Sequence[Column]):
    """Multi-column groupby keys, exact encoding -> (dense int32 id column with the rows holding a null key marked
    null, first row of every id, number of ids): aggregate by id, gather the key columns by first row.  One radix
    sort per key column (gx_dense_rank); the operators use RowKeys (one hash pass) and come here on a collision."""
    ids, rep, g = encode_rows([keys], nulls_equal=False, dense=True)
    return ids, rep, g


# ------------------------------------------------------------------------------------------------
# groupby  (cudf::groupby::groupby::aggregate / scan: groupby.hpp:89-240)
# ------------------------------------------------------------------------------------------------

def groupby_sum_count(keys: Column, values: Column, max_groups_hint: int = 1 << 20):
    """Hash groupby with SUM + COUNT_VALID + COUNT_ALL of one values column.
    Returns (keys, sum, count_valid, count_all) columns in unspecified group order."""
    if keys.size != values.size:
        raise RuntimeError("Size mismatch between request values and groupby keys.")  # groupby.cu:230
    if keys.dtype.itemsize not in (4, 8) or keys.dtype.kind not in "iu":
        raise TypeError("groupby key must be a 32/64-bit integer column")
    n = keys.size
    sum_dt = np.dtype(values.dtype if values.dtype.kind == "f" else np.int64)
    max_groups = max(1, min(n, max_groups_hint))
    while True:
        ok = Column.empty(keys.dtype, max_groups)
        osum = Column.empty(sum_dt, max_groups)
        ocv = Column.empty(np.int32, max_groups)
        oca = Column.empty(np.int32, max_groups)
        ng = _dev_i64()
        _run(_lib.gx_groupby_sum_count, keys.gx, keys.data_ptr, keys.mask_ptr if keys.has_nulls() else None,
             values.gx, values.data_ptr, values.mask_ptr if values.has_nulls() else None, n, max_groups,
             ok.data_ptr, osum.data_ptr, ocv.data_ptr, oca.data_ptr, ptr(ng))
        g = int(ng.item())
        if 0 <= g <= max_groups:
            break
        if max_groups >= n:
            raise RuntimeError("groupby table overflow")
        max_groups = min(n, max_groups * 8)
    for c in (ok, osum, ocv, oca):
        c.size = g
    return ok, osum, ocv, oca


WIDE_GROUPBY_MIN_ROWS = 1 << 18  # below: the partition pass is not worth its 4096 slots
_SIGNED_OF_WIDTH = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}


def _widen_i64(col: Column) -> Column:
    """an integer column as 8-byte words (sign / zero extended): what gx_groupby_sum_count_wide compares"""
    if col.dtype.itemsize == 8:
        return Column(col.data, col.dtype, col.size)
    # any injective map to 8-byte words will do (the words are only hashed and compared): view as the SIGNED type of the same
    # width and sign-extend; narrowing back keeps the low bytes
    t = col.data[: col.size * col.dtype.itemsize].view(_SIGNED_OF_WIDTH[col.dtype.itemsize])
    out = Column.empty(np.int64, col.size)
    out.data[: col.size * 8].view(torch.int64).copy_(t)
    return out


def _groupby_sum_count_wide(keys: Sequence[Column], values: Column, max_groups_hint: int = 1 << 20):
    """groupby on 2..4 integer key columns in ONE partition pass with the rows compared inside the LDS tables
    (gx_groupby_sum_count_wide).  None: not applicable / the device asked for the fallback."""
    n = values.size
    if not (2 <= len(keys) <= 4) or n < WIDE_GROUPBY_MIN_ROWS or values.has_nulls() or values.dtype.kind not in "if" or \
            values.dtype.itemsize not in (4, 8) or any(k.has_nulls() or k.dtype.kind not in "iu" for k in keys):
        return None
    wide = [_widen_i64(k) for k in keys]
    sum_dt = np.dtype(values.dtype if values.dtype.kind == "f" else np.int64)
    max_groups = max(1, min(n, max_groups_hint))
    kp = (ctypes.c_void_p * len(wide))(*[k.data_ptr.value or 0 for k in wide])
    while True:
        outs = [Column.empty(np.int64, max_groups) for _ in wide]
        osum, ocv = Column.empty(sum_dt, max_groups), Column.empty(np.int32, max_groups)
        op = (ctypes.c_void_p * len(outs))(*[o.data_ptr.value or 0 for o in outs])
        ng = _dev_i64()
        _run(_lib.gx_groupby_sum_count_wide, len(wide), kp, values.gx, values.data_ptr, n, max_groups, op, osum.data_ptr, ocv.data_ptr, ptr(ng))
        g = int(ng.item())
        if g == -2:
            return None
        if g >= 0:
            break
        if max_groups >= n:
            raise RuntimeError("groupby table overflow")
        max_groups = min(n, max_groups * 8)
    kcols = []
    for o, k in zip(outs, keys):
        o.size = g
        if k.dtype.itemsize == 8:
            kcols.append(Column(o.data, k.dtype, g))
        else:  # narrow the words back to the column's type
            c = Column.empty(k.dtype, g)
            c.data[: g * k.dtype.itemsize].view(_SIGNED_OF_WIDTH[k.dtype.itemsize]).copy_(o.data[: g * 8].view(torch.int64))
            kcols.append(c)
    osum.size = ocv.size = g
    return kcols, osum, ocv, ocv


def groupby_sum_count_tables(keys: Sequence[Column], values: Column, exact: bool = False):
    """groupby(keys = several columns).agg(SUM, COUNT_VALID, COUNT_ALL): the rows' 8-byte keys (RowKeys: packed values
    or row hash) go through the single-key hash groupby; the key columns come back from the distinct keys.
    exact=True, or a 64-bit collision: dense ids (gx_dense_rank), aggregate by id, gather the key columns by the
    first row of every id.  Returns ([key columns], sum, count_valid, count_all); rows with a null in any key are dropped."""
    if len(keys) == 1 and keys[0].dtype.itemsize in (4, 8) and keys[0].dtype.kind in "iu":
        k, s, cv, ca = groupby_sum_count(keys[0], values)
        return [k], s, cv, ca
    if keys[0].size != values.size:
        raise RuntimeError("Size mismatch between request values and groupby keys.")
    if not exact:
        r = _groupby_sum_count_wide(keys, values)
        if r is not None:
            return r
    if not exact and len(keys) <= 8:
        rk = RowKeys(keys)
        hk, s, cv, ca = groupby_sum_count(rk.col, values)
        kc = rk.key_columns(hk)
        if kc is not None:
            return kc, s, cv, ca
    ids, rep, g = groupby_keys_tables(keys)
    ok, s, cv, ca = groupby_sum_count(ids, values, max_groups_hint=max(g, 1))
    rows = gather(rep, ok)  # id -> its first row
    return [gather(k, rows) for k in keys], s, cv, ca


def groupby_min_max(keys: Column, values: Column, max_groups_hint: int = 1 << 20):
    """Hash groupby MIN + MAX + COUNT_VALID of one values column -> (keys, min, max, count_valid);
    min/max have the values' dtype and are meaningful where count_valid > 0."""
    if keys.size != values.size:
        raise RuntimeError("Size mismatch between request values and groupby keys.")
    if keys.dtype.itemsize not in (4, 8) or keys.dtype.kind not in "iu":
        raise TypeError("groupby key must be a 32/64-bit integer column")
    n = keys.size
    max_groups = max(1, min(n, max_groups_hint))
    while True:
        ok = Column.empty(keys.dtype, max_groups)
        omin, omax = Column.empty(values.dtype, max_groups), Column.empty(values.dtype, max_groups)
        ocv = Column.empty(np.int32, max_groups)
        ng = _dev_i64()
        _run(_lib.gx_groupby_min_max, keys.gx, keys.data_ptr, keys.mask_ptr if keys.has_nulls() else None,
             values.gx, values.data_ptr, values.mask_ptr if values.has_nulls() else None, n, max_groups,
             ok.data_ptr, omin.data_ptr, omax.data_ptr, ocv.data_ptr, ptr(ng))
        g = int(ng.item())
        if 0 <= g <= max_groups:
            break
        if max_groups >= n:
            raise RuntimeError("groupby table overflow")
        max_groups = min(n, max_groups * 8)
    for c in (ok, omin, omax, ocv):
        c.size = g
    return ok, omin, omax, ocv


def _sum_dtype(dt) -> np.dtype:
    return np.dtype(dt if np.dtype(dt).kind == "f" else np.int64)


def groupby_var_std(keys: Column, values: Column, ddof: int = 1):
    """groupby VARIANCE / STD / M2 the way the reference's hash path computes them: SUM, COUNT_VALID and
    SUM_OF_SQUARES per group, then M2 = sum_sqr - sum^2/count, VAR = M2/(count - ddof), STD = sqrt(VAR)
    (src/groupby/common/m2_var_std.cu:44-61,153-190).  Returns (keys, var, std, m2, count_valid) with
    the groups in ascending key order; var / std are null where count - ddof <= 0."""
    n = keys.size
    k, s, cv, _ = groupby_sum_count(keys, values)
    sq = Column(device_bytes(n * _sum_dtype(values.dtype).itemsize), _sum_dtype(values.dtype), n, values.mask,
                values.null_count)
    L.check(_lib.gx_square(values.gx, values.data_ptr, n, sq.data_ptr, stream_ptr()), "gx_square")
    k2, ss, _, _ = groupby_sum_count(keys, sq)
    o1, o2 = sorted_order(k), sorted_order(k2)  # the two hash passes emit the groups in their own orders
    k, s, cv, ss = gather(k, o1), gather(s, o1), gather(cv, o1), gather(ss, o2)
    g = k.size
    outs = []
    for mode in (1, 2, 0):
        o = Column.empty(np.float64, g, nullable=True)
        cnt = _dev_i64()
        L.check(_lib.gx_var_from_sums(s.gx, ss.data_ptr, s.data_ptr, cv.data_ptr, g, int(ddof), mode, o.data_ptr,
                                      o.mask_ptr, ptr(cnt), stream_ptr()), "gx_var_from_sums")
        o.null_count = int(cnt.item())
        outs.append(o)
    return k, outs[0], outs[1], outs[2], cv


def groupby_argmin_argmax(keys: Column, values: Column):
    """groupby ARGMIN / ARGMAX: (keys, argmin, argmax, count_valid); the row index (INT32) of each group's
    MIN / MAX value -- the smallest such row on ties; meaningful where count_valid > 0."""
    k, mn, mx, cv = groupby_min_max(keys, values)
    g = k.size
    amin, amax = Column.empty(np.int32, g), Column.empty(np.int32, g)
    if g == 0:
        return k, amin, amax, cv
    gid = HashJoin(k).lookup(keys)  # row -> position of its key in k (negative for null keys)
    valid = values.mask_ptr if values.has_nulls() else None
    for target, out in ((mn, amin), (mx, amax)):
        L.check(_lib.gx_groupby_arg_select(values.gx, values.data_ptr, valid, gid.data_ptr, keys.size, target.data_ptr, g,
                                           out.data_ptr, stream_ptr()), "gx_groupby_arg_select")
    return k, amin, amax, cv


def groupby_scan(sorted_keys: Column, values: Column, op: str = "sum") -> Column:
    """Segmented inclusive scan over already-sorted keys (groupby::scan's value pass)."""
    opc = {"sum": L.OP_SUM, "min": L.OP_MIN, "max": L.OP_MAX}[op]
    out_dt = (np.int64 if values.dtype.kind in "iub" else values.dtype) if op == "sum" else values.dtype
    out = Column.empty(out_dt, values.size)
    _run(_lib.gx_segmented_scan, sorted_keys.gx, sorted_keys.data_ptr, values.gx, values.data_ptr,
         values.mask_ptr if values.has_nulls() else None, values.size, opc, out.data_ptr)
    return out


# ------------------------------------------------------------------------------------------------
# reduce / scan  (cudf::reduce / cudf::scan: reduction.hpp:60-65,229-235)
# ------------------------------------------------------------------------------------------------
_OPS = {"sum": L.OP_SUM, "product": L.OP_PRODUCT, "min": L.OP_MIN, "max": L.OP_MAX}


def reduce(col: Column, op: str, out_dtype=None):
    """cudf::reduce -> (python scalar, is_valid).  Invalid iff no valid element
    (reductions.cpp; simple.cuh:47-85)."""
    if out_dtype is None:
        if op in ("sum", "product"):
            out_dtype = np.float64 if col.dtype.kind == "f" else (np.uint64 if col.dtype.kind == "u" else np.int64)
        else:
            out_dtype = col.dtype
    out_dtype = np.dtype(out_dtype)
    out = Column.empty(out_dtype, 1)
    cnt = _dev_i64()
    _run(_lib.gx_reduce, col.gx, col.data_ptr, col.mask_ptr if col.has_nulls() else None, col.size, _OPS[op],
         gx_dtype(out_dtype), out.data_ptr, ptr(cnt))
    valid = int(cnt.item()) > 0
    return out.to_numpy()[0], valid


def scan(col: Column, op: str = "sum", inclusive: bool = True, null_include: bool = False) -> Column:
    """cudf::scan; output dtype == input dtype; mask handling of scan_inclusive.cu:198-216."""
    out = Column.empty(col.dtype, col.size)
    valid = col.mask_ptr if col.has_nulls() else None
    _run(_lib.gx_scan, col.gx, col.data_ptr, valid, col.size, _OPS[op], int(inclusive), out.data_ptr)
    if col.mask is not None:  # nullable input -> nullable output, even when no bit happens to be 0
        if null_include:  # everything from the first null on is null (mask_scan :36-61)
            pos = _dev_i64()
            L.check(_lib.gx_bitmask_first_unset(col.mask_ptr, col.size, ptr(pos), stream_ptr()), "first_unset")
            first = min(col.size, int(pos.item()) + (0 if inclusive else 1)) if col.has_nulls() else col.size
            out.mask = torch.zeros(bitmask_words(col.size), dtype=torch.int32, device="cuda")
            L.check(_lib.gx_bitmask_set(ptr(out.mask), 0, first, 1, stream_ptr()), "bitmask_set")
            out.null_count = col.size - first
        else:
            out.mask = col.mask.clone()
            out.null_count = col.null_count
    return out


# ------------------------------------------------------------------------------------------------
# synthetic data / checks (bench + tests)
# ------------------------------------------------------------------------------------------------

def random_column(dtype, n: int, seed: int, lo: int = 0, hi: int = 0) -> Column:
    out = Column.empty(dtype, n)
    L.check(_lib.gx_fill_random(out.gx, out.data_ptr, n, seed, lo, hi, stream_ptr()), "gx_fill_random")
    return out


def checksum(col: Column, descending: bool = False):
    """(sum, xor, sortedness violations) computed on the device."""
    res = torch.zeros(3, dtype=torch.int64, device="cuda")
    L.check(_lib.gx_checksum(col.gx, col.data_ptr, col.size, int(descending), ptr(res), stream_ptr()), "gx_checksum")
    r = res.cpu().numpy().view(np.uint64)
    return int(r[0]), int(r[1]), int(r[2])
