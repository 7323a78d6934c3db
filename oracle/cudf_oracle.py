"""CPU oracle for the cudf hot path (sort / hash-join / groupby / scan / reduce / hash).

TEST INFRASTRUCTURE ONLY.  Nothing under ``cudf_amd/`` may import this module: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and there
only as the checker.  The product path is the HIP library behind ``include/cudf_amd/gx.h``.

Every function is a NumPy restatement of the *observable* semantics of the reference
(rapidsai/cudf @ 26.10) for the hot path, citing the reference file:line it follows.  The
digit passes / hash-table loops of the reference live in third-party CCCL (cub) and
cuCollections (cuco), which are NOT vendored under /root/reference (pinned only through
rapids-cmake's versions.json for release 26.10); their *results* are pinned here through

  * the reference's own golden vectors (tests/golden/reference_vectors.py, transcribed from
    cpp/tests/{sort,join,groupby,reductions}/*), checked in tests/test_oracle_golden.py, and
  * the published MurmurHash3_x86_32 known-answer vectors (SMHasher), same test file.

The reference itself (libcudf) cannot be built or imported in this environment (needs nvcc,
CCCL, cuco, rmm and an NVIDIA GPU), so there is no ``oracle/_ref``.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np

JOIN_NO_MATCH = np.int32(-(2**31))  # cpp/include/cudf/join/join.hpp:72  (JoinNoMatch = INT32_MIN)

# ----------------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------------


def _is_float(a: np.ndarray) -> bool:
    return a.dtype.kind == "f"


def sortable_bits(values: np.ndarray) -> np.ndarray:
    """Map fixed-width values to unsigned integers whose unsigned order is the cudf order.

    ints: two's complement sign flip (what cub's radix twiddle does; call site
    cpp/src/sort/sort_radix.cu:69-76).  floats: IEEE total-order flip with -0.0 == +0.0 and every
    NaN mapped to the maximum (NaN sorts after +Inf, all NaNs equivalent:
    cpp/include/cudf/detail/row_operator/common_utils.cuh:157-169; radix NaN rule
    cpp/src/sort/sort_radix.cu:36-45).
    """
    v = np.ascontiguousarray(values)
    nbits = v.dtype.itemsize * 8
    u = np.dtype(f"u{v.dtype.itemsize}")
    if v.dtype.kind == "b":
        return v.astype(np.uint8)
    if v.dtype.kind == "u":
        return v.copy()
    if v.dtype.kind == "i":
        return v.view(u) ^ u.type(1 << (nbits - 1))
    if v.dtype.kind == "f":
        bits = v.view(u).copy()
        sign = u.type(1 << (nbits - 1))
        bits[bits == sign] = 0  # -0.0 -> +0.0
        neg = (bits & sign) != 0
        out = np.where(neg, ~bits, bits | sign)
        out[np.isnan(v)] = u.type(~u.type(0))
        return out.astype(u)
    raise TypeError(f"unsupported dtype {v.dtype}")


def _stable_argsort_unsigned(keys: np.ndarray) -> np.ndarray:
    return np.argsort(keys, kind="stable")


# ----------------------------------------------------------------------------------------------
# sort  
# ----------------------------------------------------------------------------------------------


def sorted_order(
    values: np.ndarray,
    valid: Optional[np.ndarray] = None,
    ascending: bool = True,
    null_before: bool = True,
) -> np.ndarray:
    """cudf::sorted_order / stable_sorted_order of ONE column -> int32 row indices.

    No nulls: the radix path (cpp/src/sort/sorted_order_radix.cu:63-96,124-137): stable
    LSD sort of (key, iota); for floats the key is the pair (isnan*(idx+1), f)
    (:37-48), so ascending puts NaNs last in index order and DESCENDING puts them first in
    *reverse* index order (cub SortPairsDescending on the composite key).
    With nulls: the comparator path (cpp/src/sort/sort_column_impl.cuh:35-57): nulls are all
    equivalent, placed by null_order and flipped when descending (:42-45); NaNs equivalent and
    greater than every number; our implementation is always stable (the unstable entry point
    permits any tie order, stable_sorted_order requires this one).
    """
    v = np.asarray(values)
    n = v.shape[0]
    idx = np.arange(n, dtype=np.int64)
    if valid is not None and not bool(np.all(valid)):
        valid = np.asarray(valid, dtype=bool)
        nulls = idx[~valid]
        good = idx[valid]
        bits = sortable_bits(v[valid])
        if not ascending:
            bits = ~bits
        order_valid = good[_stable_argsort_unsigned(bits)]
        nulls_first = null_before if ascending else (not null_before)
        out = np.concatenate([nulls, order_valid] if nulls_first else [order_valid, nulls])
        return out.astype(np.int32)
    bits = sortable_bits(v)
    if ascending:
        return _stable_argsort_unsigned(bits).astype(np.int32)
    order = _stable_argsort_unsigned(~bits)
    if _is_float(v):
        nan_count = int(np.isnan(v).sum())
        if nan_count > 1:  # composite key (idx+1, f) descending => NaN block in reverse index order
            order[:nan_count] = order[:nan_count][::-1]
    return order.astype(np.int32)


def sort_keys(values: np.ndarray, ascending: bool = True) -> np.ndarray:
    """cudf::sort of a single non-null fixed-width column (cpp/src/sort/sort.cu:52-67 ->
    sort_radix, cpp/src/sort/sort_radix.cu:151-161).  Bit-exact output incl. -0.0/NaN payloads."""
    return np.asarray(values)[sorted_order(values, None, ascending)]


def sorted_order_table(cols: Sequence[np.ndarray], ascending=True) -> np.ndarray:
    """cudf::stable_sorted_order of a TABLE of numeric columns without nulls -> int32 row indices: the row indices sorted (stably) under
    the lexicographic row comparator (cpp/src/sort/sort_impl.cuh:61-93: thrust::stable_sort of the sequence with
    row::lexicographic::self_comparator).  Per column the element order of the comparator path: NaN equivalent to each other and greater
    than every number, -0.0 == +0.0 (cpp/include/cudf/detail/row_operator/common_utils.cuh:157-169), reversed as a whole for a
    DESCENDING column (so NaN first) -- NOT the single-column radix path's reverse-row-order rule for NaN.  cudf::sorted_order (the
    unstable entry point) may order tied rows any way; the product is stable, which satisfies both."""
    cols = [np.asarray(c) for c in cols]
    asc = [ascending] * len(cols) if isinstance(ascending, bool) else list(ascending)
    assert len(asc) == len(cols) and len(cols) >= 1
    keys = []
    for c, a in zip(cols, asc):
        b = sortable_bits(c)
        keys.append(b if a else ~b)
    # np.lexsort: the LAST key is the primary one; it is stable (equal tuples keep their input order)
    return np.lexsort(tuple(reversed(keys))).astype(np.int32)


def sort_by_key(values: Sequence[np.ndarray], keys: np.ndarray, key_valid=None, ascending=True,
                null_before=True):
    """cudf::sort_by_key = gather(values, sorted_order(keys)) (cpp/src/sort/sort.cu:31-50)."""
    order = sorted_order(keys, key_valid, ascending, null_before)
    return [np.asarray(c)[order] for c in values]


def gather(values: np.ndarray, gather_map: np.ndarray) -> np.ndarray:
    """out[i] = in[map[i]] (cpp/include/cudf/detail/gather.cuh:108-131)."""
    return np.asarray(values)[np.asarray(gather_map, dtype=np.int64)]


# ----------------------------------------------------------------------------------------------
# hashing
# ----------------------------------------------------------------------------------------------

_M32 = 0xFFFFFFFF


def _rotl32(x: np.ndarray, r: int) -> np.ndarray:
    x = x & _M32
    return ((x << r) | (x >> (32 - r))) & _M32


def murmur3_32_bytes(data: bytes, seed: int = 0) -> int:
    """Published MurmurHash3_x86_32 (Appleby, SMHasher) for arbitrary bytes: scalar KAT helper."""
    c1, c2 = 0xCC9E2D51, 0x1B873593
    h = seed & _M32
    nblocks = len(data) // 4
    for i in range(nblocks):
        k = int.from_bytes(data[4 * i : 4 * i + 4], "little")
        k = (k * c1) & _M32
        k = ((k << 15) | (k >> 17)) & _M32
        k = (k * c2) & _M32
        h ^= k
        h = ((h << 13) | (h >> 19)) & _M32
        h = (h * 5 + 0xE6546B64) & _M32
    tail = data[4 * nblocks :]
    k = 0
    if len(tail) >= 3:
        k ^= tail[2] << 16
    if len(tail) >= 2:
        k ^= tail[1] << 8
    if len(tail) >= 1:
        k ^= tail[0]
        k = (k * c1) & _M32
        k = ((k << 15) | (k >> 17)) & _M32
        k = (k * c2) & _M32
        h ^= k
    h ^= len(data)
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & _M32
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & _M32
    h ^= h >> 16
    return h


def murmur3_32(values: np.ndarray, seed: int = 0, valid: Optional[np.ndarray] = None) -> np.ndarray:
    """Vectorised cudf::hashing::detail::MurmurHash3_x86_32<T> over a fixed-width column.

    cpp/include/cudf/hashing/detail/murmurhash3_x86_32.cuh:22-67: hash of the element's bytes
    (4 or 8 byte types here; bool hashes as one uint8 byte), floats first normalised (NaN ->
    canonical quiet NaN, -0.0 -> +0.0: cpp/include/cudf/hashing/detail/hash_functions.cuh
    normalize_nans_and_zeros); null elements hash to UINT32_MAX
    (cpp/include/cudf/detail/row_operator/primitive_row_operators.cuh:207-268).
    """
    v = np.ascontiguousarray(values)
    if v.dtype.kind == "f":
        v = v.copy()
        v[np.isnan(v)] = np.nan  # canonical quiet NaN
        v[v == 0] = 0.0
    if v.dtype.kind == "b":
        v = v.astype(np.uint8)
    size = v.dtype.itemsize
    raw = v.view(np.uint8).reshape(-1, size)
    c1, c2 = np.uint64(0xCC9E2D51), np.uint64(0x1B873593)
    h = np.full(v.shape[0], seed & _M32, dtype=np.uint64)
    nblocks = size // 4
    if nblocks:
        words = np.ascontiguousarray(raw[:, : 4 * nblocks]).view("<u4").reshape(-1, nblocks)
        for b in range(nblocks):
            k = words[:, b].astype(np.uint64)
            k = (k * c1) & _M32
            k = _rotl32(k, 15)
            k = (k * c2) & _M32
            h ^= k
            h = _rotl32(h, 13)
            h = (h * np.uint64(5) + np.uint64(0xE6546B64)) & _M32
    tail = size - 4 * nblocks
    if tail:
        k = np.zeros(v.shape[0], dtype=np.uint64)
        for t in range(tail - 1, -1, -1):
            k ^= raw[:, 4 * nblocks + t].astype(np.uint64) << np.uint64(8 * t)
        k = (k * c1) & _M32
        k = _rotl32(k, 15)
        k = (k * c2) & _M32
        h ^= k
    h ^= np.uint64(size)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    out = h.astype(np.uint32)
    if valid is not None:
        out = np.where(np.asarray(valid, dtype=bool), out, np.uint32(0xFFFFFFFF))
    return out


def hash_combine(lhs: np.ndarray, rhs: np.ndarray) -> np.ndarray:
    """cpp/include/cudf/hashing/detail/hashing.hpp:83-86 (boost-style 32-bit combine)."""
    l = lhs.astype(np.uint64)
    r = rhs.astype(np.uint64)
    return ((l ^ ((r + np.uint64(0x9E3779B9) + (l << np.uint64(6)) + (l >> np.uint64(2))) & _M32))
            & _M32).astype(np.uint32)


def row_hash(columns: Sequence[np.ndarray], valids: Optional[Sequence] = None, seed: int = 0) -> np.ndarray:
    """Row hash used by join/groupby/partition for primitive keys
    (cpp/include/cudf/detail/row_operator/primitive_row_operators.cuh:247-268): the first
    column's element hash seeds the fold; later columns are folded with hash_combine."""
    valids = valids or [None] * len(columns)
    h = murmur3_32(columns[0], seed, valids[0])
    for c, m in zip(columns[1:], valids[1:]):
        h = hash_combine(h, murmur3_32(c, seed, m))
    return h


def identity_hash32(values: np.ndarray, valid=None) -> np.ndarray:
    """IdentityHash<T> (cpp/src/partitioning/partitioning.cu:852-872): ``static_cast<uint32_t>(key)`` for arithmetic T.  Integers
    wrap to their low 32 bits (sign-extended first when signed); bool -> 0 / 1; floating point truncates toward zero, and where C++
    leaves the cast undefined the device conversion (cvt.rzi.u32 / v_cvt_u32) decides: NaN and values <= -1 -> 0, >= 2**32 ->
    UINT32_MAX.  A null element hashes to UINT32_MAX like in every row hasher (row_operator/hashing.cuh:52-66)."""
    v = np.asarray(values)
    if v.dtype.kind == "b":
        out = v.astype(np.uint32)
    elif v.dtype.kind in "iu":
        out = (v.astype(np.int64 if v.dtype.kind == "i" else np.uint64).view(np.uint64) & _M32).astype(np.uint32)
    elif v.dtype.kind == "f":
        d = v.astype(np.float64)                                  # (exact for float32)
        out = np.zeros(len(d), np.uint32)
        big = d >= 4294967296.0
        ok = (d > -1.0) & ~big                                     # (NaN compares false: stays 0)
        out[ok] = np.trunc(d[ok]).astype(np.uint64).astype(np.uint32)
        out[big] = np.uint32(0xFFFFFFFF)
    else:
        raise TypeError("IdentityHash does not support this data type")
    if valid is not None:
        out = np.where(np.asarray(valid, dtype=bool), out, np.uint32(0xFFFFFFFF))
    return out


def row_hash_identity(columns: Sequence[np.ndarray], valids: Optional[Sequence] = None) -> np.ndarray:
    """The row hasher (row_operator/hashing.cuh:118-134) over IdentityHash: first column's element hash, the others folded with
    hash_combine; the seed is ignored (``IdentityHash(uint32_t) {}``)."""
    valids = valids or [None] * len(columns)
    h = identity_hash32(columns[0], valids[0])
    for c, m in zip(columns[1:], valids[1:]):
        h = hash_combine(h, identity_hash32(c, m))
    return h


def partition_by_map(partition_map: np.ndarray, num_partitions: int) -> Tuple[np.ndarray, np.ndarray]:
    """cudf::partition (cpp/src/partitioning/partitioning.cu:755-842): offsets = exclusive scan of the histogram of the map over
    [0, num_partitions), num_partitions + 1 entries; the reference leaves the order of rows INSIDE a partition to its atomics
    (partition_test.cpp:80-108 compares partitions as sets) -- this restatement keeps them in row order, the order the HIP path
    produces.  Returns (gather order, offsets)."""
    m = np.asarray(partition_map)
    if num_partitions <= 0 or len(m) == 0:
        return np.zeros(0, np.int32), np.zeros(max(num_partitions, 0) + 1, np.int32)
    pid = m.astype(np.int64)
    order = np.argsort(pid, kind="stable").astype(np.int32)
    offsets = np.concatenate([[0], np.cumsum(np.bincount(pid, minlength=num_partitions))]).astype(np.int32)
    return order, offsets


def hash_partition(key_columns: Sequence[np.ndarray], num_partitions: int, seed: int = 0,
                   valids=None, hash_function: str = "murmur3") -> Tuple[np.ndarray, np.ndarray]:
    """cudf::hash_partition (cpp/src/partitioning/partitioning.cu:53-92,568-660): partition id =
    row_hash % P (``& (P-1)`` when P is a power of two -- same value); rows keep their relative
    order inside a partition.  Returns (gather order, offsets[P+1])."""
    nrows = len(key_columns[0]) if len(key_columns) else 0
    if num_partitions <= 0 or nrows == 0 or len(key_columns) == 0:
        # an EMPTY result and num_partitions + 1 zeros (partitioning.cu:883-886; hash_partition_test.cpp:73-141)
        return np.zeros(0, np.int32), np.zeros(max(num_partitions, 0) + 1, np.int32)
    h = row_hash_identity(key_columns, valids) if hash_function == "identity" else row_hash(key_columns, valids, seed)
    pid = (h % np.uint32(num_partitions)).astype(np.int64)
    order = np.argsort(pid, kind="stable").astype(np.int32)
    counts = np.bincount(pid, minlength=num_partitions)
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return order, offsets


# ----------------------------------------------------------------------------------------------
# hash join
# ----------------------------------------------------------------------------------------------


def _join_key(v: np.ndarray) -> np.ndarray:
    """Equality classes of the join/groupby row comparator: NaN == NaN, -0.0 == +0.0
    (cpp/include/cudf/detail/row_operator/common_utils.cuh:215-220)."""
    v = np.asarray(v)
    if v.dtype.kind == "f":
        v = v.copy()
        v[np.isnan(v)] = np.nan
        v[v == 0] = 0.0
        return v.view(np.dtype(f"u{v.dtype.itemsize}"))
    return v


def _factorize_rows(lcols, rcols):
    """Dense ids for rows of two tables so that equal rows (across tables) share an id."""
    nl = len(lcols[0])
    combo = None
    for lc, rc in zip(lcols, rcols):
        both = np.concatenate([_join_key(lc), _join_key(rc)])
        _, inv = np.unique(both, return_inverse=True)
        combo = inv if combo is None else np.unique(
            np.stack([combo, inv], axis=1), axis=0, return_inverse=True)[1].reshape(-1)
    return combo[:nl], combo[nl:]


def inner_join(left_cols, right_cols, left_valids=None, right_valids=None,
               nulls_equal: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """cudf::inner_join (cpp/include/cudf/join/join.hpp:127-166; cpp/src/join/join.cu:27-60):
    every (left_idx, right_idx) whose key rows compare equal; a row with a null in any key
    column matches only rows null in the same columns and only when nulls_equal
    (cpp/src/join/hash_join/hash_join.cu:77-84).  The reference's pair order is unspecified
    (join.hpp:131-134), so the oracle returns pairs sorted by (left, right) -- the canonical form
    the reference's own tests compare in (cpp/tests/join/join_tests.cpp:1186-1210)."""
    if not isinstance(left_cols, (list, tuple)):
        left_cols, right_cols = [left_cols], [right_cols]
    left_cols = [np.asarray(c) for c in left_cols]
    right_cols = [np.asarray(c) for c in right_cols]
    nl, nr = len(left_cols[0]), len(right_cols[0])
    if nl == 0 or nr == 0:
        return np.empty(0, np.int32), np.empty(0, np.int32)
    lv = [np.ones(nl, bool) if m is None else np.asarray(m, bool)
          for m in (left_valids or [None] * len(left_cols))]
    rv = [np.ones(nr, bool) if m is None else np.asarray(m, bool)
          for m in (right_valids or [None] * len(right_cols))]
    # null elements: canonical value 0 plus the validity bit as an extra key column
    lcols = [np.where(m, c, c.dtype.type(0)) for c, m in zip(left_cols, lv)] + [m.astype(np.uint8) for m in lv]
    rcols = [np.where(m, c, c.dtype.type(0)) for c, m in zip(right_cols, rv)] + [m.astype(np.uint8) for m in rv]
    lid, rid = _factorize_rows(lcols, rcols)
    l_ok = np.ones(nl, bool)
    r_ok = np.ones(nr, bool)
    if not nulls_equal:
        for m in lv:
            l_ok &= m
        for m in rv:
            r_ok &= m
    lidx = np.nonzero(l_ok)[0]
    ridx = np.nonzero(r_ok)[0]
    lid, rid = lid[lidx], rid[ridx]
    r_order = np.argsort(rid, kind="stable")
    rid_sorted = rid[r_order]
    lo = np.searchsorted(rid_sorted, lid, "left")
    hi = np.searchsorted(rid_sorted, lid, "right")
    cnt = hi - lo
    total = int(cnt.sum())
    out_l = np.repeat(lidx, cnt)
    starts = np.repeat(lo, cnt)
    within = np.arange(total) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    out_r = ridx[r_order[starts + within]]
    order = np.lexsort((out_r, out_l))
    return out_l[order].astype(np.int32), out_r[order].astype(np.int32)


def canonical_pairs(left_idx, right_idx) -> Tuple[np.ndarray, np.ndarray]:
    """Sort a join result into the canonical (left, right) order used for comparison."""
    l = np.asarray(left_idx).astype(np.int64)
    r = np.asarray(right_idx).astype(np.int64)
    order = np.lexsort((r, l))
    return l[order].astype(np.int32), r[order].astype(np.int32)


def left_join(left_cols, right_cols, left_valids=None, right_valids=None, nulls_equal=True):
    """cudf::left_join: inner pairs + (l, JoinNoMatch) for unmatched left rows
    (cpp/src/join/join.cu:62-85, cpp/src/join/join_utils.cu:45-60)."""
    l, r = inner_join(left_cols, right_cols, left_valids, right_valids, nulls_equal)
    lc = left_cols if isinstance(left_cols, (list, tuple)) else [left_cols]
    nl = len(lc[0])
    matched = np.zeros(nl, bool)
    matched[l] = True
    un = np.nonzero(~matched)[0].astype(np.int32)
    return canonical_pairs(np.concatenate([l, un]),
                           np.concatenate([r, np.full(len(un), JOIN_NO_MATCH, np.int32)]))


def full_join(left_cols, right_cols, left_valids=None, right_valids=None, nulls_equal=True):
    """cudf::full_join: left join + (JoinNoMatch, r) for unmatched right rows
    (cpp/src/join/join_utils.cu:86-157)."""
    l, r = left_join(left_cols, right_cols, left_valids, right_valids, nulls_equal)
    rc = right_cols if isinstance(right_cols, (list, tuple)) else [right_cols]
    nr = len(rc[0])
    matched = np.zeros(nr, bool)
    matched[r[r != JOIN_NO_MATCH]] = True
    un = np.nonzero(~matched)[0].astype(np.int32)
    return canonical_pairs(np.concatenate([l, np.full(len(un), JOIN_NO_MATCH, np.int32)]),
                           np.concatenate([r, un]))


def semi_join(left_cols, right_cols, left_valids=None, right_valids=None, nulls_equal=True) -> np.ndarray:
    """cudf::filtered_join::semi_join: ascending indices of the left rows that have at least one
    match in right (contains map + stable copy_if: cpp/src/join/filtered_join/filtered_join.cu:124-156;
    include/cudf/join/filtered_join.hpp:96-116).  Same row equality as inner_join, without
    materialising the pairs (null x null would be a cross product)."""
    if not isinstance(left_cols, (list, tuple)):
        left_cols, right_cols = [left_cols], [right_cols]
    left_cols = [np.asarray(c) for c in left_cols]
    right_cols = [np.asarray(c) for c in right_cols]
    nl, nr = len(left_cols[0]), len(right_cols[0])
    if nl == 0 or nr == 0:
        return np.empty(0, np.int32)
    lv = [np.ones(nl, bool) if m is None else np.asarray(m, bool) for m in (left_valids or [None] * len(left_cols))]
    rv = [np.ones(nr, bool) if m is None else np.asarray(m, bool) for m in (right_valids or [None] * len(right_cols))]
    lcols = [np.where(m, c, c.dtype.type(0)) for c, m in zip(left_cols, lv)] + [m.astype(np.uint8) for m in lv]
    rcols = [np.where(m, c, c.dtype.type(0)) for c, m in zip(right_cols, rv)] + [m.astype(np.uint8) for m in rv]
    lid, rid = _factorize_rows(lcols, rcols)
    l_ok = np.ones(nl, bool)
    r_ok = np.ones(nr, bool)
    if not nulls_equal:
        for m in lv:
            l_ok &= m
        for m in rv:
            r_ok &= m
    hit = l_ok & np.isin(lid, rid[r_ok])
    return np.nonzero(hit)[0].astype(np.int32)


def anti_join(left_cols, right_cols, left_valids=None, right_valids=None, nulls_equal=True) -> np.ndarray:
    """cudf::filtered_join::anti_join: ascending indices of the left rows with no match in right
    (filtered_join.hpp:118-139); with an empty right table every left row."""
    lc = left_cols if isinstance(left_cols, (list, tuple)) else [left_cols]
    nl = len(lc[0])
    keep = np.ones(nl, bool)
    keep[semi_join(left_cols, right_cols, left_valids, right_valids, nulls_equal)] = False
    return np.nonzero(keep)[0].astype(np.int32)


def distinct_left_join(left_cols, right_cols, left_valids=None, right_valids=None, nulls_equal=True) -> np.ndarray:
    """cudf::distinct_hash_join::left_join (include/cudf/join/distinct_hash_join.hpp:96-116): right
    holds distinct keys; result[i] = the right row matching left row i, or JoinNoMatch."""
    lc = left_cols if isinstance(left_cols, (list, tuple)) else [left_cols]
    nl = len(lc[0])
    l, r = inner_join(left_cols, right_cols, left_valids, right_valids, nulls_equal)
    out = np.full(nl, JOIN_NO_MATCH, np.int32)
    out[l] = r
    return out


# ----------------------------------------------------------------------------------------------
# groupby
# ----------------------------------------------------------------------------------------------


def _exact_group_sums(labels: np.ndarray, values: np.ndarray, ngroups: int) -> np.ndarray:
    """Correctly rounded float64 sum per group (math.fsum): the parity target for f64 SUM/MEAN
    (north_star: within 1 ulp).  The reference's own sum is an unordered relaxed atomic add
    (cpp/include/cudf/detail/utilities/device_atomics.cuh:57-62), i.e. any summation order."""
    order = np.argsort(labels, kind="stable")
    sv = values[order].astype(np.float64)
    bounds = np.searchsorted(labels[order], np.arange(ngroups + 1))
    out = np.empty(ngroups, np.float64)
    for g in range(ngroups):
        out[g] = math.fsum(sv[bounds[g] : bounds[g + 1]])
    return out


def groupby_agg(keys: np.ndarray, values: np.ndarray, aggs: Sequence[str],
                keys_valid=None, values_valid=None, exact: bool = True, ddof: int = 1):
    """cudf::groupby::groupby(keys, null_policy::EXCLUDE).aggregate(...) for one key column and
    one values column (cpp/src/groupby/groupby.cu:220-237, hash path cpp/src/groupby/hash/*).

    Returns (unique_keys_sorted, {agg: (result, result_valid)}) with groups in ascending key
    order -- the canonical order the reference's tests compare in
    (cpp/tests/groupby/groupby_test_util.cpp:37-49,80-89).
    Result types (cpp/include/cudf/detail/aggregation/aggregation.hpp:879-1001): SUM of integers
    -> int64 (wraps mod 2^64), SUM of floats -> same float type (f64 here), COUNT_* -> int32,
    MEAN -> float64, MIN/MAX -> input type.  Rows whose key is null are dropped
    (cpp/src/groupby/hash/compute_groupby.cu:62-66); a group with no valid value gives a null
    SUM/MEAN/MIN/MAX and COUNT_VALID 0 (cpp/src/groupby/hash/output_utils.cu:68-70).
    """
    keys = np.asarray(keys)
    values = np.asarray(values)
    n = keys.shape[0]
    kv = np.ones(n, bool) if keys_valid is None else np.asarray(keys_valid, bool)
    vv = np.ones(n, bool) if values_valid is None else np.asarray(values_valid, bool)
    keys, values, vv = keys[kv], values[kv], vv[kv]
    uniq, labels = np.unique(_join_key(keys), return_inverse=True)
    labels = labels.reshape(-1)
    g = len(uniq)
    first = np.full(g, -1, np.int64)
    # representative key value: first occurrence (matters only for -0.0/NaN payloads)
    order = np.argsort(labels, kind="stable")
    if len(order):
        bounds = np.searchsorted(labels[order], np.arange(g))
        first = order[bounds]
    out_keys = keys[first] if g else keys[:0]
    count_valid = np.bincount(labels[vv], minlength=g).astype(np.int32)
    count_all = np.bincount(labels, minlength=g).astype(np.int32)
    res = {}
    has = count_valid > 0
    for a in aggs:
        if a == "count_valid":
            res[a] = (count_valid, np.ones(g, bool))
        elif a == "count_all":
            res[a] = (count_all, np.ones(g, bool))
        elif a in ("sum", "mean"):
            if values.dtype.kind == "f":
                if exact:
                    s = _exact_group_sums(labels[vv], values[vv].astype(np.float64), g)
                else:
                    s = np.bincount(labels[vv], weights=values[vv].astype(np.float64), minlength=g)
                s = s.astype(values.dtype) if a == "sum" else s
            else:
                acc = np.zeros(g, np.uint64)
                np.add.at(acc, labels[vv], values[vv].astype(np.int64).view(np.uint64))
                s = acc.view(np.int64)
            if a == "sum":
                res[a] = (s, has.copy())
            else:
                # MEAN = SUM / COUNT_VALID in double (hash_compound_agg_finalizer.cu:92-133)
                with np.errstate(divide="ignore", invalid="ignore"):
                    if values.dtype.kind == "f":
                        m = s.astype(np.float64) / count_valid
                    else:
                        m = s.astype(np.float64) / count_valid
                res[a] = (np.where(has, m, 0.0), has.copy())
        elif a in ("var", "std", "m2"):
            # hash path of the reference: SUM_OF_SQUARES + SUM + COUNT_VALID, then
            # M2 = sum_sqr - sum*sum/count, VAR = M2/(count - ddof), STD = sqrt(VAR); null when
            # count - ddof <= 0 (cpp/src/groupby/common/m2_var_std.cu:44-61,153-190).  Integer
            # sums (and sums of squares) are int64 and wrap.
            if values.dtype.kind == "f":
                v64 = values[vv].astype(np.float64)
                if exact:
                    sm = _exact_group_sums(labels[vv], v64, g)
                    ss = _exact_group_sums(labels[vv], v64 * v64, g)
                else:
                    sm = np.bincount(labels[vv], weights=v64, minlength=g)
                    ss = np.bincount(labels[vv], weights=v64 * v64, minlength=g)
            else:
                acc = np.zeros(g, np.uint64)
                np.add.at(acc, labels[vv], values[vv].astype(np.int64).view(np.uint64))
                sm = acc.view(np.int64).astype(np.float64)
                acc2 = np.zeros(g, np.uint64)
                x = values[vv].astype(np.int64).view(np.uint64)
                np.add.at(acc2, labels[vv], x * x)
                ss = acc2.view(np.int64).astype(np.float64)
            with np.errstate(divide="ignore", invalid="ignore"):
                cnt = count_valid.astype(np.float64)
                m2 = np.where(has, ss - sm * sm / cnt, 0.0)
                df = count_valid - ddof
                ok = has & (df > 0)
                var = np.where(ok, m2 / np.where(ok, df, 1), 0.0)
            if a == "m2":
                res[a] = (m2, np.ones(g, bool))
            elif a == "var":
                res[a] = (var, ok)
            else:
                res[a] = (np.sqrt(var), ok)
        elif a in ("argmin", "argmax"):
            # row index (in the ORIGINAL table) of the group's MIN / MAX; smallest row on ties
            rows = np.nonzero(kv)[0][vv]
            fill = values[vv]
            lab = labels[vv]
            out = np.zeros(g, np.int32)
            if len(fill):
                sb = sortable_bits(fill)
                if a == "argmax":
                    sb = ~sb
                o2 = np.lexsort((rows, sb, lab))
                b = np.searchsorted(lab[o2], np.arange(g))
                out = np.where(has, rows[o2][np.minimum(b, len(fill) - 1)], 0).astype(np.int32)
            res[a] = (out, has.copy())
        elif a in ("min", "max"):
            fill = values[vv]
            lab = labels[vv]
            out = np.zeros(g, values.dtype)
            if len(fill):
                o2 = np.lexsort((sortable_bits(fill), lab))
                b = np.searchsorted(lab[o2], np.arange(g + 1))
                sel = np.where(has, b[:-1] if a == "min" else np.maximum(b[1:] - 1, 0), 0)
                out = np.where(has, fill[o2][np.minimum(sel, len(fill) - 1)], 0).astype(values.dtype)
            res[a] = (out, has.copy())
        else:
            raise ValueError(a)
    return out_keys, res


def groupby_scan_sum(keys: np.ndarray, values: np.ndarray, keys_valid=None, values_valid=None):
    """groupby::scan with SUM (cpp/src/groupby/groupby.cu:240-259 -> sort path
    cpp/src/groupby/sort/scan.cpp:214-238, sort_helper.cu:73-162, group_scan_util.cuh:77-133):
    keys come out sorted (stable), values are the per-group inclusive prefix sum in that order;
    integer inputs accumulate in int64; null values stay null and are skipped."""
    keys = np.asarray(keys)
    values = np.asarray(values)
    n = len(keys)
    kv = np.ones(n, bool) if keys_valid is None else np.asarray(keys_valid, bool)
    vv = np.ones(n, bool) if values_valid is None else np.asarray(values_valid, bool)
    keep = np.nonzero(kv)[0]
    order = keep[np.argsort(sortable_bits(keys[keep]), kind="stable")]
    sk = keys[order]
    sv = values[order]
    svv = vv[order]
    acc_t = np.float64 if values.dtype.kind == "f" else np.int64
    x = np.where(svv, sv, 0).astype(acc_t)
    out = np.empty(len(x), acc_t)
    if len(x):
        newgrp = np.concatenate([[True], _join_key(sk)[1:] != _join_key(sk)[:-1]])
        starts = np.nonzero(newgrp)[0]
        ends = np.append(starts[1:], len(x))
        for s, e in zip(starts, ends):
            if acc_t is np.int64:
                out[s:e] = np.cumsum(x[s:e].view(np.uint64)).view(np.int64)
            else:
                out[s:e] = np.cumsum(x[s:e])
    return sk, out, svv


# ----------------------------------------------------------------------------------------------
# sort-path groupby, groupby::scan COUNT / MIN / MAX, get_groups / shift / replace_nulls  
# ----------------------------------------------------------------------------------------------


def _group_sort(keys, keys_valid=None, include_null_keys=False, keys_sorted=False):
    """The reference's sort helper (cpp/src/groupby/sort/sort_helper.cu:73-162): the stable order of the key column with
    nulls AFTER (:92-94) -- rows with a null key are dropped under null_policy::EXCLUDE (pushed to the end and cut off,
    :113-135) and form the LAST group under INCLUDE -- plus group labels / offsets of the kept rows in that order
    (:151-214).  Pre-sorted keys (sorted::YES) keep their order.  Returns (order, labels, offsets)."""
    keys = np.asarray(keys)
    n = len(keys)
    kv = np.ones(n, bool) if keys_valid is None else np.asarray(keys_valid, bool)
    if keys_sorted and (include_null_keys or bool(kv.all())):
        order = np.arange(n, dtype=np.int64)
    else:
        order = sorted_order(keys, kv, True, False).astype(np.int64)  # nulls AFTER
        if not include_null_keys:
            order = order[: int(kv.sum())]
    m = len(order)
    if m == 0:
        return order, np.zeros(0, np.int32), np.zeros(1, np.int32)
    jk = _join_key(keys[order])
    okv = kv[order]
    same = (jk[1:] == jk[:-1]) & okv[1:] & okv[:-1]
    same |= ~okv[1:] & ~okv[:-1]  # null == null: the null keys are one group
    head = np.concatenate([[True], ~same])
    labels = (np.cumsum(head) - 1).astype(np.int32)
    offsets = np.append(np.nonzero(head)[0], m).astype(np.int32)
    return order, labels, offsets


def groupby_sort_agg(keys, values, agg: str, keys_valid=None, values_valid=None, include_null_keys=False,
                     keys_sorted=False, n_th: int = 0):
    """cudf::groupby::aggregate on the SORT path (cpp/src/groupby/sort/aggregate.cpp:94-142,276-301,879-903;
    group_single_pass_reduction_util.cuh:133-200; group_count.cu:25-89; group_nth_element.cu:30-80): keys come out
    sorted (null key group last), one result per group.  agg in sum / product / min / max / count_valid / count_all /
    nth.  SUM and PRODUCT of integers are int64 and wrap (aggregation.hpp:949-970), floats keep their type; MIN / MAX
    keep the input type; a group without a valid value is null (:185-195).  NTH_ELEMENT (null_policy::INCLUDE): the
    value at group start + n (group end + n for n < 0), null when the group is shorter or the row is null.
    Returns (unique_keys, unique_keys_valid, result, result_valid); float sums are the correctly rounded ones."""
    keys, values = np.asarray(keys), np.asarray(values)
    n = len(keys)
    kv = np.ones(n, bool) if keys_valid is None else np.asarray(keys_valid, bool)
    vv = np.ones(n, bool) if values_valid is None else np.asarray(values_valid, bool)
    order, labels, offsets = _group_sort(keys, kv, include_null_keys, keys_sorted)
    g = len(offsets) - 1
    first = order[offsets[:-1]] if g else order[:0]
    ukeys, ukv = keys[first], kv[first]
    sv, svv = values[order], vv[order]
    cnt = np.bincount(labels[svv], minlength=g).astype(np.int32) if g else np.zeros(0, np.int32)
    has = cnt > 0
    isf = values.dtype.kind == "f"
    if agg == "count_valid":
        return ukeys, ukv, cnt, np.ones(g, bool)
    if agg == "count_all":
        return ukeys, ukv, np.diff(offsets).astype(np.int32), np.ones(g, bool)
    if agg == "nth":
        sizes = np.diff(offsets)
        j = np.where(n_th >= 0, n_th, sizes + n_th)
        ok = (j >= 0) & (j < sizes)
        pos = np.where(ok, offsets[:-1] + j, 0)
        out = sv[pos] if g else sv[:0]
        return ukeys, ukv, out, ok & (svv[pos] if g else ok)
    if agg == "sum":
        if isf:
            out = _exact_group_sums(labels[svv], sv[svv].astype(np.float64), g).astype(values.dtype)
        else:
            acc = np.zeros(g, np.uint64)
            np.add.at(acc, labels[svv], sv[svv].astype(np.int64).view(np.uint64))
            out = acc.view(np.int64)
        return ukeys, ukv, out, has
    if agg == "product":
        if isf:
            out = np.ones(g, np.float64)
            np.multiply.at(out, labels[svv], sv[svv].astype(np.float64))
            out = out.astype(values.dtype)
        else:
            acc = np.ones(g, np.uint64)
            with np.errstate(over="ignore"):
                np.multiply.at(acc, labels[svv], sv[svv].astype(np.int64).view(np.uint64))
            out = acc.view(np.int64)
        return ukeys, ukv, out, has
    if agg in ("min", "max"):
        out = np.zeros(g, values.dtype)
        fill, lab = sv[svv], labels[svv]
        if len(fill):
            o2 = np.lexsort((sortable_bits(fill), lab))
            b = np.searchsorted(lab[o2], np.arange(g + 1))
            sel = np.where(has, b[:-1] if agg == "min" else np.maximum(b[1:] - 1, 0), 0)
            out = np.where(has, fill[o2][np.minimum(sel, len(fill) - 1)], 0).astype(values.dtype)
        return ukeys, ukv, out, has
    raise ValueError(agg)


def groupby_scan(keys, values, op: str, keys_valid=None, values_valid=None, include_null_keys=False, keys_sorted=False):
    """groupby::scan (cpp/src/groupby/sort/scan.cpp:60-238): rows in sorted-key order, per-group INCLUSIVE scan.
    sum / min / max (group_scan_util.cuh:77-133): null values are skipped and stay null, integer SUM in int64;
    count_valid / count_all (group_count_scan.cu:24-62): INT32, never null, count of valid / all rows so far.
    Returns (sorted_keys, out, out_valid)."""
    keys, values = np.asarray(keys), np.asarray(values)
    n = len(keys)
    vv = np.ones(n, bool) if values_valid is None else np.asarray(values_valid, bool)
    order, labels, offsets = _group_sort(keys, keys_valid, include_null_keys, keys_sorted)
    sk, sv, svv = keys[order], values[order], vv[order]
    m = len(order)
    if op in ("count_valid", "count_all"):
        x = svv.astype(np.int64) if op == "count_valid" else np.ones(m, np.int64)
        c = np.cumsum(x)
        base = np.concatenate([[0], c])[offsets[:-1]] if m else c
        out = (c - base[labels]).astype(np.int32) if m else np.zeros(0, np.int32)
        return sk, out, np.ones(m, bool)
    isf = values.dtype.kind == "f"
    import pandas as pd  # segmented cumulative ops at C speed; the semantics are spelled out above
    if op == "sum":
        if isf:  # running sum inside each group, in row order (any association is within the tests' tolerance)
            out = pd.Series(np.where(svv, sv, 0).astype(np.float64)).groupby(labels).cumsum().to_numpy().astype(values.dtype)
        else:  # int64, wrapping: global wrapping prefix sum minus the prefix at the group start
            c = np.cumsum(np.where(svv, sv, 0).astype(np.int64).view(np.uint64))
            base = np.concatenate([np.zeros(1, np.uint64), c])[offsets[:-1]] if m else c
            out = (c - base[labels]).view(np.int64) if m else np.zeros(0, np.int64)
        return sk, out, svv
    if op in ("min", "max"):
        if m == 0:
            return sk, sv.copy(), svv
        if isf:
            ident = np.inf if op == "min" else -np.inf
        else:
            ident = np.iinfo(values.dtype).max if op == "min" else np.iinfo(values.dtype).min
        x = pd.Series(np.where(svv, sv, ident))
        gb = x.groupby(labels)
        out = (gb.cummin() if op == "min" else gb.cummax()).to_numpy().astype(values.dtype)
        return sk, out, svv
    raise ValueError(op)


def groupby_get_groups(keys, values, keys_valid=None, include_null_keys=False):
    """groupby::get_groups (cpp/src/groupby/groupby.cu:261-283): (sorted keys, values gathered in that order, offsets)."""
    order, _, offsets = _group_sort(keys, keys_valid, include_null_keys)
    return np.asarray(keys)[order], np.asarray(values)[order], offsets


def groupby_shift(keys, values, offset: int, fill=None, keys_valid=None, values_valid=None):
    """groupby::shift (cpp/src/groupby/groupby.cu:306-346 -> segmented_shift, cpp/src/copying/segmented_shift.cu):
    in sorted-key order, out[i] = in[i - offset] when that row lies in the same group, else the fill scalar
    (fill None = null).  Returns (sorted_keys, out, out_valid)."""
    keys, values = np.asarray(keys), np.asarray(values)
    n = len(keys)
    vv = np.ones(n, bool) if values_valid is None else np.asarray(values_valid, bool)
    order, labels, offsets = _group_sort(keys, keys_valid)
    sv, svv = values[order], vv[order]
    m = len(order)
    idx = np.arange(m, dtype=np.int64) - offset
    lo, hi = offsets[:-1][labels], offsets[1:][labels]
    inside = (idx >= lo) & (idx < hi)
    src = np.clip(idx, 0, max(m - 1, 0))
    out = np.where(inside, sv[src] if m else sv, values.dtype.type(0 if fill is None else fill))
    ov = np.where(inside, svv[src] if m else svv, fill is not None)
    return keys[order], out.astype(values.dtype), ov


def groupby_replace_nulls(keys, values, values_valid, following: bool, keys_valid=None):
    """groupby::replace_nulls (cpp/src/groupby/groupby.cu:285-321, sort/group_replace_nulls.cu): in sorted-key order a
    null takes the nearest valid value of ITS group before it (PRECEDING) or after it (FOLLOWING), else stays null."""
    keys, values = np.asarray(keys), np.asarray(values)
    vv = np.asarray(values_valid, bool)
    order, labels, offsets = _group_sort(keys, keys_valid)
    sv, svv = values[order].copy(), vv[order].copy()
    for s, e in zip(offsets[:-1], offsets[1:]):
        rng = range(s, e) if not following else range(e - 1, s - 1, -1)
        last = -1
        for i in rng:
            if svv[i]:
                last = i
            elif last >= 0:
                sv[i] = sv[last]
                svv[i] = True
    return keys[order], sv, svv


# ----------------------------------------------------------------------------------------------
# rank / top_k / segmented sort  
# ----------------------------------------------------------------------------------------------

RANK_FIRST, RANK_AVERAGE, RANK_MIN, RANK_MAX, RANK_DENSE = range(5)  # cudf::rank_method (aggregation.hpp:37-43)


def rank(values, valid=None, method: int = RANK_FIRST, ascending=True, null_include=False, null_before=False,
         percentage=False):
    """cudf::rank (cpp/src/sort/rank.cu:59-369).  Rows are ranked by their 1-based position in the (stable) sorted order
    of the column -- nulls take part in that order wherever null_precedence puts them (:290-296); rows that compare equal
    (null == null, NaN == NaN, -0.0 == +0.0: the row equality comparator, :59-97) form a tie group: FIRST = the
    position itself, MIN / MAX = first / last position of the group, AVERAGE = min + (count - 1) / 2 (:236-257),
    DENSE = 1 + number of distinct groups before.  null_policy::EXCLUDE: the output carries the input's null mask
    (:275-284; the masked values are unspecified).  percentage (:343-360): r / count of ranked rows
    (EXCLUDE: non-null rows), DENSE: r / dense rank of sorted position count - 1.  INT32, or FLOAT64 for AVERAGE /
    percentage.  Returns (ranks, out_valid)."""
    v = np.asarray(values)
    n = len(v)
    ok = np.ones(n, bool) if valid is None else np.asarray(valid, bool)
    order = sorted_order(v, ok, ascending, null_before).astype(np.int64)
    sb = sortable_bits(v)[order]
    so = ok[order]
    if n:
        same = ((sb[1:] == sb[:-1]) & so[1:] & so[:-1]) | (~so[1:] & ~so[:-1])
        head = np.concatenate([[True], ~same])
    else:
        head = np.zeros(0, bool)
    dense = np.cumsum(head)  # 1-based dense rank in sorted order
    pos = np.arange(1, n + 1)
    starts = np.nonzero(head)[0]
    ends = np.append(starts[1:], n)
    gid = dense - 1
    if method == RANK_FIRST:
        r = pos.astype(np.float64)
    elif method == RANK_DENSE:
        r = dense.astype(np.float64)
    elif method == RANK_MIN:
        r = (starts[gid] + 1).astype(np.float64)
    elif method == RANK_MAX:
        r = ends[gid].astype(np.float64)
    elif method == RANK_AVERAGE:
        r = (starts[gid] + 1) + ((ends - starts)[gid] - 1) / 2.0
    else:
        raise ValueError(method)
    if percentage:
        count = int(ok.sum()) if not null_include else n
        denom = float(dense[count - 1]) if (method == RANK_DENSE and count > 0) else float(count)
        with np.errstate(divide="ignore", invalid="ignore"):
            r = r / denom
    out = np.empty(n, np.float64)
    out[order] = r
    if not (percentage or method == RANK_AVERAGE):
        out = out.astype(np.int32)
    return out, (np.ones(n, bool) if null_include else ok.copy())


def top_k(values, k: int, descending=True, valid=None):
    """cudf::top_k / top_k_order (cpp/src/sort/top_k.cu:104-150).  k >= size: the column itself / iota (:112,:139-145).
    Otherwise the first k rows of the stable sorted order with nulls never making the top (:121-123); the reference's
    fast path (no nulls, integers: cub::DeviceTopK, :47-74) promises neither order nor tie choice, so the checkable
    contract is the MULTISET of the k values.  Returns (values in this oracle's order, their rows, valid flags)."""
    v = np.asarray(values)
    n = len(v)
    ok = np.ones(n, bool) if valid is None else np.asarray(valid, bool)
    if k == 0 or n == 0:
        return v[:0], np.zeros(0, np.int32), ok[:0]
    if k >= n:
        idx = np.arange(n, dtype=np.int32)
        return v.copy(), idx, ok.copy()
    idx = sorted_order(v, ok, not descending, descending)[:k]  # ascending: nulls AFTER, descending: nulls BEFORE
    return v[idx], idx.astype(np.int32), ok[idx]


def segmented_sorted_order(key_cols, offsets, valids=None, ascending=None, null_before=None):
    """cudf::stable_segmented_sorted_order (cpp/src/sort/segmented_sort_impl.cuh:178-203,265-293): every row gets a
    segment id -- rows of [offsets[j], offsets[j+1]) share one, rows outside every segment each their own, ascending
    with the row -- and the table (segment id, keys...) is sorted lexicographically and stably.  Defaults: all
    ASCENDING, nulls BEFORE (sorting.hpp:33-38)."""
    cols = [np.asarray(c) for c in key_cols]
    n = len(cols[0])
    nc = len(cols)
    valids = [None] * nc if valids is None else valids
    ascending = [True] * nc if ascending is None else ascending
    null_before = [True] * nc if null_before is None else null_before
    offsets = np.asarray(offsets, np.int64)
    ids = np.arange(n, dtype=np.int64)  # rows outside every segment: unique, in place
    for j in range(len(offsets) - 1):
        ids[offsets[j] : offsets[j + 1]] = offsets[j + 1]
    if len(offsets):
        ids[: offsets[0]] = np.arange(offsets[0])
        tail = np.arange(offsets[-1], n)
        ids[offsets[-1] :] = tail + 1 if len(tail) else tail
    order = np.arange(n, dtype=np.int64)
    for c in range(nc - 1, -1, -1):  # LSD over the columns, each pass stable
        ok = np.ones(n, bool) if valids[c] is None else np.asarray(valids[c], bool)
        sub = sorted_order(cols[c][order], ok[order], ascending[c], null_before[c]).astype(np.int64)
        order = order[sub]
    order = order[np.argsort(ids[order], kind="stable")]
    return order.astype(np.int32)


def join_match_counts(left, right, kind: str = "inner", left_valid=None, right_valid=None, nulls_equal=True):
    """hash_join::{inner,left,full}_join_match_context (cpp/include/cudf/join/hash_join.hpp:259-340;
    cpp/src/join/hash_join/size_impl.cuh:26-62): matching right rows per left row; left / full: at least 1 (the row is
    emitted with JoinNoMatch).  A null left key matches the null right keys iff nulls_equal."""
    left, right = np.asarray(left), np.asarray(right)
    lv = np.ones(len(left), bool) if left_valid is None else np.asarray(left_valid, bool)
    rv = np.ones(len(right), bool) if right_valid is None else np.asarray(right_valid, bool)
    rk = np.sort(_join_key(right[rv]))
    lk = _join_key(left)
    c = (np.searchsorted(rk, lk, "right") - np.searchsorted(rk, lk, "left")).astype(np.int64)
    c[~lv] = int((~rv).sum()) if nulls_equal else 0
    if kind in ("left", "full"):
        c = np.maximum(c, 1)
    return c.astype(np.int32)


# ----------------------------------------------------------------------------------------------
# reduce / scan
# ----------------------------------------------------------------------------------------------


def reduce(values: np.ndarray, op: str, valid=None, out_dtype=None, init=None, init_valid=True):
    """cudf::reduce (cpp/src/reductions/reductions.cpp:484-507, simple.cuh:47-85): nulls skipped;
    returns (value, is_valid); is_valid False iff there is no valid element; computed in
    out_dtype (ints wrap).  init (reduction.hpp:124-130): an initial value of the column's type, cast to out_dtype and
    folded in with the operator (simple.cuh:56-77); the result is valid iff the column has a valid row AND the initial
    value is valid (simple.cuh:80-83)."""
    v = np.asarray(values)
    if op in ("count_valid", "count_all"):
        # reductions/count.cpp:37-46: size - null_count (EXCLUDE) or size, cast to the output type; ALWAYS valid -- also for an
        # empty or all-null column (reductions.cpp:252-275: reduce_no_data is reduce); no initial value (reductions.cpp:492-499)
        if init is not None:
            raise ValueError("Initial value is only supported for SUM, SUM_OVERFLOW, PRODUCT, MIN, MAX, ANY, ALL, and HOST_UDF aggregation types")
        od = np.dtype(out_dtype or np.int32)
        if od.kind not in "iuf":
            raise ValueError("COUNT is not supported for boolean or non-numeric types")
        m = np.ones(len(v), bool) if valid is None else np.asarray(valid, bool)
        return od.type(len(v) if op == "count_all" else int(m.sum())), True
    if op in ("any", "all"):
        # reductions/any.cu:79-95, all.cu, simple.cuh:47-85,238-259: max / min over static_cast<bool>(x), nulls skipped, BOOL8 out;
        # no valid row: any = false / all = true and VALID, whatever the initial value (reductions.cpp:163-186); otherwise the
        # initial value is cast to bool and folded in, and an invalid one invalidates the result (simple.cuh:80-83)
        if out_dtype is not None and np.dtype(out_dtype) != np.dtype(bool):
            raise ValueError("any() / all() operation can be applied with output type `bool8` only")
        m = np.ones(len(v), bool) if valid is None else np.asarray(valid, bool)
        x = v[m]
        if len(x) == 0:
            return np.bool_(op == "all"), True
        r = bool((x != 0).any()) if op == "any" else bool((x != 0).all())
        if init is not None:
            if not init_valid:
                return np.bool_(r), False
            iv = bool(np.asarray(init, v.dtype) != 0)
            r = (r or iv) if op == "any" else (r and iv)
        return np.bool_(r), True
    if init is not None:
        if op not in ("sum", "product", "min", "max"):
            raise ValueError("Initial value is only supported for SUM, SUM_OVERFLOW, PRODUCT, MIN, MAX, ANY, ALL, and HOST_UDF aggregation types")
        r, ok = reduce(v, op, valid, out_dtype)
        if not ok or not init_valid:
            return r, False
        od = np.dtype(out_dtype or v.dtype)
        i = np.asarray(init, v.dtype).astype(od)
        with np.errstate(over="ignore"):
            if op == "sum":
                return (np.asarray(r, od) + i).astype(od)[()], True
            if op == "product":
                return (np.asarray(r, od) * i).astype(od)[()], True
        pair = np.array([r, i], od)
        return (pair[np.argmin(sortable_bits(pair))] if op == "min" else pair[np.argmax(sortable_bits(pair))]), True
    m = np.ones(len(v), bool) if valid is None else np.asarray(valid, bool)
    x = v[m]
    explicit_out = out_dtype is not None
    out_dtype = np.dtype(out_dtype or v.dtype)
    if op == "mean" and explicit_out and out_dtype.kind != "f":
        raise ValueError("Unsupported output data type")
    if len(x) == 0:
        return out_dtype.type(0), False
    if op == "product":
        with np.errstate(over="ignore"):
            return x.astype(out_dtype).prod(dtype=out_dtype), True
    if op == "sum":
        if out_dtype.kind == "f":
            return out_dtype.type(math.fsum(x.astype(np.float64))), True
        return x.astype(out_dtype).sum(dtype=out_dtype), True
    if op == "min":
        return x[np.argmin(sortable_bits(x))].astype(out_dtype), True
    if op == "max":
        return x[np.argmax(sortable_bits(x))].astype(out_dtype), True
    if op == "mean":
        # reductions/mean.cu, compound.cuh:41-84, reduction_operators.cuh:256-275: sum / valid count in the floating OUTPUT type
        # ("Unsupported output data type" otherwise); here the correctly rounded quotient of the exact sum
        if not explicit_out:
            out_dtype = np.dtype(np.float64)
        if out_dtype.kind != "f":
            raise ValueError("Unsupported output data type")
        return out_dtype.type(math.fsum(x.astype(np.float64)) / len(x)), True
    raise ValueError(op)


def scan(values: np.ndarray, op: str = "sum", inclusive: bool = True, valid=None,
         null_include: bool = False):
    """cudf::scan (cpp/src/reductions/scan/scan.cpp:13-54, scan_inclusive.cu:36-240,
    scan_exclusive.cu): output dtype == input dtype (ints wrap); null_policy EXCLUDE: nulls are
    replaced by the identity and stay null in the output (mask copied); INCLUDE: everything from
    the first null on is null (mask_scan :36-61).  Returns (values, valid mask)."""
    v = np.asarray(values)
    n = len(v)
    m = np.ones(n, bool) if valid is None else np.asarray(valid, bool)
    dt = v.dtype
    if op == "sum":
        ident = dt.type(0)
        f = np.cumsum
    elif op == "min":
        ident = dt.type(np.inf) if dt.kind == "f" else np.iinfo(dt).max
        f = np.minimum.accumulate
    elif op == "max":
        ident = dt.type(-np.inf) if dt.kind == "f" else np.iinfo(dt).min
        f = np.maximum.accumulate
    elif op == "product":
        ident = dt.type(1)
        f = np.cumprod
    else:
        raise ValueError(op)
    x = np.where(m, v, ident).astype(dt)
    with np.errstate(over="ignore"):
        inc = f(x, dtype=dt) if op in ("sum", "product") else f(x)
    if inclusive:
        out = inc
    else:
        out = np.concatenate([np.array([ident], dt), inc[:-1]]) if n else inc  # typed identity: no float64 promotion
    if null_include:
        first_null = int(np.argmin(m)) if not m.all() else n
        pos = min(n, first_null + (0 if inclusive else 1))
        om = np.arange(n) < pos
    else:
        om = m.copy()
    return out, om


# ----------------------------------------------------------------------------------------------
# ulp distance helper for float parity (tolerance stated in the tests)
# ----------------------------------------------------------------------------------------------


def ulp_diff(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    ia = a.view(np.int64).copy()
    ib = b.view(np.int64).copy()
    ia = np.where(ia < 0, np.int64(-(2**63)) - ia, ia)
    ib = np.where(ib < 0, np.int64(-(2**63)) - ib, ib)
    return np.abs(ia - ib)
