"""ctypes loader for oracle/liboracle.so (plain-C restatement, TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        p = ctypes.c_void_p
        i64 = ctypes.c_int64
        L.orc_murmur3_u32_array.argtypes = [p, i64, ctypes.c_uint32, p]
        L.orc_murmur3_u64_array.argtypes = [p, i64, ctypes.c_uint32, p]
        L.orc_sort_i64.argtypes = [p, p, p, i64, ctypes.c_int]
        L.orc_sorted_order_i64.argtypes = [p, p, i64, ctypes.c_int]
        L.orc_sort_32.argtypes = [p, p, p, i64, ctypes.c_int, ctypes.c_int]
        L.orc_inner_join_i64.argtypes = [p, i64, p, i64, p, p, i64]
        L.orc_inner_join_i64.restype = i64
        L.orc_groupby_dense_sum_count.argtypes = [p, p, i64, ctypes.c_int32, p, p]
        L.orc_inclusive_sum_i64.argtypes = [p, p, i64]
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def murmur3(values, seed=0):
    v = np.ascontiguousarray(values)
    out = np.empty(len(v), np.uint32)
    if v.dtype.itemsize == 4:
        lib().orc_murmur3_u32_array(_ptr(v), len(v), seed, _ptr(out))
    else:
        lib().orc_murmur3_u64_array(_ptr(v), len(v), seed, _ptr(out))
    return out


def sort_i64(values, descending=False):
    v = np.ascontiguousarray(values, dtype=np.int64)
    out = np.empty_like(v)
    tmp = np.empty_like(v)
    lib().orc_sort_i64(_ptr(v), _ptr(out), _ptr(tmp), len(v), int(descending))
    return out


def sort_32(values, descending=False):
    """int32 / uint32 keys (the dtype decides the sign flip)"""
    v = np.ascontiguousarray(values)
    assert v.dtype in (np.int32, np.uint32)
    out = np.empty_like(v)
    tmp = np.empty_like(v)
    lib().orc_sort_32(_ptr(v), _ptr(out), _ptr(tmp), len(v), int(v.dtype == np.int32), int(descending))
    return out


def sorted_order_i64(values, descending=False):
    v = np.ascontiguousarray(values, dtype=np.int64)
    out = np.empty(len(v), np.int32)
    lib().orc_sorted_order_i64(_ptr(v), _ptr(out), len(v), int(descending))
    return out


def inner_join_i64(left, right):
    l = np.ascontiguousarray(left, dtype=np.int64)
    r = np.ascontiguousarray(right, dtype=np.int64)
    n = lib().orc_inner_join_i64(_ptr(l), len(l), _ptr(r), len(r), None, None, 0)
    ol = np.empty(n, np.int32)
    orr = np.empty(n, np.int32)
    lib().orc_inner_join_i64(_ptr(l), len(l), _ptr(r), len(r), _ptr(ol), _ptr(orr), n)
    return ol, orr


def groupby_dense_sum_count(keys, vals, ngroups):
    k = np.ascontiguousarray(keys, dtype=np.int32)
    v = np.ascontiguousarray(vals, dtype=np.float64)
    s = np.empty(ngroups, np.float64)
    c = np.empty(ngroups, np.int32)
    lib().orc_groupby_dense_sum_count(_ptr(k), _ptr(v), len(k), ngroups, _ptr(s), _ptr(c))
    return s, c


def inclusive_sum_i64(values):
    v = np.ascontiguousarray(values, dtype=np.int64)
    out = np.empty_like(v)
    lib().orc_inclusive_sum_i64(_ptr(v), _ptr(out), len(v))
    return out
