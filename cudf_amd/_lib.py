"""ctypes binding of the C ABI in include/cudf_amd/gx.h.

The product path REQUIRES the HIP library: importing this module raises if
cudf_amd/libcudf_amd.so is missing (build it with ``python scripts/build_ext.py``); there is no CPU
fallback anywhere in the package.
"""
import ctypes
import os

# PyTorch's HIP runtime must be in the process BEFORE libcudf_amd.so is loaded: the kernels are handed
# torch's device pointers and streams, so both have to bind to the same libamdhip64 instance (loading the
# library first pulls in /opt/rocm's copy, torch then brings its own, and launches fail with hipErrorNoDevice).
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcudf_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: the HIP extension is mandatory (no CPU fallback). "
        "Run `python scripts/build_ext.py` (needs hipcc).")

lib = ctypes.CDLL(LIB_PATH)

_p = ctypes.c_void_p
_i64 = ctypes.c_int64
_i = ctypes.c_int
_sz = ctypes.POINTER(ctypes.c_size_t)

# dtype codes (== cudf::type_id)
INT8, INT16, INT32, INT64, UINT8, UINT16, UINT32, UINT64, FLOAT32, FLOAT64, BOOL8 = range(1, 12)
OP_SUM, OP_PRODUCT, OP_MIN, OP_MAX, OP_COUNT_VALID, OP_COUNT_ALL = 0, 1, 2, 3, 4, 5
OP_MEAN = 10

_PROTOS = {
    "gx_version": (ctypes.c_char_p, []),
    "gx_dtype_size": (_i, [_i]),
    "gx_sort_keys": (_i, [_i, _p, _p, _i64, _i, _p, _sz, _p]),
    "gx_sort_pairs": (_i, [_i, _p, _p, _p, _p, _i64, _i, _p, _sz, _p]),
    "gx_sorted_order": (_i, [_i, _p, _p, _i64, _i64, _i, _i, _p, _p, _sz, _p]),
    "gx_sorted_order_table": (_i, [_i, _p, _p, _p, _i64, _p, _p, _sz, _p]),
    "gx_sort_status": (_i, [_p, ctypes.POINTER(_i), _p]),
    "gx_sort_status_async": (_i, [_p, _p, _p]),
    "gx_sort_set_fault_mode": (None, [_i]),
    "gx_sort_set_order_map": (None, [_i]),
    "gx_order_map_applies": (_i, [_i, _i64]),
    "gx_sort_order_map_info": (_i, [_p, _i64, ctypes.POINTER(ctypes.c_int32), _p]),
    "gx_sort_set_algorithm": (None, [_i]),
    "gx_sort_profile": (_i, [_i]),
    "gx_sort_profile_slot": (_i, [_i]),
    "gx_sort_profile_read": (_i, [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(_i)]),
    "gx_sort_profile_read_hybrid": (_i, [ctypes.POINTER(ctypes.c_float)]),
    "gx_sort_set_hybrid": (None, [_i]),
    "gx_sort_set_experiment": (None, [_i]),
    "gx_sort_set_cursor_path": (None, [_i, ctypes.c_float]),
    "gx_sort_set_counting": (None, [_i]),
    "gx_sort_set_splitters": (None, [_i]),
    "gx_sort_set_float_cursor": (None, [_i]),
    "gx_sort_set_spin_limit_ms": (None, [_i]),
    "gx_sort_inject_lost_tile": (None, [ctypes.c_longlong]),
    "gx_sort_split_info": (_i, [_p, ctypes.POINTER(ctypes.c_int32), _p]),
    "gx_sort_cursor_state": (_i, [_p, ctypes.POINTER(ctypes.c_int32), _p]),
    "gx_sort_place_info": (_i, [_p, ctypes.POINTER(ctypes.c_int32), _p]),
    "gx_sort_set_place_grid": (None, [_i]),
    "gx_sort_set_order_words": (None, [_i]),
    "gx_sort_set_cell": (None, [_i]),
    "gx_sort_set_lookback": (None, [_i]),
    "gx_sort_info": (_i, [_p, ctypes.POINTER(ctypes.c_int32), _p]),
    "gx_sort_big_info": (_i, [_p, ctypes.POINTER(ctypes.c_int64), _p]),
    # the sharded sort's halves (called from C++: cudf_amd/cpp/src/distributed.cpp; bound here for the ABI check and ad-hoc use)
    "gx_sortx_sample": (_i, [_i, _p, _i64, _i64, _p, _sz, _p]),
    "gx_sortx_masks": (_i, [_p, ctypes.POINTER(ctypes.c_uint64), _p]),
    "gx_sortx_level0": (_i, [_i, _p, _i64, _i64, ctypes.POINTER(ctypes.c_uint64), _p, _p]),
    "gx_sortx_tables": (_i, [_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32),
                             ctypes.POINTER(ctypes.c_int32), _p]),
    "gx_sortx_level0_buffer": (_p, [_i, _p, _i64, _i64, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "gx_sortx_finish": (_i, [_i, _i64, _i64, _i64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32),
                             ctypes.POINTER(ctypes.c_uint32), _i, _p, _p, _p]),
    "gx_sortx_status": (_i, [_p, ctypes.POINTER(ctypes.c_int32), _p]),
    "gx_gather": (_i, [_i, _p, _p, _i64, _p, _i64, _i, _p, _p, _p]),
    "gx_gather_global_rows": (_i, [_p, _i64, _p, _i64, _i, _p, _p, _p, _p]),
    "gx_gather_global_rows_dev": (_i, [_p, _i64, _p, _i64, _i, _p, _p, _p]),
    "gx_widen_i32_i64": (_i, [_p, _i64, _p, _p]),
    "gx_decode_global_rows": (_i, [_p, _i64, _i, _p, _p, _p, _i, _p, _p]),
    "gx_join_build_pl": (_i, [_i, _p, _p, _p, _i64, _p, ctypes.c_size_t, ctypes.c_double, _p]),
    "gx_join_build_partitioned_pl": (_i, [_i, _p, _p, _i64, _p, ctypes.c_size_t, ctypes.c_double, _p, _sz, _p]),
    "gx_join_probe_partitioned_pl": (_i, [_i, _p, _p, _i64, ctypes.c_int32, _p, ctypes.c_size_t, _i, _p, _p, _i64, _p, _p, _sz, _p]),
    "gx_bitmask_set": (_i, [_p, _i64, _i64, _i, _p]),
    "gx_bitmask_count": (_i, [_p, _i64, _i64, _p, _p]),
    "gx_bitmask_and": (_i, [ctypes.POINTER(_p), _i, _i64, _p, _p, _p]),
    "gx_bitmask_first_unset": (_i, [_p, _i64, _p, _p]),
    "gx_murmur3_32": (_i, [_i, _p, _p, _i64, ctypes.c_uint32, _i, _p, _p]),
    "gx_identity_hash_32": (_i, [_i, _p, _p, _i64, _i, _p, _p]),
    "gx_hash_partition_map": (_i, [_p, _i64, _i, _p, _p, _p, _sz, _p]),
    "gx_join_table_bytes": (ctypes.c_size_t, [_i, _i64, ctypes.c_double]),
    "gx_join_build": (_i, [_i, _p, _p, _i64, _p, ctypes.c_size_t, ctypes.c_double, _p]),
    "gx_join_count": (_i, [_i, _p, _p, _i64, _p, ctypes.c_size_t, _p, _p]),
    "gx_join_probe": (_i, [_i, _p, _p, _i64, _p, ctypes.c_size_t, _i, _p, _p, _i64, _p, _p]),
    "gx_groupby_sum_count": (_i, [_i, _p, _p, _i, _p, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gx_groupby_sum_count_wide": (_i, [_i, ctypes.POINTER(_p), _i, _p, _i64, _i64, ctypes.POINTER(_p), _p, _p, _p, _p, _sz, _p]),
    "gx_groupby_min_max": (_i, [_i, _p, _p, _i, _p, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gx_valid_from_counts": (_i, [_p, _i64, _p, _p, _p]),
    "gx_mean_from_sum": (_i, [_i, _p, _p, _i64, _p, _p]),
    "gx_join_probe_partitioned": (_i, [_i, _p, _i64, _p, ctypes.c_size_t, _i, _p, _p, _i64, _p, _p, _sz, _p]),
    "gx_join_probe_partitioned_at": (_i, [_i, _p, _i64, ctypes.c_int32, _p, ctypes.c_size_t, _i, _p, _p, _i64, _p, _p, _sz, _p]),
    "gx_join_build_partitioned": (_i, [_i, _p, _i64, _p, ctypes.c_size_t, ctypes.c_double, _p, _sz, _p]),
    "gx_partition_rows": (_i, [_i, _p, _i64, _i, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "gx_partition_rows_at": (_i, [_i, _p, _i64, ctypes.c_int32, _i, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "gx_partition_rows_spec_at": (_i, [_i, _p, _i64, ctypes.c_int32, _i, _i, _p, _i64, _p, _p, _p, _p, _sz, _p]),
    "gx_join_count_rows": (_i, [_i, _p, _p, _i64, _p, ctypes.c_size_t, ctypes.c_int32, _p, _p]),
    "gx_add_i32": (_i, [_p, _i64, ctypes.c_int32, _p]),
    "gx_join_partition_bits": (_i, [_i, ctypes.c_size_t]),
    "gx_join_profile": (_i, [_i]),
    "gx_join_profile_slot": (_i, [_i]),
    "gx_join_profile_read": (_i, [ctypes.POINTER(ctypes.c_float)]),
    "gx_join_set_scatter_tile": (None, [_i]),
    "gx_join_set_experiment": (None, [_i]),
    "gx_join_set_overflow_slice": (None, [_i]),
    "gx_join_set_build_kernel": (None, [_i]),
    "gx_join_set_probe_kernel": (None, [_i]),
    "gx_join_set_partition_mode": (None, [_i, _i]),
    "gx_bitmask_copy": (_i, [_p, _i64, _p, _i64, _i64, _p]),
    "gx_pack_keys": (_i, [_i, _p, _p, _i64, _p, _p]),
    "gx_unpack_keys": (_i, [_i, _p, _p, _i64, _p, _p]),
    "gx_hash_rows64": (_i, [_i, _p, _p, _i64, ctypes.c_uint64, _p, _p]),
    "gx_rows_mismatch_count": (_i, [_i, _p, _p, _p, _p, _p, _i64, _p, _p]),
    "gx_dense_rank": (_i, [_i, _p, _p, _i64, _i64, _p, _p, _p, _p, _sz, _p]),
    "gx_square": (_i, [_i, _p, _i64, _p, _p]),
    "gx_var_from_sums": (_i, [_i, _p, _p, _p, _i64, _i, _i, _p, _p, _p, _p]),
    "gx_groupby_arg_select": (_i, [_i, _p, _p, _p, _i64, _p, _i64, _p, _p]),
    "gx_fill_nulls": (_i, [_i, _p, _p, _i64, ctypes.c_uint64, _p]),
    "gx_join_lookup": (_i, [_i, _p, _p, _i64, _p, ctypes.c_size_t, _p, _p]),
    "gx_join_filter": (_i, [_i, _p, _p, _i64, _p, ctypes.c_size_t, _i, _i, _p, _p, _p, _sz, _p]),
    "gx_join_complement": (_i, [_p, _i64, _i64, _p, _p, _i64, _p, _p, _sz, _p]),
    "gx_groupby_set_algorithm": (None, [_i, _i]),
    "gx_groupby_set_dense": (None, [_i]),
    "gx_groupby_plan_info": (_i, [_p, _i64, _p, _p]),
    "gx_groupby_set_partition_mode": (None, [_i]),
    "gx_groupby_set_partition_bits": (_i, [_i]),
    "gx_group_heads": (_i, [_i, _p, _p, _p, _i64, _i, _p, _p]),
    "gx_group_offsets": (_i, [_p, _i64, _p, _p, _p, _p, _p, _sz, _p]),
    "gx_segmented_reduce": (_i, [_i, _p, _p, _p, _p, _i64, _i, _p, _p, _p, _sz, _p]),
    "gx_segmented_shift": (_i, [_i, _p, _p, _p, _i64, _i64, ctypes.c_uint64, _i, _p, _p, _p]),
    "gx_segmented_fill_nulls": (_i, [_i, _p, _p, _p, _i64, _i, _p, _p, _p, _sz, _p]),
    "gx_rank_from_groups": (_i, [_p, _p, _p, _i64, _i, ctypes.c_double, _i, _p, _p, _p]),
    "gx_segment_ids": (_i, [_p, _i64, _i64, _p, _p]),
    "gx_segmented_scan": (_i, [_i, _p, _i, _p, _p, _i64, _i, _p, _p, _sz, _p]),
    "gx_reduce": (_i, [_i, _p, _p, _i64, _i, _i, _p, _p, _p, _sz, _p]),
    "gx_scan": (_i, [_i, _p, _p, _i64, _i, _i, _p, _p, _sz, _p]),
    "gx_fill_random": (_i, [_i, _p, _i64, ctypes.c_uint64, _i64, _i64, _p]),
    "gx_mix64_inplace": (_i, [_p, _i64, _p]),
    "gx_sequence_i32": (_i, [_p, _i64, ctypes.c_int32, _p]),
    "gx_copy_bytes": (_i, [_p, _p, ctypes.c_size_t, _p]),
    "gx_checksum": (_i, [_i, _p, _i64, _i, _p, _p]),
}

for _name, (_res, _args) in _PROTOS.items():
    _f = getattr(lib, _name)  # AttributeError here == the library does not export the ABI
    _f.restype = _res
    _f.argtypes = _args

EXPORTED = tuple(_PROTOS)


class GxError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        raise GxError(f"{what} failed with code {rc}" + (" (hipError)" if rc > 0 else ""))
