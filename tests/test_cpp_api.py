"""The C++ drop-in surface: libcudf.so (cudf:: API over the C ABI).
CPU: it builds with plain g++ and exports the reference's entry points.
GPU: tests/cpp/cudf_api_tests drives cudf::sort / hash_join / groupby / reduce / scan with the
literal vectors of the reference's gtest suites."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "cudf_amd", "libcudf.so")
BIN = os.path.join(ROOT, "tests", "cpp", "cudf_api_tests")


def _build():
    import __graft_entry__ as ge
    ge.build()


def test_host_library_exports_the_cudf_api():
    _build()
    assert os.path.exists(LIB) and os.path.exists(BIN)
    syms = subprocess.check_output(["nm", "-DC", "--defined-only", LIB], text=True)
    for s in ["cudf::sort(cudf::table_view const&", "cudf::sorted_order(cudf::table_view const&",
              "cudf::stable_sorted_order(", "cudf::sort_by_key(", "cudf::inner_join(cudf::table_view const&",
              "cudf::left_join(", "cudf::full_join(", "cudf::hash_join::hash_join(", "cudf::hash_join::inner_join(",
              "cudf::hash_join::inner_join_size(", "cudf::groupby::groupby::aggregate(", "cudf::groupby::groupby::scan(",
              "cudf::reduce(cudf::column_view const&", "cudf::scan(cudf::column_view const&", "cudf::gather(",
              "cudf::hashing::murmurhash3_x86_32(", "cudf::hash_partition(", "cudf::column::release()",
              "cudf::table::table(", "cudf::bitmask_and("]:
        assert s in syms, f"libcudf.so does not export {s}"
    # the host library contains no device code and no CPU implementation of the kernels: every
    # data-path symbol it needs is an undefined reference into libcudf_amd.so
    und = subprocess.check_output(["nm", "-D", "--undefined-only", LIB], text=True)
    for s in ["gx_sort_keys", "gx_sorted_order", "gx_join_build", "gx_join_probe", "gx_groupby_sum_count", "gx_reduce",
              "gx_scan", "gx_gather"]:
        assert s in und


def test_host_side_validation():
    """tests/cpp/cudf_host_tests: constructor / argument checks, aggregation objects and the Arrow schema export --
    everything the C++ surface decides before its first device call -- run without a GPU."""
    _build()
    r = subprocess.run([os.path.join(ROOT, "tests", "cpp", "cudf_host_tests")], capture_output=True, text=True, timeout=120)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 failed" in r.stdout


@pytest.mark.gpu
def test_cpp_api_against_reference_vectors():
    _build()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=600)
    print(r.stdout[-4000:])
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "0 failed" in r.stdout
