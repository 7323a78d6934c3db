"""CPU: the C-ABI library loads and exports every symbol include/cudf_amd/gx.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cudf_amd", "gx.h")).read()
    text += open(os.path.join(ROOT, "include", "cudf_amd", "gx_knobs.h")).read()  # tuning / measurement hooks: a separate header
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    lib = ctypes.CDLL(os.path.join(ROOT, "cudf_amd", "libcudf_amd.so"))
    syms = _declared_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_binding_covers_header():
    from cudf_amd import _lib
    assert sorted(_lib.EXPORTED) == _declared_symbols()


def test_pure_host_entry_points():
    from cudf_amd import _lib
    assert b"gfx950" in _lib.lib.gx_version()
    assert [_lib.lib.gx_dtype_size(d) for d in range(1, 12)] == [1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 1]
    # table sizing is host arithmetic: power-of-two slots, load factor <= requested
    b8 = _lib.lib.gx_join_table_bytes(8, 1000, 0.5)
    # header + slots + one 4-bit tag per slot
    assert b8 == 256 + 16 * 2048 + 2048 // 2
    assert _lib.lib.gx_join_table_bytes(4, 1000, 0.5) == 256 + 8 * 2048 + 2048 // 2
    assert _lib.lib.gx_join_table_bytes(3, 1000, 0.5) == 0


def test_public_headers_compile_clean(tmp_path):
    """The drop-in boundary is a C header: gx.h must compile as C99 (-pedantic), and the cudf:: API headers
    as C++20 with -Wall -Wextra, without a GPU and without warnings."""
    import glob
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = ["-I" + os.path.join(root, "include"), "-I/opt/rocm/include"]
    c = tmp_path / "abi.c"
    c.write_text("#include <cudf_amd/gx.h>\nint main(void) { return gx_dtype_size(GX_INT64) == 8 ? 0 : 1; }\n")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", *inc, str(c)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    hdrs = sorted(os.path.relpath(h, os.path.join(root, "include"))
                  for h in glob.glob(os.path.join(root, "include", "cudf", "**", "*.hpp"), recursive=True))
    cpp = tmp_path / "api.cpp"
    cpp.write_text("".join(f"#include <{h}>\n" for h in hdrs) + "int main() { return 0; }\n")
    r = subprocess.run(["g++", "-std=c++20", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-fsyntax-only", *inc, str(cpp)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_join_table_sizing_round_trips_through_the_blob_size():
    """The join table blob is header + slots + 4-bit tags; the probe entry points recover log2(capacity) from
    the blob size alone (no host round trip).  Host arithmetic, checked here for every capacity 2^4..2^33 and
    both slot widths: gx_join_partition_bits = log2(capacity) - 17 (2^17-slot sub-tables), 0 below 2^20, max 12."""
    from cudf_amd import _lib
    lib = _lib.lib
    for key_size, slot in ((8, 16), (4, 8)):
        for lf in (0.5, 1.0, 0.3):
            for rows in [0, 1, 7, 8, 9, 1000, 2**17, 2**19 - 1, 2**19, 2**20, 10**8, 10**9, 2**31 - 1]:
                want = max(rows, 1) / lf + 1.0
                lg = 4
                while float(1 << lg) < want:
                    lg += 1
                nbytes = lib.gx_join_table_bytes(key_size, rows, lf)
                assert nbytes == 256 + (slot << lg) + (1 << lg) // 2, (key_size, lf, rows)
                pb = lg - 17
                pb = 0 if pb < 3 else min(pb, 12)
                assert lib.gx_join_partition_bits(key_size, nbytes) == pb, (key_size, lf, rows, lg)


def test_partition_of_a_key_is_the_top_bits_of_its_slot():
    """Partitioned join: slot = (key * phi) >> (64 - log2cap), partition = (key * phi) >> (64 - pbits) with
    pbits = log2cap - 17, so partition p owns exactly the slots [p, p + 1) << 17 -- the sub-table whose tags a
    workgroup keeps in LDS; the tag is the 4 bits just below the slot index (0 remapped to 8)."""
    import numpy as np
    rng = np.random.default_rng(0)
    keys = rng.integers(0, 2**63, 5000, dtype=np.int64).astype(np.uint64)
    phi = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        prod = keys * phi
    for lg in (20, 24, 28, 29):
        slot = prod >> np.uint64(64 - lg)
        part = prod >> np.uint64(64 - (lg - 17))
        assert np.array_equal(slot >> np.uint64(17), part)
        tag = (prod >> np.uint64(60 - lg)) & np.uint64(15)
        assert np.array_equal(tag, ((prod >> np.uint64(64 - lg - 4)) & np.uint64(15)))   # next 4 bits below the slot index


def test_scratch_queries_and_argument_checks_without_a_gpu():
    """Every data-path entry point follows the cub convention (tmp == NULL -> *tmp_bytes): the query is pure
    host arithmetic, so it must work here; sizes grow with n; bad arguments come back as gx_error codes
    (GX_EINVAL -1, GX_EDTYPE -2) before anything touches the device."""
    import ctypes
    from cudf_amd import _lib as L
    lib = L.lib
    nb = ctypes.c_size_t(0)

    def q(name, *args):
        nb.value = 0
        rc = getattr(lib, name)(*args, None, ctypes.byref(nb), None)
        return rc, nb.value

    queries = {
        "gx_sort_keys": lambda n: q("gx_sort_keys", L.INT64, None, None, n, 0),
        "gx_sort_pairs": lambda n: q("gx_sort_pairs", L.INT64, None, None, None, None, n, 0),
        "gx_sorted_order": lambda n: q("gx_sorted_order", L.FLOAT64, None, None, n, 0, 0, 1, None),
        "gx_groupby_sum_count": lambda n: q("gx_groupby_sum_count", L.INT32, None, None, L.FLOAT64, None, None, n, 1 << 20, None, None, None, None, None),
        "gx_groupby_min_max": lambda n: q("gx_groupby_min_max", L.INT64, None, None, L.INT32, None, None, n, 1 << 20, None, None, None, None, None),
        "gx_reduce": lambda n: q("gx_reduce", L.FLOAT64, None, None, n, L.OP_SUM, L.FLOAT64, None, None),
        "gx_scan": lambda n: q("gx_scan", L.INT64, None, None, n, L.OP_SUM, 1, None),
        "gx_join_filter": lambda n: q("gx_join_filter", 8, None, None, n, None, 0, 0, 0, None, None),
        "gx_dense_rank": lambda n: q("gx_dense_rank", L.UINT64, None, None, n, 0, None, None, None),
    }
    for name, fn in queries.items():
        sizes = []
        for n in (0, 1000, 10**6, 10**9):
            rc, b = fn(n)
            assert rc == 0, (name, n, rc)
            sizes.append(b)
        assert sizes == sorted(sizes) and sizes[-1] > 0, (name, sizes)
    # the 1e9-row keys-only sort: 512 eight-byte look-back granules per 8192-key tile (0.5 GB), the padded level-0 output of the
    # cursor path (n + 8 % of slack for its 2048 sampled slots: 8.7 GB), and the cell buffer of level 1 -- since round 4 sized per
    # bucket from the exact level-0 histogram (the bucket's mean cell + 6 sigma + 64 keys per slot: n keys + 640 per cell = 9.3 GB,
    # + n / 16 so that a half-covered edge bucket can take its neighbours' capacity, where round 3's 2^18 slots of 8192 keys took
    # 17.2 GB) = 19.05 GB next to the caller's 8 GB output column (2.4 x the input; VERDICT r3 item 6: <= 20 GB); a caller that
    # passes no output buffer gets one more key buffer (27.05 GB; round 3: 34.4 GB)
    assert 25.5e9 < queries["gx_sort_keys"](10**9)[1] < 27.5e9
    with_out = q("gx_sort_keys", L.INT64, None, ctypes.c_void_p(256), 10**9, 0)[1]
    assert 17.5e9 < with_out < 20.0e9
    # ... and proportional beyond: 1.25e9 rows (config 5's shard) take 1.25 x that, not the 2^17 slots of 16384 keys the look-back
    # path would want there (it is not the cursor path's fallback above 1.03e9 rows: the LSD passes are)
    assert q("gx_sort_keys", L.INT64, None, ctypes.c_void_p(256), 1_250_000_000, 0)[1] < 1.27 * with_out
    assert q("gx_sort_keys", 99, None, None, 10, 0)[0] == -2                 # GX_EDTYPE
    assert q("gx_sort_keys", L.INT64, None, None, -1, 0)[0] == -1            # GX_EINVAL
    assert q("gx_sort_keys", L.INT64, None, None, 2**31, 0)[0] == -1         # more than size_type rows
    assert q("gx_dense_rank", L.INT64, None, None, 10, 11, None, None, None)[0] == -1   # null_count > n
    assert lib.gx_pack_keys(0, None, None, 0, None, None) == -1
    assert lib.gx_join_table_bytes(3, 1000, 0.5) == 0


def test_import_cudf_is_the_alias_of_cudf_amd():
    """north_star: "the cudf.DataFrame Python wrapper so it is a drop-in" -- `import cudf` resolves to the alias package at the repo
    root and names the same class; what is out of scope raises instead of pretending (no GPU needed: nothing is launched)."""
    import cudf
    import cudf_amd
    assert cudf.DataFrame is cudf_amd.DataFrame and callable(cudf.from_pandas)
    assert sorted(cudf.__all__) == ["DataFrame", "__version__", "from_pandas"]
    import pytest
    with pytest.raises(AttributeError):
        cudf.Series  # noqa: B018
