"""CPU: the C-ABI library loads and exports every symbol include/cudf_amd/gx.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cudf_amd", "gx.h")).read()
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
