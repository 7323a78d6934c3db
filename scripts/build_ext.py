"""Builds cudf_amd/libcudf_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

Lives outside the package on purpose: importing ``cudf_amd`` requires the library to exist.
Usage: python scripts/build_ext.py [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cudf_amd")
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libcudf_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-munsafe-fp-atomics",  # hardware f64 atomics: all our buffers are coarse-grained device memory
         # a returning atomic issued by ONE lane (tickets, output reservations) must not be followed by an immediate
         # s_waitcnt: LLVM's atomic optimizer rewrites it into reduce + atomic + readfirstlane and waits on the spot
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdrs.append(os.path.join(HERE, "..", "include", "cudf_amd", "gx.h"))
    objs = []
    jobs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src[:-4] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _newer(OUT, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
