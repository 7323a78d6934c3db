"""Per-kernel register / LDS / scratch / occupancy table of the kernel library (hipcc -Rpass-analysis=kernel-resource-usage).
No GPU needed.  Usage: python scripts/kernel_resources.py > profiles/r2_kernel_resources.txt"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
         "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", "-o", "/dev/null"]
# the instantiations the bench configurations run
WANT = re.compile(r"k_hy_hist<unsigned long, 0|k_hy_plan|k_plan2|k_msd_pass<unsigned long, 0, false|k_local_sort<unsigned long, 0, false|"
                  r"k_radix_pass<unsigned long, 0, false|k_hist_all<unsigned long, 0|"
                  r"k_pj_hist<unsigned long, gx::join::TableTop|k_pj_scatter<unsigned long, 16, 1024, gx::join::TableTop|k_pj_probe_pipe<unsigned long>|"
                  r"k_pj_build<unsigned long>|k_tags<unsigned long>|k_lookup<unsigned long>|"
                  r"k_part_hist<unsigned int>|k_part_scatter<unsigned int, double, false>|k_part_aggregate<unsigned int, double, true, false>|"
                  r"k_part_minmax<unsigned int, double, false>|k_part_minmax<unsigned long, long, false>|k_part_aggregate<unsigned long, double, true, false>|"
                  r"k_lookback_scan<unsigned long, long, gx::SumOp.*true>|k_stream_reduce<.*DD|k_stream_reduce<double|"
                  r"k_hash_rows|k_rows_mismatch|k_gather<unsigned long, false>")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.strip().split("\n")


def main():
    print("# per-kernel resource usage, hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage (ROCm 7.2.0), round 2")
    print("# occ = waves per SIMD allowed by registers (LDS limits come on top: static LDS bytes/block shown; dynamic LDS is not included)")
    print("%-14s %5s %5s %5s %8s %4s %9s  %s" % ("file", "VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS(stat)", "kernel"))
    total = spills = 0
    rows = []
    for src in sorted(glob.glob(os.path.join(ROOT, "cudf_amd", "csrc", "*.hip"))):
        err = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, src], capture_output=True, text=True).stderr
        cur = {}
        for line in err.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = {"name": m.group(1)}
                continue
            for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"),
                             ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                             ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
                m = re.search(pat, line)
                if m and cur:
                    cur[key] = int(m.group(1))
                    if key == "lds":
                        rows.append((os.path.basename(src)[:-4], dict(cur)))
                        cur = {}
    names = demangle([r[1]["name"] for r in rows])
    for (f, r), dn in zip(rows, names):
        total += 1
        spills += 1 if r.get("scratch", 0) > 0 else 0
        if WANT.search(dn):
            print("%-14s %5d %5d %5d %8d %4d %9d  %s" % (f, r.get("vgpr", 0), r.get("agpr", 0), r.get("sgpr", 0), r.get("scratch", 0),
                                                        r.get("occ", 0), r.get("lds", 0), dn.split("(")[0][:150]))
    print(f"# {total} kernel instantiations in the library; kernels with register spills (scratch > 0): {spills}")


if __name__ == "__main__":
    main()
