#!/bin/bash
# round 4, run 18: the sign fold of the level-0 digit (keys spread around zero) on the cursor and look-back paths -- every sort test,
# loopback sort, sort-using C++ parity tests, then the headline + robustness lines
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
O=gpurun_out; R=18; L=$O/r4_run18.log; : > $L
{
timeout 1200 python -m pytest tests/test_gpu_sort.py tests/test_gpu_sort_place.py tests/test_gpu_sort_cursor_path.py tests/test_gpu_sort_big_cells.py -q 2>&1 | tail -25
timeout 600 python -m pytest tests/test_gpu_distributed_loopback.py -x -q -k "sort" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity_1e8.py tests/test_gpu_cpp_parity.py -x -q -k "sort or rank or top_k or segmented or scan" 2>&1 | tail -3
} > $O/r4_run${R}_tests.log 2>&1
rb() { local tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>> $L | tail -1 > $O/r4_run${R}_bench_${tag}.jsonl; }
rb sort --workload sort --steps 5
rb sort_range1e12 --workload sort --key-range 0 1000000000000
rb sort_signed_range --workload sort --key-range -1000000000000 1000000000000
rb sort_normal --workload sort --key-dist normal
rb sorted_order --workload sorted_order
rb sorted_order_signed_range --workload sorted_order --key-range -1000000000000 1000000000000
cat $O/r4_run${R}_tests.log
python - <<PY | tee $O/r4_run${R}_sort_robustness.txt
import json, glob
print("# round 4 run $R: python bench.py --workload sort|sorted_order --steps 3 (1e9 int64 rows) with the sign fold of the level-0 digit")
for f in sorted(glob.glob("$O/r4_run${R}_bench_*.jsonl")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        si = (d.get("roofline") or {}).get("sort_info") or {}
        print(f.split("_bench_")[1][:-6], "|", d["config"]["workload"], "|", round(d["ms_per_step"], 3), "ms |", {k: si.get(k) for k in ("shift0", "bits2", "max_cell", "lsd_passes", "cursor_path_state", "big_cells")})
    except Exception as e:
        print(f, "unreadable", e)
PY
grep -v amdgpu.ids $L | tail -5
